R=${GRAFT_REPO_ROOT:-$(pwd)}
NS="1 2 4 8" GS="0" RS="1,1,1,1 2,1,2,1 4,2,4,1 8,2,4,1" bash $R/tools/experiments/exp_small_rounds.sh
NS="24 48 64" GS="0 4" RS="4,2,4,1 8,2,4,1" bash $R/tools/experiments/exp_small_rounds.sh
NS="32" GS="3 4 5 6" RS="8,2,4,1 8,4,8,1 8,2,8,2" bash $R/tools/experiments/exp_small_rounds.sh
