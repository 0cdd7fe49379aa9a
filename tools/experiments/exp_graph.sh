# hipGraph replay of the queued super-steps of small groups against plain stream launches (experiments build, LLD_BA_GRAPH=1), one gpurun call:
#   bash tools/experiments/exp_graph.sh [windows ...]      default 1 4 12
R=${GRAFT_REPO_ROOT:-$(pwd)}
export LLD_AMD_LIB=$R/lld_slam_amd/csrc/liblld_amd_exp.so
for NW in ${@:-1 4 12}; do
for rep in 1 2; do
for g in off on; do
  if [ $g = on ]; then export LLD_BA_GRAPH=1; else unset LLD_BA_GRAPH; fi
  python3 $R/tools/experiments/exp_phase_events.py $NW 30 2>&1 | tail -1
done
done
done
