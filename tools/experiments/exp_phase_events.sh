# Cost of the per-phase HIP events (6 records per super-step) on small, latency-bound batches; one gpurun call:
#   bash tools/experiments/exp_phase_events.sh [windows ...]      default 1 4 16 32
R=${GRAFT_REPO_ROOT:-$(pwd)}
for NW in ${@:-1 4 16 32}; do
for rep in 1 2; do
for ev in with without; do
  if [ $ev = with ]; then export LLD_PHASE_EVENTS=1; else unset LLD_PHASE_EVENTS; fi
  python3 $R/tools/experiments/exp_phase_events.py $NW 30 2>/dev/null
done
done
done
