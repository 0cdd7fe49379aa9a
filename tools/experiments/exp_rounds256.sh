# Tasks per wavefront of the four landmark kernels (lin_pt, lin_ln, backsub_pt, backsub_ln) at 256 windows; experiments build (LLD_BA_ROUNDS), one gpurun call:
#   bash tools/experiments/exp_rounds256.sh ["r0,r1,r2,r3" ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
export LLD_AMD_LIB=$R/lld_slam_amd/csrc/liblld_amd_exp.so
for rs in ${@:-16,4,16,1 8,4,16,1 12,4,16,1 24,4,16,1 16,2,16,1 16,3,16,1 16,6,16,1 16,4,8,1 16,4,32,1 16,4,16,2 16,4,16,4 16,4,16,1}; do
  LLD_BA_ROUNDS=$rs python3 $R/bench.py --windows-per-gpu ${NW:-256} --steps 8 --warmup 2 --no-secondary --no-e2e --no-cpu-baseline --no-rccl-check 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('rounds %-12s  %8.1f windows/s  %7.3f ms/solve  %s' % ('$rs', d['value'], d['ms_per_step'], d['roofline']['phase_ms_single_stream_step']))"
done
