# A/B of two (or more) builds of the library in ONE gpurun call (box-to-box spread is +-3 %, so only same-box comparisons count):
#   bash tools/experiments/exp_ab_libs.sh [windows] [lib ...]     default libs: liblld_amd.so liblld_amd_exp.so; each is run twice, interleaved
R=${GRAFT_REPO_ROOT:-$(pwd)}
NW=${1:-256}; shift
LIBS=${@:-liblld_amd.so liblld_amd_exp.so}
for rep in 1 2; do
for lib in $LIBS; do
  export LLD_AMD_LIB=$R/lld_slam_amd/csrc/$lib
  python3 $R/bench.py --windows-per-gpu $NW --steps 10 --warmup 3 --no-secondary --no-e2e --no-cpu-baseline --no-rccl-check 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('%-24s %4d windows  %8.1f windows/s  %7.3f ms/step  phases(1 stream) %s' % ('$lib', $NW, d['value'], d['ms_per_step'], d['roofline']['phase_ms_single_stream_step']))"
done
done
