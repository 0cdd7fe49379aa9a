R=${GRAFT_REPO_ROOT:-$(pwd)}
export LLD_AMD_LIB=$R/lld_slam_amd/csrc/liblld_amd_exp.so
for n in 32 64; do for rep in 1 2; do
for cfg in "24 24 -1" "24 0 0" "24 0 9999" "24 24 0" "9999 24 -1" "9999 9999 -1"; do
  set -- $cfg
  LLD_BA_CHUNK_FROM=$1 LLD_BA_FUSE_BELOW=$2 LLD_BA_FUSE_BS_BELOW=$3 python3 $R/bench.py --windows-per-gpu $n --steps 12 --warmup 3 --no-secondary --no-e2e --no-cpu-baseline --no-rccl-check 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('windows %4d  queue below %5d  fuse lin below %5d  fuse backsub below %5d  %8.1f windows/s  %7.3f ms/solve' % ($n, $1, $2, $3, d['value'], d['ms_per_step']))"
done; done; done
