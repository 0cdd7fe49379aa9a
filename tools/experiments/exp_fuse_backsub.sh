R=${GRAFT_REPO_ROOT:-$(pwd)}
export LLD_AMD_LIB=$R/lld_slam_amd/csrc/liblld_amd_exp.so
for n in 256 64; do for rep in 1 2; do for f in -1 9999; do
LLD_BA_FUSE_BS_BELOW=$f python3 $R/bench.py --windows-per-gpu $n --steps 10 --warmup 3 --no-secondary --no-e2e --no-cpu-baseline --no-rccl-check 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('windows %4d  fuse backsub below %5d  %8.1f windows/s  %7.3f ms/solve  %s' % ($n, $f, d['value'], d['ms_per_step'], d['roofline']['phase_ms_single_stream_step']))"
done; done; done
