R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do for lib in liblld_amd_head.so liblld_amd.so; do
  echo "== $lib"; LLD_AMD_LIB=$R/lld_slam_amd/csrc/$lib python3 $R/tools/time_lba_single.py
  for n in 8 32; do LLD_AMD_LIB=$R/lld_slam_amd/csrc/$lib python3 $R/bench.py --windows-per-gpu $n --steps 12 --warmup 3 --no-secondary --no-e2e --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('windows %4d  %8.1f windows/s  %7.3f ms/step  %s' % ($n, d['value'], d['ms_per_step'], d['roofline']['phase_ms_single_stream_step']))"; done
done; done
