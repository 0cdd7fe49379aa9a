R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/tools/experiments/exp_ab_libs.sh 256 liblld_amd_base.so liblld_amd_sr6.so
export LLD_AMD_LIB=$R/lld_slam_amd/csrc/liblld_amd_exp.so
for c in 256 128 512 1024 256; do
LLD_BA_CHUNK=$c python3 $R/bench.py --windows-per-gpu 256 --steps 10 --warmup 3 --no-secondary --no-e2e --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('chunk $c  %8.1f windows/s  %7.3f ms/step  %s' % (d['value'], d['ms_per_step'], d['roofline']['phase_ms_single_stream_step']))"
done
