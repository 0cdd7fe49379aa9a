# 256 windows against the number of stream groups and hardware queues (experiments build):  bash tools/experiments/exp_groups256.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
export LLD_AMD_LIB=$R/lld_slam_amd/csrc/liblld_amd_exp.so
for hq in 16 24 32; do for g in 4 6 8; do
GPU_MAX_HW_QUEUES=$hq LLD_BA_GROUPS=$g python3 $R/bench.py --windows-per-gpu 256 --steps 10 --warmup 3 --no-secondary --no-e2e --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('hw queues $hq groups $g  %8.1f windows/s  %7.3f ms/step' % (d['value'], d['ms_per_step']))"
done; done
