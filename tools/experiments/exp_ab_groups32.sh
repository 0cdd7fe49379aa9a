R=${GRAFT_REPO_ROOT:-$(pwd)}
python -m pytest tests/test_gpu_ba.py -m gpu -q -k "two_unconnected" 2>&1 | grep -E "Error|assert|^E" | head -20
for rep in 1 2; do for g in 2 3 4; do
python3 $R/bench.py --windows-per-gpu 32 --groups $g --steps 12 --warmup 3 --no-secondary --no-e2e --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('windows 32 groups $g  %8.1f windows/s  %7.3f ms/step' % (d['value'], d['ms_per_step']))"
done; done
