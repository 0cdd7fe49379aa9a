# ba_schur_items_both dispatch tile (windows per tile, LLD_BA_SCHUR_TILE, experiments build): time of the Schur phase and FETCH_SIZE of the kernel
#   bash tools/experiments/exp_schur_tile.sh [tiles ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
export LLD_AMD_LIB=$R/lld_slam_amd/csrc/liblld_amd_exp.so
OUT=$R/gpurun_out/schur_tile; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for t in ${@:-1 8 32 64 256}; do
  export LLD_BA_SCHUR_TILE=$t
  for rep in 1 2; do
  python3 $R/bench.py --windows-per-gpu 256 --steps 8 --warmup 2 --no-secondary --no-e2e --no-cpu-baseline --no-rccl-check 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('tile %4d  %8.1f windows/s  %7.3f ms/solve  ba_schur %.3f ms' % ($t, d['value'], d['ms_per_step'], d['roofline']['phase_ms_single_stream_step']['ba_schur']))"
  done
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/t$t -o f -- python3 $R/bench.py --steps 1 --warmup 1 --windows-per-gpu 256 --no-cpu-baseline --no-secondary --no-e2e --no-rccl-check --gen-workers 1 --groups 1 > $OUT/t$t.log 2>&1
  python3 $R/tools/rocpd_summary.py $(find $OUT/t$t -name "*_results.db" | head -1) 2>/dev/null | grep -E "schur_items_both.*FETCH_SIZE" | cut -c1-160
  rm -rf $OUT/t$t
done
