# host-buffer pipeline (3 lanes x 8 batches of 256 windows) of two or more builds, interleaved, in one gpurun call:
#   bash tools/experiments/exp_ab_e2e.sh liblld_amd_base.so liblld_amd.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do for lib in "$@"; do
  echo "== $lib"; LLD_AMD_LIB=$R/lld_slam_amd/csrc/$lib python3 $R/tools/experiments/exp_e2e_lanes.py 256 8 3 2>&1 | grep "^lanes"
done; done
