#!/bin/bash
# Register / scratch / LDS use of every kernel of one .hip file (device assembly in build/): tools/kernel_regs.sh lld_ba
set -e
cd "$(dirname "$0")/.."
mkdir -p build
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=on --cuda-device-only -S lld_slam_amd/csrc/$1.hip -o build/$1.s 2>/dev/null
python3 - "$1" <<'PY'
import re, sys
s = open(f"build/{sys.argv[1]}.s").read()
for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", s, re.S):
    ag, lds, name, priv, sg, vg, sp = m.groups()
    print(f"{name[:70]:70s} vgpr {vg:>4} agpr {ag:>3} sgpr {sg:>3} spill {sp:>3} scratch {priv:>5} lds {lds:>6}")
PY
