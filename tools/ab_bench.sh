#!/bin/bash
# A/B of alternative builds of liblld_amd.so on one box: tools/ab_bench.sh <rounds> <label=path-or-"default"[:ENVVAR=value]> ...   (one line per run: label, windows/s, ms per step, phase table)
R=$1; shift
for i in $(seq 1 $R); do
  for spec in "$@"; do
    L=${spec%%=*}; P=${spec#*=}; E=""
    case "$P" in *:*) E=${P#*:}; P=${P%%:*};; esac
    if [ "$P" = default ]; then unset LLD_AMD_LIB; else export LLD_AMD_LIB=$PWD/$P; fi
    env $E python bench.py --steps 5 --no-cpu-baseline --no-secondary --no-e2e --no-rccl-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'], d['roofline']['phase_ms_single_stream_step'])"
  done
done
