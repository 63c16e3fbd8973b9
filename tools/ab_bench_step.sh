#!/bin/bash
# tools/ab_bench_step.sh <rounds> <tag> ... -- the bench's end-to-end step (ms_per_step, C3 views/s) for library builds on one box:
# `base` = libgeograster.so, anything else csrc/libgeograster_<tag>.so.  The order of the builds is ROTATED from round to round: two
# copies of one library, always run in the same order, differ by 0.8 % (the second run of a pair finds a warmer chip).
R=$1; shift
L=("$@"); N=${#L[@]}
for r in $(seq 1 $R); do
  for k in $(seq 0 $((N - 1))); do
    T=${L[$(((k + r - 1) % N))]}
    if [ $T = base ]; then unset GEOGRAYPHER_AMD_LIB; else export GEOGRAYPHER_AMD_LIB=$PWD/geograypher_amd/csrc/libgeograster_$T.so; fi
    python bench.py --no-cpu-baseline --no-api --no-io --no-c4 --no-c5 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r = j['roofline']
print('$T', 'round $r', 'ms_per_step', j['ms_per_step'], 'value', j['value'], 'kernel_ms', r['kernel_ms_per_launch'], 'setup_us', r['setup_us_per_view'], 'c3', j['aggregate']['views_per_s'], 'forest', r.get('hostile_gpix_scale_1'), r.get('hostile_gpix_scale_0.25'), 'c2q', r.get('c2_quarter_scale_gpix'))"
  done
done
