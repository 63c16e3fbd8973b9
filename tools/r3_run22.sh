#!/bin/bash
# 40-byte entries staged 48 bytes apart in LDS (s48) vs packed (s40), each against 48-byte entries (variant 128)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for i in 1 2; do
for L in s48 s40; do
  cp gpurun_tmp/lib_$L.so geograypher_amd/csrc/libgeograster.so
  echo "== $L"
  timeout 600 python tools/ab_kernel.py 50 5 short:0 full:128 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['variant'], 'setup', d['plain']['setup_ms'], 'plain', d['plain']['raster_ms'], 'fused-setup', d['fused']['setup_ms'], 'fused', d['fused']['raster_ms'])"
done
done
