#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do
  timeout 600 python tools/ab_kernel.py 50 5 short:0 full:128 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['variant'], 'setup', d['plain']['setup_ms'], 'plain', d['plain']['raster_ms'], 'fused-setup', d['fused']['setup_ms'], 'fused', d['fused']['raster_ms'])"
done
