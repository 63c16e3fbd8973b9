#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for i in 1 2 3; do
  timeout 600 python tools/ab_kernel.py 50 5 spec:0 nospec:8 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['variant'], 'setup', d['plain']['setup_ms'], 'plain', d['plain']['raster_ms'], 'fused-setup', d['fused']['setup_ms'], 'fused', d['fused']['raster_ms'])"
done
