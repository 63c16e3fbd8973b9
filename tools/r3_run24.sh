#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for i in 1 2; do
  timeout 600 python tools/ab_kernel.py 50 5 short:0 full:128 short_nostore:0:2 full_nostore:128:2 short_notri:0:4 full_notri:128:4 short_noloop:0:1 full_noloop:128:1 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['variant'], 'setup', d['plain']['setup_ms'], 'plain', d['plain']['raster_ms'], 'fused-setup', d['fused']['setup_ms'], 'fused', d['fused']['raster_ms'])"
done
