#!/bin/bash
# tools/rebuild.sh -- build the HIP library and the oracle; prints the last line on success, the error otherwise (exit 1)
cd "$(dirname "$0")/.." && python -c "import __graft_entry__ as g; g.build()" > /tmp/gr_build.log 2>&1 && tail -1 /tmp/gr_build.log || { tail -15 /tmp/gr_build.log; exit 1; }
