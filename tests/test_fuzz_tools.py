"""The differential campaigns of tools/ (fuzz_stages.py, fuzz_api.py) have a `selftest` mode in which the oracle-backed stand-in
takes the library's place on BOTH sides: it checks the checkers (generators, expectations, comparisons) without a GPU, so that
the tools a GPU box runs do not rot.  A few seconds each."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("tool,seed", [("fuzz_stages.py", 200000), ("fuzz_api.py", 600000)])
def test_campaign_tools_agree_with_themselves(tool, seed):
    res = subprocess.run([sys.executable, str(ROOT / "tools" / tool), "4", str(seed), "selftest"], capture_output=True, text=True,
                         timeout=300)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert res.returncode == 0 and lines, (res.stdout[-800:], res.stderr[-800:])
    summary = json.loads(lines[-1])
    assert summary["failures"] == 0 and summary["cases"] >= 5, summary
