"""-m gpu: a fixed-seed slice of the three randomised cross-checks (tools/fuzz_engines.py, fuzz_score.py, fuzz_carnn.py) - the tools that
found round 3's two real bugs - so that the driver's test run executes them too.  Each tool runs in its own process (one context, every
configuration meeting the workspace an arbitrary earlier one left behind) and exits non-zero on the first disagreement; ~20 s each."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,args", [("fuzz_engines.py", ["150", "0"]), ("fuzz_score.py", ["40", "0"]), ("fuzz_carnn.py", ["25", "0"])])
def test_fuzz_slice(tool, args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + args, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "%s %s failed:\n%s\n%s" % (tool, " ".join(args), r.stdout[-3000:], r.stderr[-3000:])
