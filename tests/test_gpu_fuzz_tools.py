"""Short runs of the fuzz tools under tools/exp (each compares the HIP path with the oracle and exits non-zero on any difference):
one long-lived context through random parameter / tuning / batch changes, point tables, per-stage entry points at extreme poses,
the entry points next to the path, the asynchronous pipeline's ordering, frame-by-frame sequences through a ring of slots, many threads with a context each, every upload form.  The long runs are
in profiles/r06/*_fuzz.txt (tools/verify_all.sh)."""
import os
import subprocess
import sys

import pytest

ARITH_INDEPENDENT = True   # the tools draw the arithmetic set themselves
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("tool,args", [("stateful_fuzz.py", ["120", "21"]), ("points_fuzz.py", ["60", "21"]), ("stage_fuzz.py", ["80", "21"]),
                                       ("aux_fuzz.py", ["60", "21"]), ("stream_fuzz.py", ["150", "21"]), ("sequence_fuzz.py", ["40", "21"]),
                                       ("thread_fuzz.py", ["4", "150", "21"]), ("upload_fuzz.py", ["120", "21"])])
def test_fuzz_tool_finds_no_difference(tool, args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exp", tool)] + args, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "0 differ" in r.stdout or " 0/" in r.stdout
