"""Short runs of the developer fuzzers (tools/fuzz_graph.py, tools/fuzz_auto_graph.py): random geometry / dtype / batch /
extension modes — the eager patched loop against DecodeGraph, and the reference's caller loop with and without auto_graph —
must agree bit for bit.  Needs an MI355X."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), *args], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


def test_random_stacks_eager_loop_equals_decode_graph():
    out = _run("fuzz_graph.py", "24", "7")
    assert "24 / 24 cases agree" in out, out[-3000:]


def test_random_sessions_of_the_reference_loop_with_and_without_auto_graph():
    out = _run("fuzz_auto_graph.py", "14", "5")
    # (a session whose random window sizes leave fewer candidates than important_size raises in BOTH runs — the reference's
    #  own behaviour — and is not a disagreement)
    fails = [l for l in out.splitlines() if l.startswith("FAIL") and "top-k window" not in l]
    assert not fails and "sessions agree" in out, out[-3000:]
