"""The randomised parity campaign and the determinism soak as part of the GPU suite (round 5; before, only their logs under
gpurun_out/fuzz/ recorded them): tools/fuzz_parity.py in its four modes with fixed seeds and tools/soak.py for 100 repetitions of
the bench batch -- about two minutes on the MI355X box.  Each runs as its own process, one at a time (the campaign creates and
destroys hundreds of contexts; a child process keeps that away from the session's fixtures)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    env = dict(os.environ)
    env.pop("ORBX_LIB", None)  # the shipped library, not an instrumented build
    p = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    tail = "\n".join(p.stdout.strip().splitlines()[-12:])
    assert p.returncode == 0, tail
    return p.stdout.strip().splitlines()[-1]


@pytest.mark.parametrize("mode,trials,seed", [("mixed", 220, 501), ("big", 60, 502), ("batched", 50, 503), ("stateful", 14, 504)])
def test_fuzz_parity(mode, trials, seed):
    """Random frame sizes, extractor parameters, image content, batch sizes and matcher settings through the host API, the batched
    device API, the fused extract + match call and its stream-ordered form, bit for bit against the oracle."""
    last = _run([os.path.join(ROOT, "tools", "fuzz_parity.py"), str(trials), str(seed), mode], 420)
    assert last.startswith("FUZZ OK") and " 0 mismatching" in last, last


def test_soak_determinism():
    """The bench batch (256 frames 640x480 + 128 pairs) a hundred times through the synchronous and the stream-ordered call:
    every output buffer byte-identical to the first run."""
    last = _run([os.path.join(ROOT, "tools", "soak.py"), "100"], 300)
    assert last == "SOAK OK: 100 steps, 0 mismatching", last
