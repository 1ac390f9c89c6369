"""CPU: `python bench.py --gpus N` starts its own ranks (torch.distributed.run on 127.0.0.1) -- the way the driver's scaling run
calls it -- and ends loudly where the GPUs are not there: every rank prints "N GPUs requested, V visible" after the launcher ran,
the exit code is non-zero and no record goes to stdout (nothing waits in a rendezvous)."""
import os
import subprocess
import sys

from cuopt_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_more_ranks_than_gpus_ends_with_a_message_not_a_hang():
    visible = capi.device_count()
    want = visible + 2
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--workload", "tiny", "--steps", "5", "--warmup", "2"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "%d GPUs requested, %d visible" % (want, visible) in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
