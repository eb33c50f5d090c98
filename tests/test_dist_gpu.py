"""The N-rank code path on ONE GPU (r4): bench.py under torch.distributed.run with one rank creates an RCCL process group and runs
every collective of a node run - all_reduce of ones, flat weight broadcast, all_gather of the latents, barriers, max-over-ranks
timing - so that the first 8-GPU launch has no untested call left (the build container has no GPU and the test boxes have one)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


SMALL = ["--steps", "1", "--warmup", "0", "--batch", "2", "--sample-steps", "3", "--views", "2", "--res", "64", "--no-cpu-baseline", "--no-probes"]


def _line(out):
    for ln in reversed(out.strip().splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    raise AssertionError("no JSON line in: " + out[-2000:])


def test_bench_world1_rccl_launcher_path(hip_lib):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["finite"] is True
    assert d["collectives"]["backend"] == "nccl" and d["collectives"].get("rccl_version")
    assert d["bcast_ms"] >= 0.0


def test_bench_dist_flag_respawns_under_the_launcher(hip_lib):
    """`python bench.py --gpus 1 --dist` = the same thing without typing the launcher command."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dist"] + SMALL, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["collectives"]["backend"] == "nccl" and d["ranks_seen"] == 1


def test_bench_mismatched_launch_fails_fast(hip_lib):
    """--gpus 2 under a 1-rank launcher: refused before any rendezvous, non-zero exit (a hang here would cost the driver its slot)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0
    assert "refusing to print a line" in (r.stderr + r.stdout)
