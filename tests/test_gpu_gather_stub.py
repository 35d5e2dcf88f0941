"""The N > 1 branch of rt_gather_tiles_device (rust-tracing_amd/csrc/rt_gather.cpp: grouped ncclSend / ncclRecv, peers, slot
offsets, counts, ragged last shards) executed on a one-GPU box: a child process binds librt_amd to tests/rccl_stub's in-process
stand-in for RCCL (RT_RCCL_LIB) and drives 3, 8 and 30 rt_comm ranks on the one device.  The real library over xGMI needs the
driver's multi-GPU node; everything on OUR side of the ncclSend / ncclRecv calls runs here."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
STUB = ROOT / "tests" / "rccl_stub" / "librccl_stub.so"


def run_worker(*args, timeout=600, **env_extra):
    assert STUB.exists(), "tests/rccl_stub/librccl_stub.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    env = dict(os.environ, RT_RCCL_LIB=str(STUB), **env_extra)
    return subprocess.run([sys.executable, str(ROOT / "tests" / "_stub_gather_worker.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)


def test_gather_of_3_8_and_30_ranks_reassembles_the_whole_frame(gpu):
    p = run_worker()
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), p.stdout[-2000:] + p.stderr[-4000:]
    assert p.stdout.count("frames identical") == 5, p.stdout


def test_a_gather_whose_sides_disagree_or_whose_peer_is_missing_is_an_error_not_a_hang(gpu):
    p = run_worker("mismatch", timeout=120, RCCL_STUB_TIMEOUT_S="3")
    assert p.returncode == 0 and "mismatch refused" in p.stdout and "missing peer reported" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_a_communicator_that_cannot_come_up_is_reported(gpu, rt):
    """ncclCommInitRank failing (RCCL_STUB_FAIL=init) surfaces as RT_ERR_COMM with RCCL's own message."""
    code = ("import importlib, sys; sys.path.insert(0, %r); rt = importlib.import_module('rust-tracing_amd')\n"
            "try:\n    rt.Comm.create(rt.Comm.unique_id(), 0, 2, 0)\nexcept rt.RtError as e:\n    print('refused:', e)\n" % str(ROOT))
    env = dict(os.environ, RT_RCCL_LIB=str(STUB), RCCL_STUB_FAIL="init")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "refused:" in p.stdout and "ncclCommInitRank" in p.stdout, p.stdout + p.stderr
