"""The engine's own RCCL communicator (pvae_comm_*) on the one GPU of the test box: a one-rank communicator is a real
ncclCommInitRank / ncclAllReduce on the engine's stream, so the binding (dlopen, by-value unique id, in-place all-reduce
on the exchange buffers) and the in-stream placement of the two all-reduces are exercised; results must equal the plain
single-GPU path bit for bit.  Runs in a child process (tests/comm_worker.py) with the library order of real use -- torch,
and with it PyTorch's librccl, before the engine -- and checks that the process also EXITS cleanly.  More ranks need one GPU
each: bench.py --gpus N."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_matches_plain_engine_and_exits_cleanly():
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, os.path.join(here, "comm_worker.py")], capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    assert p.stdout.strip().endswith("OK")
