"""GPU rehearsal of the N > 1 pipeline on ONE GPU: two ranks share cuda:0, every kernel is the real
HIP one, only the transport is gloo (host-staged) instead of RCCL. The fused embeddings must be
identical to the single-process run over the same four interval graphs."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(cmd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(line) == 1, out.stdout[-2000:]
    return json.loads(line[0])


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_match_single_process():
    common = ["--steps", "1", "--warmup", "1", "--scale", "0.004", "--no-cpu-baseline"]
    one = _run([sys.executable, "bench.py", "--intervals-per-gpu", "4"] + common)
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(_port()), "bench.py", "--gpus", "2",
                "--dist-backend", "gloo"] + common)
    assert one["config"]["intervals_total"] == two["config"]["intervals_total"] == 4
    assert two["n_gpus"] == 2 and two["config"]["exchange"] == "alltoall"
    assert one["final_abs_mean"] == two["final_abs_mean"]
    assert one["roofline"]["launches"] == 16 and two["roofline"]["launches"] == 8      # rank 0's SpMM launches


def test_three_ranks_round_wise_fusion_matches_single_process():
    """Three ranks, six intervals: two exchange rounds, the LSTM continued across them."""
    common = ["--steps", "1", "--warmup", "1", "--scale", "0.003", "--no-cpu-baseline"]
    one = _run([sys.executable, "bench.py", "--intervals-per-gpu", "6"] + common)
    three = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                  "--master-addr", "127.0.0.1", "--master-port", str(_port()), "bench.py", "--gpus", "3",
                  "--dist-backend", "gloo"] + common)
    assert one["config"]["intervals_total"] == three["config"]["intervals_total"] == 6
    assert one["final_abs_mean"] == three["final_abs_mean"]


def test_four_ranks_match_single_process():
    """Four ranks sharing the GPU, eight intervals (two per rank, as the scaling benchmark runs)."""
    common = ["--steps", "1", "--warmup", "1", "--scale", "0.002", "--no-cpu-baseline"]
    one = _run([sys.executable, "bench.py", "--intervals-per-gpu", "8"] + common)
    four = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
                 "--master-addr", "127.0.0.1", "--master-port", str(_port()), "bench.py", "--gpus", "4",
                 "--dist-backend", "gloo"] + common)
    assert one["config"]["intervals_total"] == four["config"]["intervals_total"] == 8
    assert one["final_abs_mean"] == four["final_abs_mean"]
