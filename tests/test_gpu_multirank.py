"""GPU rehearsal of the N > 1 pipeline on ONE GPU: two ranks share cuda:0, every kernel is the real
HIP one, only the transport is gloo (host-staged) instead of RCCL. The fused embeddings must be
identical to the single-process run over the same four interval graphs. Every N > 1 run goes through the entry the
driver uses — `python bench.py --gpus N ...` from a bare shell — which starts its own ranks (bench.self_launch)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(cmd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(line) == 1, out.stdout[-2000:]
    return json.loads(line[0])


def test_two_ranks_match_single_process():
    common = ["--steps", "1", "--warmup", "1", "--scale", "0.004", "--no-cpu-baseline"]
    one = _run([sys.executable, "bench.py", "--intervals", "4"] + common)
    two = _run([sys.executable, "bench.py", "--gpus", "2",
                "--dist-backend", "gloo", "--intervals", "4"] + common)
    assert one["config"]["intervals_total"] == two["config"]["intervals_total"] == 4
    assert two["n_gpus"] == 2 and two["config"]["exchange"] == "alltoall"
    assert one["final_abs_mean"] == two["final_abs_mean"]
    assert one["final_position_checksum"] == two["final_position_checksum"]      # row order, not only magnitudes
    assert set(two["breakdown_ms"]) >= {"spmm_only", "exchange_alltoall_only", "fusion_only", "exchange_allgather_only"}
    assert one["roofline"]["launches"] == 16 and two["roofline"]["launches"] == 8      # rank 0's SpMM launches


def test_three_ranks_round_wise_fusion_matches_single_process():
    """Three ranks, six intervals: two exchange rounds, the LSTM continued across them."""
    common = ["--steps", "1", "--warmup", "1", "--scale", "0.003", "--no-cpu-baseline"]
    one = _run([sys.executable, "bench.py", "--intervals", "6"] + common)
    three = _run([sys.executable, "bench.py", "--gpus", "3",
                  "--dist-backend", "gloo", "--intervals", "6"] + common)
    assert one["config"]["intervals_total"] == three["config"]["intervals_total"] == 6
    assert one["final_abs_mean"] == three["final_abs_mean"]
    assert one["final_position_checksum"] == three["final_position_checksum"]


@pytest.mark.parametrize("world,T", [(2, 3), (3, 7)])
def test_interval_count_not_a_multiple_of_the_rank_count(world, T):
    """T = 3 on 2 ranks, 7 on 3: the last exchange round is short, a rank with one interval fewer joins it with empty
    sends — posted in the order the other ranks post it (bench.spmm_stack), or the users' all-gather that the other
    ranks issue in between meets it out of order and the run deadlocks (found by tools/fuzz_ranks.py)."""
    common = ["--steps", "1", "--warmup", "1", "--scale", "0.002", "--no-cpu-baseline", "--intervals", str(T)]
    one = _run([sys.executable, "bench.py"] + common)
    many = _run([sys.executable, "bench.py", "--gpus", str(world), "--dist-backend", "gloo"] + common)
    assert one["config"]["intervals_total"] == many["config"]["intervals_total"] == T
    assert one["final_abs_mean"] == many["final_abs_mean"]
    assert one["final_position_checksum"] == many["final_position_checksum"]
    assert many["breakdown_ms"]["exchange_rounds"] == -(-T // world) and "error" not in many["breakdown_ms"]


def test_four_ranks_match_single_process():
    """Four ranks sharing the GPU on the DEFAULT configuration (strong scaling: the 16 intervals of configs[4], four
    per rank, four exchange rounds, T = 16 fusion) at a reduced scale."""
    common = ["--steps", "1", "--warmup", "1", "--scale", "0.002", "--no-cpu-baseline"]
    one = _run([sys.executable, "bench.py"] + common)
    four = _run([sys.executable, "bench.py", "--gpus", "4",
                 "--dist-backend", "gloo"] + common)
    assert one["scaling"] == four["scaling"] == "strong"
    assert one["config"]["intervals_total"] == four["config"]["intervals_total"] == 16
    assert one["final_abs_mean"] == four["final_abs_mean"]
    assert one["final_position_checksum"] == four["final_position_checksum"]


@pytest.mark.parametrize("split", ["fractional", "groups"])
def test_fewer_intervals_than_ranks_split_rows_match_single_process(split):
    """Gowalla-shaped (T = 3, L = 2) on 4 ranks. fractional (the default): every rank computes a quarter of the
    intervals' target rows laid end to end — ranks 1 and 2 the tail of one interval and the head of the next — and
    takes part in the layer all-gathers of both (parallel.FractionalRunner); groups: one interval is computed by a
    group of two ranks (parallel.SplitIntervalRunner). The ONE all-to-all and the fusion are unchanged. Real kernels,
    gloo transport, one GPU. (Four ranks: the GPU box allows six processes on the card, and the test runner and the
    launcher are two of them; T = 5 on 8 ranks runs on the CPU in tests/test_parallel.py.)"""
    common = ["--workload", "gowalla-shaped", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    one = _run([sys.executable, "bench.py"] + common)
    six = _run([sys.executable, "bench.py", "--gpus", "4", "--split", split,
                "--dist-backend", "gloo"] + common)
    assert one["config"]["intervals_total"] == six["config"]["intervals_total"] == 3
    assert "T < world" in six["config"]["partitioning"]
    assert one["config"]["edges_per_interval"] == six["config"]["edges_per_interval"]
    # the running sum is built in a different association (acc += e^l per layer instead of the fused
    # epilogue order), so the last bits may differ: compare to 1e-6 relative
    for a_, b_ in zip(one["final_abs_mean"] + one["final_position_checksum"], six["final_abs_mean"] + six["final_position_checksum"]):
        assert abs(a_ - b_) <= 1e-6 * abs(a_)


@pytest.mark.parametrize("stages", ["full", "train"])
def test_rccl_single_rank_runs_the_multi_rank_code_path(stages):
    """What a one-GPU box can check of RCCL: `--dist-single` initialises the nccl process group with ONE rank and sends
    the pipeline through the N > 1 code (exchange rounds posted after every interval's SpMMs, round-wise LSTM, chunked
    and asynchronous all-gathers, the adjoints in training) with device tensors — argument types, devices, contiguity,
    split lists. Same results as the plain single-process run."""
    common = ["--steps", "2" if stages == "train" else "1", "--warmup", "0" if stages == "train" else "1", "--scale", "0.004",
              "--no-cpu-baseline", "--intervals", "4", "--stages", stages]
    one = _run([sys.executable, "bench.py"] + common)
    rccl = _run([sys.executable, "bench.py", "--dist-single"] + common)
    assert rccl["n_gpus"] == 1 and rccl["config"]["exchange"] == "alltoall"
    tol = 2e-5 if stages == "train" else 0.0
    for a_, b_ in zip(one["final_abs_mean"] + one["final_position_checksum"], rccl["final_abs_mean"] + rccl["final_position_checksum"]):
        assert abs(a_ - b_) <= tol * abs(a_), (one["final_abs_mean"], rccl["final_abs_mean"])
    if stages == "full":
        assert set(rccl["breakdown_ms"]) >= {"spmm_only", "exchange_alltoall_only", "fusion_only", "exchange_allgather_only"}


def test_rccl_two_gpus_smoke():
    """The N = 2 pipeline over RCCL itself (backend nccl); needs two visible GPUs, skipped on the 1-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (RCCL over xGMI)")
    common = ["--steps", "1", "--warmup", "1", "--scale", "0.004", "--no-cpu-baseline", "--intervals", "4"]
    one = _run([sys.executable, "bench.py"] + common)
    two = _run([sys.executable, "bench.py", "--gpus", "2"] + common)
    assert one["final_abs_mean"] == two["final_abs_mean"]
    assert one["final_position_checksum"] == two["final_position_checksum"]


def test_two_rank_training_step_matches_single_process():
    """Forward + backward + Adam at N = 2 (gloo rehearsal on one GPU): the exchange and the gather carry their adjoints
    (reverse all-to-all, reduce-scatter), the fusion weights' gradients are all-reduced. After two steps the fused
    embeddings of both runs agree (weight-gradient sums use float atomics: not bit for bit)."""
    common = ["--stages", "train", "--steps", "2", "--warmup", "0", "--scale", "0.002", "--no-cpu-baseline", "--intervals", "4"]
    one = _run([sys.executable, "bench.py"] + common)
    two = _run([sys.executable, "bench.py", "--gpus", "2",
                "--dist-backend", "gloo"] + common)
    for a_, b_ in zip(one["final_abs_mean"] + one["final_position_checksum"], two["final_abs_mean"] + two["final_position_checksum"]):
        assert abs(a_ - b_) <= 2e-5 * abs(a_), (one["final_abs_mean"], two["final_abs_mean"])


def test_training_with_fewer_intervals_than_ranks_matches_single_process():
    """Forward + backward + Adam of the Gowalla-shaped workload (T = 3) on 4 ranks sharing the GPU (gloo): fractional row
    stretches, masks recorded per row slice, the backward on all-gathered masked gradient tables, the reverse
    all-to-all, replicated embedding tables stepped identically inside every interval's group. After two steps the fused
    embeddings agree with the single-process run (weight-gradient sums use float atomics: not bit for bit)."""
    common = ["--workload", "gowalla-shaped", "--stages", "train", "--steps", "2", "--warmup", "0", "--no-cpu-baseline"]
    one = _run([sys.executable, "bench.py"] + common)
    four = _run([sys.executable, "bench.py", "--gpus", "4", "--dist-backend", "gloo"] + common)
    assert "T < world" in four["config"]["partitioning"] and four["config"]["stages"] == "train"
    for a_, b_ in zip(one["final_abs_mean"] + one["final_position_checksum"], four["final_abs_mean"] + four["final_position_checksum"]):
        assert abs(a_ - b_) <= 2e-5 * abs(a_), (one["final_abs_mean"], four["final_abs_mean"])
