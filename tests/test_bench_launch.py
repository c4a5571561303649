"""bench.py as the driver calls it: `python bench.py --gpus N --steps K --warmup W`, with and without a launcher
(replaces the reference's single-process loop, cald_train.py:434-447, and the pickled all_gather of
detection/utils.py:75-115).  The GPU tests run the N = 2 command on whatever the box has: RCCL across two devices when two are
visible, two ranks sharing the one GPU over gloo otherwise -- the JSON line says which."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = ["--no-cpu-baseline", "--no-full-pool", "--no-f16x3", "--no-train", "--no-cfg4"]


def _run(argv, env=None, timeout=900):
    e = dict(os.environ, **(env or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    p = subprocess.run([sys.executable] + argv, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None), lines


def test_self_launch_fails_loudly_without_a_gpu():
    """No GPU -> every rank refuses (no CPU fallback), the launcher reaps its ranks and returns non-zero instead of hanging
    or printing a number."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-box behaviour")
    p, line, _ = _run(["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"] + FAST, timeout=300)
    assert p.returncode != 0
    assert line is None
    assert "needs an MI355X" in p.stderr


@pytest.mark.gpu
def test_bench_two_ranks_self_launched_equals_one_rank():
    """The driver's command, unchanged, for N = 1 and N = 2: rc 0, ONE JSON line, n_gpus == N, both ranks seen through the
    collective, the same fixed pool (strong scaling) -> the SAME selected indices as the 1-rank run."""
    args = ["--steps", "2", "--warmup", "1"] + FAST
    p1, one, l1 = _run(["bench.py", "--gpus", "1"] + args)
    assert p1.returncode == 0, p1.stderr[-2000:]
    p2, two, l2 = _run(["bench.py", "--gpus", "2"] + args)
    assert p2.returncode == 0, p2.stderr[-2000:]
    assert len(l1) == 1 and len(l2) == 1
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["scaling"] == two["scaling"] == "strong"
    assert one["config"]["pool_images"] == two["config"]["pool_images"] == 128
    assert two["config"]["selected_sha1"] == one["config"]["selected_sha1"]
    assert two["config"]["n_selected"] == one["config"]["n_selected"] > 0
    r = two["rccl"]
    assert r["world_size"] == 2 and r["ranks_seen"] == 2
    assert sorted(x["rank"] for x in r["per_rank"]) == [0, 1]
    assert sum(x["images"] for x in r["per_rank"]) == 128
    import torch
    if torch.cuda.device_count() >= 2:
        assert r["backend"] == "nccl" and r["distinct_devices"] == 2 and not r["shared_gpu"]
        # with a GPU per rank the score rows travel through the C ABI's own collective (cald_allgather_scores), not torch.distributed
        assert r["cabi"]["world_size"] == 2 and "cald_allgather_scores" in r["collective"]
    else:
        assert r["backend"] == "gloo" and r["shared_gpu"] and r["cabi"] is None
    assert abs(two["ms_per_step"] * two["steps"] - 128 / two["value"] * 1e3) < 1e-3 * two["ms_per_step"] * two["steps"] + 1e-6


@pytest.mark.gpu
def test_bench_under_torch_distributed_run_weak_scaling():
    """The launcher form of the contract (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`) and the
    weak-scaling option: K x 64 images per rank."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    p, line, lines = _run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1",
                           "--scaling", "weak"] + FAST)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["pool_images"] == 128
    assert line["rccl"]["ranks_seen"] == 2 and [x["images"] for x in line["rccl"]["per_rank"]] == [64, 64]


@pytest.mark.gpu
def test_bench_cabi_collective_falls_back_together_when_a_rank_cannot_join():
    """`--comm cabi` on a box where RCCL cannot form the communicator (two ranks on ONE device: ncclCommInitRank returns 'invalid usage'):
    every step of the bootstrap is agreed on through the launcher's process group, so all ranks leave the C-ABI path TOGETHER, the rows
    travel through torch.distributed, rc is 0, there is ONE JSON line, it says what happened -- nobody is left alone in a collective."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box where RCCL refuses the communicator (one visible GPU)")
    p, line, lines = _run(["bench.py", "--gpus", "2", "--comm", "cabi", "--steps", "2", "--warmup", "1"] + FAST, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1 and line["n_gpus"] == 2
    r = line["rccl"]
    assert r["ranks_seen"] == 2 and r["backend"] == "gloo"
    assert "error" in r["cabi"] and "cald_comm_init_rank" in r["cabi"]["error"]
    assert line["config"]["n_selected"] > 0
