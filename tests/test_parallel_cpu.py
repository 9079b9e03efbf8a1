"""world_size-2 gloo tests (CPU) of the data-parallel host logic: batch sharding, the bench.py timing contract
(barrier + max over ranks), and DDP gradient averaging through conformer_amd.parallel.wrap_ddp."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from conformer_amd import parallel
    env = parallel.env_from_os()
    assert parallel.init_distributed(env, torch.device("cpu"))
    try:
        # 1. sharding: contiguous, disjoint, covering
        lo, hi = parallel.shard_range(7, rank, world)
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([hi - lo]))
        # 2. timing contract: rank 1 is slower; every rank must report the max
        import time
        calls = {"n": 0}

        def step():
            calls["n"] += 1
            time.sleep(0.02 * (rank + 1))

        dt = parallel.timed_steps(step, steps=3, warmup=2, sync=lambda: None)
        # 3. DDP gradient averaging == full-batch gradient of the mean loss
        torch.manual_seed(0)
        model = torch.nn.Linear(8, 4)
        ddp = parallel.wrap_ddp(model)
        xs = torch.arange(6 * 8, dtype=torch.float32).reshape(6, 8) / 10
        a, b = parallel.shard_range(6, rank, world)
        ddp(xs[a:b]).pow(2).mean().backward()
        ref = torch.nn.Linear(8, 4)
        ref.load_state_dict(model.state_dict())
        ref(xs).pow(2).mean().backward()
        gerr = float((model.weight.grad - ref.weight.grad).abs().max())
        q.put((rank, (lo, hi), [int(s) for s in sizes], calls["n"], dt, gerr))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_harness():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, sizes0, n0, dt0, g0), (r1, s1, sizes1, n1, dt1, g1) = res
    assert s0 == (0, 4) and s1 == (4, 7) and sizes0 == [4, 3] == sizes1
    assert n0 == n1 == 5                                   # 2 warm-up + exactly 3 timed steps
    assert abs(dt0 - dt1) < 1e-6 and dt0 >= 3 * 0.04 - 1e-3    # both report the slow rank's time
    assert g0 < 1e-6 and g1 < 1e-6


def test_shard_range_properties():
    from conformer_amd.parallel import shard_range
    for n in (0, 1, 7, 32, 512):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in parts) - min(h - l for l, h in parts) <= 1


def _run_bench(argv, env_extra=None, timeout=300):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r.returncode, [json.loads(l) for l in lines], r.stderr


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (WORLD_SIZE unset) must itself start two rank processes, rendezvous
    them on 127.0.0.1 and print exactly ONE JSON line from rank 0 (the driver's scale command; reference train.py:364-379).
    Plumbing only: --selftest-cpu swaps the kernels for a no-op step over gloo."""
    rc, js, err = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--selftest-cpu"])
    assert rc == 0, err
    assert len(js) == 1
    j = js[0]
    assert j["n_gpus"] == 2 and j["ranks"] == 2 and j["collective_backend"] == "gloo"
    assert j["local_ranks_plus_one"] == [1, 2]              # rank r ran with LOCAL_RANK r
    assert j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    # the data-parallel fields of the --train line, exercised over gloo by the selftest's DDP step
    assert len(j["per_rank_step_ms"]) == 2 and all(v > 0 for v in j["per_rank_step_ms"])
    assert j["allreduce_bytes_per_step"] == 4 * (64 * 64 + 64 + 8 * 64 + 8) and j["allreduce_buckets"] >= 1
    assert j["exposed_comm_ms"] is not None and j["exposed_comm_ms"] >= 0 and j["comm_span_ms"] >= j["exposed_comm_ms"] - 1e-6
    assert len(j["exposed_comm_ms_per_rank"]) == 2
    assert j["ddp"] == {"bucket_cap_mb": 25, "gradient_as_bucket_view": False, "static_graph": False}


def test_bench_ddp_knobs_reach_ddp():
    rc, js, err = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--selftest-cpu", "--bucket-cap-mb", "1",
                              "--gradient-as-bucket-view", "--static-graph"])
    assert rc == 0, err
    assert js[0]["ddp"] == {"bucket_cap_mb": 1, "gradient_as_bucket_view": True, "static_graph": True}
    assert js[0]["allreduce_bytes_per_step"] == 4 * (64 * 64 + 64 + 8 * 64 + 8)


def test_bench_under_a_launcher_and_mismatch():
    """Started by a launcher (RANK/WORLD_SIZE set) the file is a rank, not a parent; a --gpus that disagrees with
    WORLD_SIZE is an error, not a silently relabelled single-rank run (round-1 finding)."""
    rc, js, err = _run_bench(["--gpus", "1", "--selftest-cpu", "--steps", "2"], dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    assert rc == 0 and js[0]["n_gpus"] == 1 and js[0]["ranks"] == 1, err
    rc, js, err = _run_bench(["--gpus", "2", "--selftest-cpu"], dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    assert rc != 0 and not js and "WORLD_SIZE=1" in err


def test_bench_parent_propagates_a_failing_rank():
    """Without a GPU every rank of the real bench exits non-zero (no CPU fallback); the parent must report failure."""
    if torch.cuda.is_available():
        pytest.skip("needs a GPU-less host")
    rc, js, err = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert rc != 0 and not js
    assert "no CPU fallback" in err
