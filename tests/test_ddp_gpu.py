"""Data-parallel training step on the GPU box: 2 ranks (both on cuda:0, gloo carrying the gradient all-reduce, because
the box has ONE GPU and RCCL needs one device per rank) wrap the Conformer in DistributedDataParallel; the averaged
gradients must equal the single-process gradients of the same global batch.  This exercises the autograd hooks /
bucketed all-reduce path of conformer_amd.parallel.wrap_ddp around the explicit HIP backward kernels."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(dev):
    from model.conformer import Conformer
    from oracle import conformer_oracle as O
    cfg = dict(vocab=17, n_mel=80, n_blocks=2, d=32, n_heads=4, ksize=31, lstm_hidden=24, seed=31)
    m = Conformer(17, 80, 2, 32, 4, 31, 24, 1, 0.0)
    m.load_state_dict(O.make_params(**cfg), strict=True)
    m = m.to(dev).train()                        # MIOpen's LSTM backward needs training mode ...
    for mod in m.modules():                      # ... but BatchNorm stays on running statistics: rank-independent math
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.eval()
    return m


def _data():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 80, 103, generator=g)
    L = torch.tensor([103, 103, 103, 103])        # equal lengths so both shards keep lengths.max() == T'
    w = torch.randn(4, 25, 17, generator=g)      # T=103 -> T'=25
    return x, L, w


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch.distributed as dist
    from conformer_amd import parallel
    dev = torch.device("cuda:0")
    parallel.init_distributed(parallel.env_from_os(), dev, backend="gloo")
    try:
        model = _build(dev)
        ddp = parallel.wrap_ddp(model, dev)
        x, L, w = _data()
        lo, hi = parallel.shard_range(4, rank, world)
        logits, _ = ddp(x[lo:hi].to(dev), L[lo:hi].to(dev))
        (logits * w[lo:hi].to(dev)).sum().div(hi - lo).backward()
        grads = {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters() if p.grad is not None}
        q.put((rank, grads))                      # numpy: pickled by value (no fd passing after the worker exits)
    finally:
        dist.destroy_process_group()


def test_ddp_two_ranks_match_single_process():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    sys.path.insert(0, ROOT)
    model = _build(dev)
    x, L, w = _data()
    logits, _ = model(x.to(dev), L.to(dev))
    (logits * w.to(dev)).sum().div(4).backward()
    ref = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert set(out[0]) == set(ref) and len(ref) > 80
    for n, g in ref.items():
        for r in range(world):
            o = torch.from_numpy(out[r][n])
            err = float((o - g).norm() / (g.norm() + 1e-12))
            if float(g.norm()) < 1e-4:
                assert float((o - g).abs().max()) < 1e-4, n
            else:
                assert err < 2e-4, (n, r, err)
