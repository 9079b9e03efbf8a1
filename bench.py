#!/usr/bin/env python3
"""Headline benchmark: Conformer-L encoder forward, audio(mel)-frames/sec on MI355X (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]            # starts its own N ranks when no launcher did
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
    python bench.py --train [--gpus N]                             # BASELINE cfg-3 / cfg-4: bf16 training step, B=64 per GPU, DDP

Without RANK/WORLD_SIZE in the environment and with --gpus N > 1 this process is only a PARENT: it parses the arguments,
starts N child processes of this same file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set
(as the reference's train.py:364-379 spawns its own ranks) and relays rank 0's JSON line; it never touches the GPU itself.

A "step" = one pass of the hot path (Encoder.forward: conv-subsampling stem -> input Linear -> 16 Conformer
blocks) over one synthetic batch that is already resident in HBM.  Utterances shard over the batch axis, so
N GPUs run N independent replicas with their own B=32 (weak scaling, no data-path collective).
Rank 0 prints ONE JSON line (see the driver contract in the task statement) carrying two extra objects:
  roofline      -- the dominant kernel (fp32 MFMA GEMM family) timed live with HIP events on the launch stream
  cpu_baseline  -- the CPU oracle (oracle/conformer_oracle.py, a restatement pinned against the reference)
                   timed on this host's cores on a bounded sample of the same workload (N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.json configs[1]: 16-layer d=512 Conformer-L, B=32 T=1000 synthetic, fp32 forward
CFG = dict(n_mel=80, n_blocks=16, d=512, n_heads=8, ksize=31, B=32, T=1000)
PEAK_MFMA_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_HBM_GBS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def algorithmic_flops(B, T, d, L, K, n_mel=80):
    """2*MAC of the encoder forward."""
    T1 = (T - 1) // 2; Tp = (T1 - 1) // 2
    F1 = (n_mel - 1) // 2; Fp = (F1 - 1) // 2
    N = B * Tp; P = 2 * Tp - 1
    front = 18 * d * F1 * T1 * B + 18 * d * d * Fp * Tp * B + 2 * (Fp * d) * d * N
    block = 46 * N * d * d + 2 * P * d * d + 6 * B * Tp * Tp * d + 2 * N * d * K
    return front + L * block


def time_events(fn, iters, warm=2, inner=10):
    """Mean launch duration in ms from HIP events on the launch stream (torch's current stream): `inner` back-to-back
    launches between two events, repeated `iters` times (avg over everything, median over the repeats).  Back to back because
    an event pair around EVERY launch also times the host-side gap of the Python call -- 5-10 % on the 40-150 us layer GEMMs
    (round-1 site table vs rocprofv3: 155 vs 145 us on the FFN-hidden GEMM)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    ts.sort()
    return sum(ts) / len(ts), ts[len(ts) // 2]


def gemm_site_table(enc, x, iters):
    """Times every distinct MFMA-GEMM launch of the forward in isolation (HIP events on the launch stream =
    torch's current stream) and returns rows: kernel instance, shape, launches/step, avg ms, TFLOP/s."""
    from conformer_amd import ops
    B, T, d, L = CFG["B"], CFG["T"], CFG["d"], CFG["n_blocks"]
    dev = x.device
    T1 = (T - 1) // 2; Tp = (T1 - 1) // 2; F1 = 39; Fp = 19
    N = B * Tp; P = 2 * Tp - 1
    lay = enc.layers[0]
    att = lay.attention.attention
    a_d = torch.randn(N, d, device=dev); a_4d = torch.randn(N, 4 * d, device=dev)
    res = torch.randn(N, d, device=dev)
    h2 = torch.randn(N, Fp * d, device=dev)
    pe = torch.randn(P, d, device=dev)
    stem = enc.downsampling_conv
    w2p = stem._packs.get("w2p", (stem.conv_2.weight,), lambda: ops.pack_conv2_weight(stem.conv_2.weight))
    h1 = torch.randn(B, T1, F1, d, device=dev).relu_()
    h2o = torch.empty(B, Tp, Fp * d, device=dev)
    from conformer_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream

    def conv2():
        _lib.check(lib.cfm_subsample_conv2_relu_f32(h1.data_ptr(), w2p.data_ptr(), stem.conv_2.bias.data_ptr(),
                                                    h2o.data_ptr(), B, F1, T1, d, st), "conv2")

    qkvw, qkvb = att._qkv_params()
    wl = enc.linear.weight
    fold = ops.ln_fold_ok(d)
    posw, posb = torch.cat([att.pos_proj.weight.detach()] * L, 0).contiguous(), torch.cat([att.pos_proj.bias.detach()] * L, 0).contiguous()
    if fold:
        # the folded-LayerNorm forms the forward really launches (ops.linear_lnfold consumers, emit_stats producers)
        _, stt = ops.linear_residual(a_d, att.out_proj.weight, att.out_proj.bias, res, 1.0, emit_stats=True)
        f1, fq, fc = (ops.fold_layernorm(w_, b_, ln.weight, ln.bias) for w_, b_, ln in (
            (lay.ffn_1.hidden_linear.weight, lay.ffn_1.hidden_linear.bias, lay.ffn_1.layer_norm), (qkvw, qkvb, lay.attention.layer_norm),
            (lay.conv.pointwise_conv_1.weight, lay.conv.pointwise_conv_1.bias, lay.conv.layer_norm)))
        sites = [
            ("gemm<relu,conv>  stem conv2 implicit GEMM", (B * Tp * Fp, d, 9 * d), 1, conv2),
            ("gemm<bias+stats> input linear", (N, d, Fp * d), 1, lambda: ops.linear(h2, wl, enc.linear.bias, emit_stats=True)),
        ]
        if ops.ffn_fused_ok(d, 4 * d, N):
            # the one-kernel feed-forward sub-layers (csrc/ffn_fused_f32.hip): (M, N, K) = (rows, 2 x 4d, d) carries the FLOPs of both products
            wp, bf_, cs_ = lay.ffn_1._pack()
            ln_b = lay.layer_norm
            sites += [
                ("ffn_fused<stats>  FFN-1: LN fold + Linear + Swish + Linear + x/2 residual", (N, 8 * d, d), L,
                 lambda: ops.ffn_fused(a_d, stt, wp, bf_, cs_, lay.ffn_1.out_linear.bias, 0.5, 1e-5, emit_stats=True)),
                ("ffn_fused<LN>     FFN-2 + the block's closing LayerNorm", (N, 8 * d, d), L,
                 lambda: ops.ffn_fused(a_d, stt, wp, bf_, cs_, lay.ffn_1.out_linear.bias, 0.5, 1e-5, emit_stats=True,
                                       closing_ln=(ln_b.weight, ln_b.bias, ln_b.eps))),
            ]
        else:
            sites += [
                ("gemm<LN,swish>   FFN hidden (LayerNorm folded)", (N, 4 * d, d), 2 * L, lambda: ops.linear_lnfold(a_d, stt, *f1, 1e-5, act="swish")),
                ("gemm<resid+stats> FFN-1 out", (N, d, 4 * d), L,
                 lambda: ops.linear_residual(a_4d, lay.ffn_1.out_linear.weight, lay.ffn_1.out_linear.bias, res, 0.5, emit_stats=True)),
                ("gemm<residual>   FFN-2 out", (N, d, 4 * d), L,
                 lambda: ops.linear_residual(a_4d, lay.ffn_1.out_linear.weight, lay.ffn_1.out_linear.bias, res, 0.5)),
            ]
        sites += [
            ("gemm<LN,bias>    fused QKV (LayerNorm folded)", (N, 3 * d, d), L, lambda: ops.linear_lnfold(a_d, stt, *fq, 1e-5)),
            ("gemm<bias>       pos proj (all layers, one launch)", (P, L * d, d), 1, lambda: ops.linear(pe, posw, posb)),
            ("gemm<resid+stats> attn out / pw2", (N, d, d), 2 * L,
             lambda: ops.linear_residual(a_d, att.out_proj.weight, att.out_proj.bias, res, 1.0, emit_stats=True)),
            ("gemm<LN,glu>     pw1+GLU (LayerNorm folded)", (N, 2 * d, d), L, lambda: ops.linear_lnfold(a_d, stt, *fc, 1e-5, glu=True)),
        ]
    else:
      sites = [
        ("gemm<relu,conv>  stem conv2 implicit GEMM", (B * Tp * Fp, d, 9 * d), 1, conv2),   # pmc: gemm_f32_kernel<128, 128, 2, true>
        ("gemm<bias>       input linear", (N, d, Fp * d), 1, lambda: ops.linear(h2, wl, enc.linear.bias)),
        ("gemm<swish>      FFN hidden", (N, 4 * d, d), 2 * L,
         lambda: ops.linear(a_d, lay.ffn_1.hidden_linear.weight, lay.ffn_1.hidden_linear.bias, act="swish")),
        ("gemm<residual>   FFN out", (N, d, 4 * d), 2 * L,
         lambda: ops.linear_residual(a_4d, lay.ffn_1.out_linear.weight, lay.ffn_1.out_linear.bias, res, 0.5)),
        ("gemm<bias>       fused QKV", (N, 3 * d, d), L, lambda: ops.linear(a_d, qkvw, qkvb)),
        ("gemm<bias>       pos proj", (P, d, d), L, lambda: ops.linear(pe, att.pos_proj.weight, att.pos_proj.bias)),
        ("gemm<residual>   attn out / pw2", (N, d, d), 2 * L,
         lambda: ops.linear_residual(a_d, att.out_proj.weight, att.out_proj.bias, res, 1.0)),
        ("gemm<glu>        pw1+GLU", (N, 2 * d, d), L,
         lambda: ops.linear_glu(a_d, lay.conv.pointwise_conv_1.weight, lay.conv.pointwise_conv_1.bias)),
      ]
    rows = []
    for name, (m, n, k), per_step, fn in sites:
        avg, med = time_events(fn, max(3, iters // 3), inner=2 if "conv2" in name else 10)
        log(f"[bench] {name}: {m}x{n}x{k} avg {avg:.3f} ms  {2.0 * m * n * k / avg / 1e9:.1f} TFLOP/s")
        fl = 2.0 * m * n * k
        row = dict(kernel=name, M=m, N=n, K=k, launches_per_step=per_step, avg_ms=avg, med_ms=med,
                   tflops=fl / (avg * 1e-3) / 1e12, flops=fl)
        if "conv2" in name:
            row["pmc_name"] = "gemm_f32_kernel<128, 128, 2, true"
            # h1 read once + packed weight + h2 written (the stem's ideal traffic)
            row["alg_bytes"] = 4.0 * (B * T1 * F1 * d + 9 * d * d + B * Tp * Fp * d)
        rows.append(row)
    return rows


def host_cores() -> int:
    """CPU cores this process may really use: affinity mask, capped by the cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(enc, sample_b):
    """The oracle on the host cores, same model weights, same T, a bounded batch sample."""
    from oracle import conformer_oracle as O
    P = {"encoder." + k: v.detach().cpu() for k, v in enc.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    x = torch.randn(sample_b, CFG["n_mel"], CFG["T"], generator=g)
    L = torch.full((sample_b,), CFG["T"], dtype=torch.int64)
    cores = min(host_cores(), 32)
    torch.set_num_threads(cores)
    log(f"[bench] cpu_baseline: oracle on {cores} threads (os.cpu_count()={os.cpu_count()}), B={sample_b}")
    ts = []
    with torch.no_grad():
        for i in range(4):                                  # 1 warm-up + 3 timed runs, median
            t0 = time.perf_counter()
            O.encoder_forward(x, L, P, CFG["n_blocks"], CFG["n_heads"])
            ts.append(time.perf_counter() - t0)
    t = sorted(ts[1:])[1]
    return dict(value=sample_b * CFG["T"] / t, unit="audio-frames/sec", cores=cores, kind="port",
                sample=f"oracle Encoder.forward fp32 on the same weights, B={sample_b} of 32, T=1000, 16 blocks, median of 3 after 1 warm-up "
                       f"({t:.2f} s/run; runs {', '.join(f'{v:.2f}' for v in ts[1:])} s; torch {torch.get_num_threads()} threads)")


def pmc_traffic_bytes(kernel_substr: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    collected in separate --pmc runs of this same command, corrected as MI355X_MICROARCH.md prescribes: KiB units,
    FETCH_SIZE doubled on gfx950) -- tools/pmc_traffic.py writes the file; None when it is absent."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))   # the latest round's passes
    try:
        table = json.load(open(paths[-1]))
    except (OSError, ValueError, IndexError):
        return None
    for name, v in table.items():
        if kernel_substr in name:
            return v["hbm_bytes_per_launch"]
    return None


def pmc_mfma_row(kernel_substr: str):
    """Matrix-pipe utilisation of a kernel from the committed counter pass of the latest round (tools/pmc_mfma.py ->
    profiles/rNN_pmc_mfma.json: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) and the wave-cycle split)."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_mfma.json")))
    try:
        table = json.load(open(paths[-1]))
    except (OSError, ValueError, IndexError):
        return None
    for name, v in table.items():
        if kernel_substr in name:
            return {"mfma_util": round(v["mfma_util"], 4), "effective_clock_ghz": round(v.get("effective_clock_ghz", 0.0), 3),
                    "avg_us_under_counters": round(v.get("avg_us", 0.0), 1),
                    "wave_cycle_split": {k: round(x, 3) for k, x in v.get("wave_cycle_split", {}).items()}, "source": os.path.basename(paths[-1])}
    return None


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n: int, argv) -> int:
    """Parent side of `bench.py --gpus N` when no launcher set RANK: one child per GPU, rendezvous on 127.0.0.1.
    Rank 0 inherits stdout (its single JSON line is the result); the other ranks' stdout goes to stderr.  The first
    failing child ends the job: the others are terminated by PID and its exit code is returned."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    log(f"[bench] parent {os.getpid()}: started {n} ranks (pids {[p.pid for p in procs]}), rendezvous 127.0.0.1:{port}")
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                log(f"[bench] rank process {p.pid} exited with {code}; stopping the other ranks")
                for q in alive:
                    q.terminate()
    return rc


def selftest_cpu(args, env):
    """Plumbing only (no kernels, no GPU): the ranks this file started rendezvous over gloo and run the timing contract
    on a trivial CPU step.  Exists so the N>1 launch path is exercised by `-m "not gpu"` tests."""
    from conformer_amd import parallel
    import torch.distributed as dist
    started = parallel.init_distributed(env, torch.device("cpu"), backend="gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 8))
    ddp = parallel.wrap_ddp(net, None, args.bucket_cap_mb, args.gradient_as_bucket_view, args.static_graph)
    probe = parallel.DdpCommProbe(ddp)
    a = torch.ones(16, 64) * (1 + env.rank)
    step_ms = []

    def step():
        t0 = time.perf_counter()
        probe.begin_step()
        net.zero_grad(set_to_none=True)
        ddp(a).pow(2).mean().backward()
        step_ms.append((time.perf_counter() - t0) * 1e3)

    dt = parallel.timed_steps(step, args.steps, args.warmup, lambda: None)
    comm = probe.summary(skip=args.warmup)
    per_rank = parallel.gather_to_rank0(sum(step_ms[args.warmup:]) / max(1, args.steps))
    comms = parallel.gather_to_rank0(comm)
    ranks = dist.get_world_size() if started else 1
    seen = torch.zeros(ranks, dtype=torch.int64)
    seen[env.rank] = 1 + env.local_rank
    if started:
        dist.all_reduce(seen)
    if env.rank == 0:
        print(json.dumps({"metric": "plumbing-selftest", "value": 0.0, "unit": "none", "n_gpus": env.world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "plumbing-selftest (no kernels)",
                          "ranks": ranks, "collective_backend": "gloo" if started else None,
                          "local_ranks_plus_one": seen.tolist(), "config": {"workload": "none"},
                          **comm_fields(comms, per_rank, args)}), flush=True)
    if started:
        dist.barrier()
        dist.destroy_process_group()


def comm_fields(comms, per_rank_step_ms, args):
    """The data-parallel part of the --train JSON line: what the gradient exchange cost (DdpCommProbe; worst rank), what every
    rank's step took, and DDP's settings -- so that the first multi-GPU run explains itself."""
    vals = [c["exposed_comm_ms"] for c in comms if c.get("exposed_comm_ms") is not None]
    spans = [c["comm_span_ms"] for c in comms if c.get("comm_span_ms") is not None]
    return {"exposed_comm_ms": max(vals) if vals else None, "comm_span_ms": max(spans) if spans else None,
            "exposed_comm_ms_per_rank": [c.get("exposed_comm_ms") for c in comms],
            "allreduce_bytes_per_step": comms[0]["allreduce_bytes_per_step"], "allreduce_buckets": comms[0]["allreduce_buckets"],
            "per_rank_step_ms": [round(float(v), 4) for v in per_rank_step_ms],
            "ddp": {"bucket_cap_mb": args.bucket_cap_mb, "gradient_as_bucket_view": bool(args.gradient_as_bucket_view),
                    "static_graph": bool(args.static_graph)}}


def train_bench(args, env, dev, dist):
    """BASELINE cfg-3 / cfg-4: one optimiser step of the reference's training loop (train.py:225-243) per "step":
    Conformer-L under bf16 autocast -> CTC on fp32 logits -> backward (DDP all-reduce of the gradients over RCCL,
    overlapped with the backward by the bucket hooks) -> Adam.  B=64 utterances per GPU, T=1000 mel frames, 40 target
    tokens, train-mode BatchNorm, dropout 0.1 (the reference's default), synthetic data resident in HBM."""
    from conformer_amd import parallel
    from conformer_amd.evaluation import ConformerCriterion
    from conformer_amd.optim import FusedAdam
    from model.conformer import Conformer
    B, T = 64, CFG["T"]
    torch.manual_seed(0)
    model = Conformer(370, CFG["n_mel"], CFG["n_blocks"], CFG["d"], CFG["n_heads"], CFG["ksize"], 640, 1, args.dropout).to(dev).train()
    ddp = parallel.wrap_ddp(model, dev, args.bucket_cap_mb, args.gradient_as_bucket_view, args.static_graph)
    probe = parallel.DdpCommProbe(ddp)       # the default exchange (divide + all-reduce per bucket), time-stamped
    opt = FusedAdam(model.parameters(), lr=2e-5)
    g = torch.Generator().manual_seed(100 + env.rank)
    x = torch.randn(B, CFG["n_mel"], T, generator=g).to(dev)
    lengths = torch.full((B,), T, dtype=torch.int64, device=dev)
    targets = torch.randint(1, 370, (B, 40), generator=g).to(dev)
    tlen = torch.full((B,), 40, dtype=torch.int64, device=dev)
    crit = ConformerCriterion(blank_id=0)
    last = {}
    marks = []                                # one HIP event per step start (+ one at the end): per-rank step times, no syncs

    def step():
        marks.append(torch.cuda.Event(enable_timing=True))
        marks[-1].record()
        probe.begin_step()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits, out_len = ddp(x, lengths)
            with torch.autocast("cuda", enabled=False):
                loss = crit.ctc_loss(logits, targets, out_len, tlen)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        last["loss"] = loss

    dt = parallel.timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize, dev)
    marks.append(torch.cuda.Event(enable_timing=True))
    marks[-1].record()
    torch.cuda.synchronize()
    mine = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.warmup, args.warmup + args.steps)]
    per_rank = parallel.gather_to_rank0(sum(mine) / len(mine))
    comms = parallel.gather_to_rank0(probe.summary(skip=args.warmup))
    loss = float(last["loss"])
    if not loss == loss:
        raise SystemExit("non-finite training loss")
    if env.rank == 0:
        ms = dt / args.steps * 1e3
        fwd = algorithmic_flops(B, T, CFG["d"], CFG["n_blocks"], CFG["ksize"])
        print(json.dumps({
            "metric": "training audio-frames/sec (Conformer-L bf16 autocast fwd+CTC+bwd+Adam, B=64/GPU, T=1000)",
            "value": env.world * B * T * args.steps / dt, "unit": "audio-frames/sec", "n_gpus": env.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 MFMA operands, f32 accumulate / master weights / gradients (autocast)", "data": "synthetic",
            "ranks": env.world, "collective_backend": (dist.get_backend() + " (RCCL)") if dist else None,
            "config": {"workload": "cfg3/cfg4 Conformer-L training step: per-GPU B=64, T=1000 mel frames, 40 target tokens, "
                                   f"dropout {args.dropout}, BatchNorm train, CTC fp32, FusedAdam, DDP gradient all-reduce",
                       "per_gpu_batch": B, "global_batch": B * env.world, "mel_frames": T},
            **comm_fields(comms, per_rank, args),
            "encoder_fwd_bwd_tflops": 3 * fwd / (ms * 1e-3) / 1e12, "loss": loss,
            "max_mem_gib": torch.cuda.max_memory_allocated() / 2 ** 30}), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--train", action="store_true",
                    help="BASELINE cfg-3 (1 GPU) / cfg-4 (N GPUs): Conformer-L training step under bf16 autocast, B=64 per GPU, "
                         "forward + CTC + backward + Adam, gradients all-reduced by DDP over RCCL")
    ap.add_argument("--dropout", type=float, default=0.1, help="--train only (train.py default 0.1)")
    ap.add_argument("--bucket-cap-mb", type=int, default=25, help="--train: DDP bucket size (train.py:186 uses the default 25)")
    ap.add_argument("--gradient-as-bucket-view", action="store_true", help="--train: DDP gradient_as_bucket_view")
    ap.add_argument("--static-graph", action="store_true", help="--train: DDP static_graph")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="plumbing test only (tests/test_parallel_cpu.py): ranks rendezvous over gloo on the CPU and time a "
                         "no-kernel step; the JSON line says so and carries value 0")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the ~260 kernels eagerly instead of one hipGraph replay")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the extra (never headline) timings of the opt-in modes appended under \"secondary\"")
    ap.add_argument("--dtype", choices=["f32", "bf16", "f32x6", "f32x3"], default="f32",
                    help="f32 = the headline configuration (BASELINE cfg-2, native fp32 MFMA); bf16 = torch.autocast(bfloat16): "
                         "dense GEMMs on the bf16 matrix pipe with fp32 accumulate and fp32 tensors; f32x6 / f32x3 = fp32 GEMMs "
                         "computed from exact bf16 expansions of the fp32 operands (ops.set_fp32_matmul: six / three bf16 "
                         "MFMAs per K-step, fp32-class / 2^-15 error).  All but f32 are secondary results, never the default")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))          # parent only: nothing below runs in this process

    from conformer_amd import parallel
    env = parallel.env_from_os()
    world, rank, local = env.world, env.rank, env.local_rank
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; start it as "
                         f"`python bench.py --gpus {args.gpus}` (it spawns its own ranks) or as `python -m torch.distributed.run "
                         f"--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus}`")
    if args.selftest_cpu:
        return selftest_cpu(args, env)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the Conformer hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if parallel.init_distributed(env, dev):                 # backend "nccl" = RCCL on ROCm
        import torch.distributed as dist_
        dist = dist_
    if args.train:
        return train_bench(args, env, dev, dist)

    from conformer_amd import _lib, ops
    _lib.check(_lib.load().cfm_device_check(), "cfm_device_check")
    from model.modules.encoder import Encoder
    if args.dtype in ("f32x6", "f32x3"):
        ops.set_fp32_matmul("bf16x6" if args.dtype == "f32x6" else "bf16x3")

    torch.manual_seed(0)
    enc = Encoder(CFG["n_mel"], CFG["n_blocks"], CFG["d"], CFG["n_heads"], CFG["ksize"], 0.0).to(dev).eval()
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(CFG["B"], CFG["n_mel"], CFG["T"], generator=g).to(dev)          # resident in HBM before timing
    lengths = torch.full((CFG["B"],), CFG["T"], dtype=torch.int64, device=dev)

    def step():
        with torch.no_grad():
            return enc(x, lengths)

    log(f"[bench] rank {rank}/{world}: model on {dev}, warm-up {args.warmup} steps")
    last = {}
    runner, graphed = enc, False
    if not args.no_graph:
        try:
            from conformer_amd.graph import GraphedEncoder
            runner, graphed = GraphedEncoder(enc, x, lengths, autocast_dtype=torch.bfloat16 if args.dtype == "bf16" else None), True
        except Exception as e:                               # capture is an optimisation: fall back to eager launches
            log(f"[bench] hipGraph capture unavailable ({type(e).__name__}: {e}); eager launches")

    import contextlib
    amp = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if args.dtype == "bf16" else contextlib.nullcontext

    def step():
        with torch.no_grad(), amp():
            last["y"], _ = runner(x, lengths)

    dt = parallel.timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize, dev)
    y = last["y"]
    if not torch.isfinite(y).all():
        raise SystemExit("non-finite encoder output")
    n0 = _lib.CALLS[0]
    with torch.no_grad(), amp():
        enc(x, lengths)                                        # one eager forward, untimed: how many kernels the path launches
    torch.cuda.synchronize()
    launches = _lib.CALLS[0] - n0

    frames = world * CFG["B"] * CFG["T"] * args.steps
    ms = dt / args.steps * 1e3
    log(f"[bench] {args.steps} steps in {dt:.3f} s -> {ms:.2f} ms/step, {frames / dt:,.0f} frames/s")
    flops = algorithmic_flops(CFG["B"], CFG["T"], CFG["d"], CFG["n_blocks"], CFG["ksize"])
    out = {
        "metric": "encoder audio-frames/sec (B=32,T=1000,d=512,L=16)",
        "value": frames / dt, "unit": "audio-frames/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f32": "f32", "bf16": "bf16 MFMA operands, f32 accumulate/storage (autocast)",
                  "f32x6": "f32 operands split exactly into 3 bf16 terms, 6 bf16 MFMAs per K-step, f32 accumulate/storage",
                  "f32x3": "f32 operands split into 2 bf16 terms, 3 bf16 MFMAs per K-step, f32 accumulate/storage"}[args.dtype],
        "data": "synthetic", "launch": "hipGraph replay" if graphed else "eager",
        "ranks": world, "collective_backend": (dist.get_backend() + " (RCCL)") if dist else None,
        "config": {"workload": "cfg2 Conformer-L Encoder.forward: B=32/GPU, T=1000 mel frames (T'=249), d=512, 16 blocks, "
                               "8 heads, k=31, fp32, eval, random-init weights, batch-sharded replicas",
                   "per_gpu_batch": CFG["B"], "mel_frames": CFG["T"]},
        "kernel_launches_per_forward": launches, "layernorm_folded_into_gemms": bool(ops.ln_fold_ok(CFG["d"])) and args.dtype == "f32",
        "path_tflops": flops / (ms * 1e-3) / 1e12,
        "path_frac_of_mfma_f32_peak": flops / (ms * 1e-3) / 1e12 / PEAK_MFMA_F32_TFLOPS,
    }

    if rank == 0 and not args.no_roofline and args.dtype == "f32":
        rows = gemm_site_table(enc, x, iters=10)
        tot = sum(r["avg_ms"] * r["launches_per_step"] for r in rows)
        dom = max(rows, key=lambda r: r["avg_ms"] * r["launches_per_step"])
        out["roofline"] = {
            "bound": "mfma", "kernel": dom["kernel"], "shape_MNK": [dom["M"], dom["N"], dom["K"]],
            "achieved": dom["tflops"], "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
            "frac": dom["tflops"] / PEAK_MFMA_F32_TFLOPS,
            "traffic": pmc_traffic_bytes(dom["pmc_name"]) if dom.get("pmc_name") else None,
            "traffic_unit": "HBM bytes per launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, latest profiles/rNN_pmc_traffic.json)",
            "algorithmic_bytes": dom.get("alg_bytes"),
            "mfma_util": (pmc_mfma_row(dom["pmc_name"]) or {}).get("mfma_util") if dom.get("pmc_name") else None,
            "pmc": {"dominant": pmc_mfma_row(dom["pmc_name"]) if dom.get("pmc_name") else None,
                    "attention_forward": pmc_mfma_row("relpos_attn_fwd8p_kernel") or pmc_mfma_row("relpos_attn_fwd_kernel"),
                    "ffn_fused": pmc_mfma_row("ffn_fused_f32_kernel<16, 1") or pmc_mfma_row("gemm_f32_kernel<128, 64, 1, false")},
            "avg_launch_ms": dom["avg_ms"], "launches_per_step": dom["launches_per_step"],
            "gemm_ms_per_step": tot, "gemm_share_of_step": tot / ms,
            "all_gemm_sites": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()
                                if k not in ("flops", "pmc_name", "alg_bytes")}
                               for r in rows],
        }
    if rank == 0 and world == 1 and args.dtype == "f32" and not args.no_secondary:
        # The same forward in the opt-in modes, for the record only: `value` above is the native-fp32-MFMA path.
        sec = {}
        for name, mode, autocast in (("f32x6", "bf16x6", False), ("f32x3", "bf16x3", False), ("bf16_autocast", "native", True)):
            try:
                ops.set_fp32_matmul(mode)
                run2 = enc
                if not args.no_graph:
                    from conformer_amd.graph import GraphedEncoder
                    run2 = GraphedEncoder(enc, x, lengths, autocast_dtype=torch.bfloat16 if autocast else None)
                amp2 = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if autocast else contextlib.nullcontext
                res = {}

                def step2():
                    with torch.no_grad(), amp2():
                        res["y"], _ = run2(x, lengths)

                dt2 = parallel.timed_steps(step2, args.steps, args.warmup, torch.cuda.synchronize, dev)
                sec[name] = {"ms_per_step": dt2 / args.steps * 1e3, "value": CFG["B"] * CFG["T"] * args.steps / dt2,
                             "rel_l2_vs_headline_output": float((res["y"].float() - y).norm() / y.norm())}
            finally:
                ops.set_fp32_matmul("native")
        out["secondary"] = {"note": "opt-in modes, same workload, never the headline: f32x6 / f32x3 = fp32 GEMMs from exact "
                                    "bf16 operand expansions (6 / 3 bf16 MFMAs per K-step, fp32 accumulate, fp32 tensors); "
                                    "bf16_autocast = torch.autocast(bfloat16)", **sec}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(enc, sample_b=32)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
