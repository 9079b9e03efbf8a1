#!/usr/bin/env python3
"""Yard-stick, not product: the encoder forward of BASELINE cfg-2 expressed with STOCK PyTorch-ROCm operators (rocBLAS /
hipBLASLt GEMMs, MIOpen convolutions, ATen elementwise kernels) in the operator order SURVEY.md section 8(a) records for the
reference (channel-first conv module with its two transposes, batch-repeated positional projection, full-width positional
scores + pad/view relative shift, masked_fill + softmax, materialised (B,C,F,T) stem activations).  It answers "what would
the reference's own code path reach on this MI355X" -- the number the gfx950 path has to beat -- without the reference's
sources travelling to the GPU box.  Random weights, synthetic input, eval, no_grad.

    python tools/stock_torch_encoder.py [--dtype f32|bf16] [--batch 32] [--frames 1000] [--steps 5]
    python tools/stock_torch_encoder.py --train --dtype bf16 --batch 64     # + nn.LSTM decoder, CTC, backward, Adam (cfg-3)
"""
import argparse
import json
import math
import time

import torch
import torch.nn.functional as F


def make_weights(d, layers, n_mel, ksize, dev):
    g = torch.Generator().manual_seed(0)

    def w(*shape, fan):
        return (torch.randn(*shape, generator=g) / math.sqrt(fan)).to(dev)

    fsub = ((n_mel - 1) // 2 - 1) // 2
    W = {"c1w": w(d, 1, 3, 3, fan=9), "c1b": w(d, fan=9), "c2w": w(d, d, 3, 3, fan=9 * d), "c2b": w(d, fan=9 * d),
         "inw": w(d, d * fsub, fan=d * fsub), "inb": w(d, fan=d), "layers": []}
    for _ in range(layers):
        lay = {}
        for ffn in ("f1", "f2"):
            lay[ffn] = {"lnw": torch.ones(d, device=dev), "lnb": torch.zeros(d, device=dev), "w1": w(4 * d, d, fan=d),
                        "b1": w(4 * d, fan=d), "w2": w(d, 4 * d, fan=4 * d), "b2": w(d, fan=4 * d)}
        lay["att"] = {"lnw": torch.ones(d, device=dev), "lnb": torch.zeros(d, device=dev),
                      **{n: w(d, d, fan=d) for n in ("wq", "wk", "wv", "wp", "wo")},
                      **{n: w(d, fan=d) for n in ("bq", "bk", "bv", "bp", "bo")}}
        lay["conv"] = {"lnw": torch.ones(d, device=dev), "lnb": torch.zeros(d, device=dev), "pw1": w(2 * d, d, 1, fan=d),
                       "pb1": w(2 * d, fan=d), "dw": w(d, 1, ksize, fan=ksize), "db": w(d, fan=ksize),
                       "bnw": torch.ones(d, device=dev), "bnb": torch.zeros(d, device=dev), "bnm": torch.zeros(d, device=dev),
                       "bnv": torch.ones(d, device=dev), "pw2": w(d, d, 1, fan=d), "pb2": w(d, fan=d)}
        lay["lnw"], lay["lnb"] = torch.ones(d, device=dev), torch.zeros(d, device=dev)
        W["layers"].append(lay)
    return W


def feed_forward(x, p):
    h = F.layer_norm(x, x.shape[-1:], p["lnw"], p["lnb"])
    h = F.linear(h, p["w1"], p["b1"])
    h = h * torch.sigmoid(h)
    return F.linear(h, p["w2"], p["b2"])


def rel_shift(s):
    b, h, t, p = s.shape
    s = torch.cat([s.new_zeros(b, h, t, 1), s], dim=-1).view(b, h, p + 1, t)[:, :, 1:].reshape(b, h, t, p)
    return s[..., :p // 2 + 1]


def self_attention(x, pos, pad_mask, p, heads, ubias, vbias):
    b, t, d = x.shape
    dh = d // heads
    h = F.layer_norm(x, (d,), p["lnw"], p["lnb"])
    q = F.linear(h, p["wq"], p["bq"]).view(b, t, heads, dh)
    k = F.linear(h, p["wk"], p["bk"]).view(b, t, heads, dh).permute(0, 2, 1, 3)
    v = F.linear(h, p["wv"], p["bv"]).view(b, t, heads, dh).permute(0, 2, 1, 3)
    pe = F.linear(pos, p["wp"], p["bp"]).view(b, -1, heads, dh).permute(0, 2, 3, 1)      # batch-repeated, as written
    content = torch.matmul((q + ubias).transpose(1, 2), k.transpose(2, 3))
    position = rel_shift(torch.matmul((q + vbias).transpose(1, 2), pe))
    score = (content + position) / math.sqrt(dh)
    score = score.masked_fill(pad_mask, torch.finfo(score.dtype).min)
    ctx = torch.matmul(torch.softmax(score, dim=-1), v).transpose(1, 2).reshape(b, t, d)
    return F.linear(ctx, p["wo"], p["bo"])


def conv_module(x, p, ksize, train=False):
    h = F.layer_norm(x, x.shape[-1:], p["lnw"], p["lnb"]).transpose(1, 2)
    h = F.glu(F.conv1d(h, p["pw1"], p["pb1"]), dim=1)
    h = F.conv1d(h, p["dw"], p["db"], padding=(ksize - 1) // 2, groups=h.shape[1])
    h = F.batch_norm(h, p["bnm"], p["bnv"], p["bnw"], p["bnb"], training=train)
    h = h * torch.sigmoid(h)
    return F.conv1d(h, p["pw2"], p["pb2"]).transpose(1, 2)


def encoder(x, lengths, W, heads, ksize, ubias, vbias, train=False):
    h = F.relu(F.conv2d(x.unsqueeze(1), W["c1w"], W["c1b"], stride=2))
    h = F.relu(F.conv2d(h, W["c2w"], W["c2b"], stride=2))
    b, c, f, t = h.shape
    h = F.linear(h.permute(0, 3, 1, 2).reshape(b, t, c * f), W["inw"], W["inb"])
    out_len = ((lengths - 1) // 2 - 1) // 2
    pad_mask = ~(out_len[:, None] > torch.arange(t, device=x.device))[:, None, None, :]
    d = h.shape[-1]
    rel = torch.arange(t - 1, -t, -1.0, device=x.device)[:, None]
    freq = torch.exp(torch.arange(0, d, 2.0, device=x.device) * (-math.log(10000.0) / d))
    pe = torch.zeros(2 * t - 1, d, device=x.device)
    pe[:, 0::2], pe[:, 1::2] = torch.sin(rel * freq), torch.cos(rel * freq)
    pos = pe.unsqueeze(0).repeat(b, 1, 1)
    for lay in W["layers"]:
        h = h + 0.5 * feed_forward(h, lay["f1"])
        h = h + self_attention(h, pos, pad_mask, lay["att"], heads, ubias, vbias)
        h = h + conv_module(h, lay["conv"], ksize, train)
        h = h + 0.5 * feed_forward(h, lay["f2"])
        h = F.layer_norm(h, (d,), lay["lnw"], lay["lnb"])
    return h, out_len


def leaves(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from leaves(v)
    elif isinstance(obj, list):
        for v in obj:
            yield from leaves(v)


def train_bench(args, dev, W, ubias, vbias, x, lengths, heads, ksize):
    """Full step as train.py:225-243 runs it: forward under autocast, CTC on fp32 log-softmax, backward, Adam."""
    d, hidden, vocab = 512, 640, 370
    lstm = torch.nn.LSTM(d, hidden, 1, batch_first=True).to(dev)
    bn = torch.nn.BatchNorm1d(hidden).to(dev)
    out = torch.nn.Linear(hidden, vocab).to(dev)
    buffers = {id(t) for lay in W["layers"] for t in (lay["conv"]["bnm"], lay["conv"]["bnv"])}
    params = [t.requires_grad_() for t in leaves(W) if id(t) not in buffers] + [ubias.requires_grad_(), vbias.requires_grad_()]
    params += list(lstm.parameters()) + list(bn.parameters()) + list(out.parameters())
    opt = torch.optim.Adam(params, lr=2e-5)
    targets = torch.randint(1, vocab, (args.batch, 40), device=dev)
    tlen = torch.full((args.batch,), 40, dtype=torch.int64, device=dev)
    ctc = torch.nn.CTCLoss(blank=0, zero_infinity=True)
    amp = torch.bfloat16 if args.dtype == "bf16" else None

    def step():
        with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
            h, out_len = encoder(x, lengths, W, heads, ksize, ubias, vbias, train=True)
            pk = torch.nn.utils.rnn.pack_padded_sequence(h, out_len.cpu(), batch_first=True, enforce_sorted=True)
            y, _ = torch.nn.utils.rnn.pad_packed_sequence(lstm(pk)[0], batch_first=True)
            y = y * torch.sigmoid(y)
            y = bn(y.transpose(1, 2)).transpose(1, 2)
            logits = out(y)
        loss = ctc(logits.float().log_softmax(-1).transpose(0, 1), targets, out_len, tlen)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    print(json.dumps({"what": "training step (fwd + CTC + bwd + Adam) with stock PyTorch-ROCm operators (yard-stick)",
                      "dtype": args.dtype, "batch": args.batch, "mel_frames": args.frames, "ms_per_step": ms,
                      "frames_per_sec": args.batch * args.frames / (ms / 1e3), "loss": float(loss),
                      "params": sum(p.numel() for p in params), "torch": torch.__version__,
                      "max_mem_gib": torch.cuda.max_memory_allocated() / 2 ** 30}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", action="store_true")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    d, layers, heads, ksize = 512, 16, 8, 31
    W = make_weights(d, layers, 80, ksize, dev)
    g = torch.Generator().manual_seed(1)
    ubias = (torch.randn(heads, d // heads, generator=g) * 0.1).to(dev)
    vbias = (torch.randn(heads, d // heads, generator=g) * 0.1).to(dev)
    x = torch.randn(args.batch, 80, args.frames, generator=g).to(dev)
    lengths = torch.full((args.batch,), args.frames, dtype=torch.int64, device=dev)

    if args.train:
        return train_bench(args, dev, W, ubias, vbias, x, lengths, heads, ksize)

    def step():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.dtype == "bf16"):
            return encoder(x, lengths, W, heads, ksize, ubias, vbias)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y, _ = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = dt / args.steps * 1e3
    print(json.dumps({"what": "encoder forward with stock PyTorch-ROCm operators in the reference's operator order (yard-stick)",
                      "dtype": args.dtype, "batch": args.batch, "mel_frames": args.frames, "ms_per_step": ms,
                      "frames_per_sec": args.batch * args.frames / (ms / 1e3), "finite": bool(torch.isfinite(y).all()),
                      "torch": torch.__version__, "max_mem_gib": torch.cuda.max_memory_allocated() / 2 ** 30}))


if __name__ == "__main__":
    main()
