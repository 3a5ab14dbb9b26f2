#!/usr/bin/env python3
"""Throughput of encode and of one fused LBG pass over a range of (K, d, N) shapes, with the
ambiguous fraction; used to spot performance cliffs outside the benchmark shape."""
import json, os, sys, statistics
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantization_amd import HipEngine, train_lbg
from vector_quantization_amd._lib import lib
dev = torch.device("cuda:0")
eng = HipEngine(dev)

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in ev)

shapes = [(16, 12, 1 << 22), (256, 16, 1 << 22), (1024, 16, 1 << 22), (4096, 16, 1 << 20), (256, 48, 1 << 21),
          (256, 64, 1 << 20), (1024, 192, 1 << 18), (4096, 192, 1 << 18), (2, 16, 1 << 22), (8, 16, 1 << 22), (64, 4, 1 << 22)]
g = torch.Generator(device=dev).manual_seed(0)
rows = []
print(f"{'K':>5} {'d':>4} {'N':>9} | {'encode ms':>10} {'Gvec/s':>8} {'Tops/s':>8} | {'lbg ms':>9} {'Gvec/s':>8}")
for K, d, N in shapes:
    # clustered data: K centres + noise; the codebook is the centres plus a sub-pixel offset
    centres = torch.randint(0, 256, (K, d), device=dev, generator=g)
    idx = torch.randint(0, K, (N,), device=dev, generator=g)
    blocks = (centres[idx] + torch.randint(-12, 13, (N, d), device=dev, generator=g)).clamp(0, 255).to(torch.uint8)
    cb = centres.double() + torch.rand((K, d), device=dev, dtype=torch.float64, generator=g) - 0.5
    cb = cb.clamp(0, 255.9)
    t_enc = timeit(lambda: eng.encode(blocks, cb))
    t_lbg = timeit(lambda: eng.lbg_accum(blocks, cb))
    srt = eng.sort_blocks(blocks)
    pruned = srt is not None and eng.prune_supported(K, d)
    t_enc_p = timeit(lambda: eng.encode(blocks, cb, sorted=srt)) if pruned else float("nan")
    t_lbg_p = timeit(lambda: eng.lbg_accum(blocks, cb, sorted=srt)) if pruned else float("nan")
    # ambiguous vectors of the last full search: the 64 sub-queue counters sit 64 bytes apart in front of
    # the queues, which end the workspace
    need = lib.vq_workspace_bytes(K, d, N)
    cap = ((N + 63) // 64 + 63) // 64 * 64
    qoff = need - ((64 * cap * 4 + 255) // 256) * 256 - 64 * 64
    eng.encode(blocks, cb)
    counters = eng._ws[qoff:qoff + 64 * 64].view(torch.int32)[::16]
    amb = int(counters.sum().item()) / N
    row = dict(K=K, d=d, N=N, encode_ms=t_enc, encode_tops=2.0 * K * d * N / t_enc / 1e9, lbg_ms=t_lbg,
               pruned_encode_ms=t_enc_p, pruned_lbg_ms=t_lbg_p, ambiguous_frac=amb)
    rows.append(row)
    print(f"{K:5d} {d:4d} {N:9d} | {t_enc:10.3f} {N / t_enc / 1e6:8.2f} {2.0 * K * d * N / t_enc / 1e9:8.1f} | {t_lbg:9.3f} {N / t_lbg / 1e6:8.2f} | "
          f"pruned {t_enc_p:8.3f} {t_lbg_p:8.3f} | amb {amb * 100:6.3f}%", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/shape_sweep.json", "w") as f:
    json.dump(rows, f, indent=1)
