#!/usr/bin/env python3
"""Random small images on the GPU against the CPU oracle (test infrastructure: this file lives under tests/ because it
loads oracle/): block matrix in both padding modes, whole training runs (codebook, iterations per stage, the
`Converged at iter` indices and every distortion bit for bit), search with distances, decode, packed labels.
tools/gpu_fuzz.py covers large shapes against the library's own all-fp64 search; this one keeps the inputs at sizes
the C restatement finishes in milliseconds and checks against an implementation that shares no code with the library.
Usage: python3 tests/gpu_fuzz_oracle.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vq_oracle  # noqa: E402
from vector_quantization_amd import HipEngine, train_lbg  # noqa: E402

vq_oracle.build()
dev = torch.device("cuda:0")
eng = HipEngine(dev)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    C = int(rng.choice([1, 1, 3]))
    bh, bw = (int(v) for v in rng.choice([1, 2, 3, 4, 8], size=2))
    d = bh * bw * C
    H = int(rng.integers(1, 70 * bh)) if rng.random() < 0.7 else int(rng.integers(1, 400))
    W = int(rng.integers(1, 70 * bw)) if rng.random() < 0.7 else int(rng.integers(1, 400))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        img = rng.integers(0, 256, size=(H, W, C), dtype=np.uint8)
    elif kind == 3:                               # few grey levels: empty clusters, duplicate codewords, exact ties
        img = (rng.integers(0, 4, size=(H, W, C)) * 85).astype(np.uint8)
    else:
        yy, xx = np.mgrid[0:H, 0:W]
        base = 128 + 100 * np.sin(xx / 17.0 + rng.random() * 6) * np.cos(yy / 11.0) + rng.normal(0, 4 if kind == 1 else 0.5, (H, W))
        img = np.clip(np.stack([base + 9 * c for c in range(C)], -1), 0, 255).astype(np.uint8)
    pad = int(rng.integers(0, 2))
    want_blocks = vq_oracle.blocks_u8(img, bh, bw, pad)
    timg = torch.from_numpy(img).to(dev)
    blocks = eng.blocks_from_image(timg, bh, bw, pad)
    N = blocks.shape[0]
    tag = f"case {cases}: H={H} W={W} C={C} {bh}x{bw} d={d} N={N} kind={kind} pad={pad}"
    assert np.array_equal(blocks.cpu().numpy(), want_blocks), "blocks differ, " + tag
    k = int(rng.choice([1, 2, 3, 5, 8, 16, 20, 33, 64, 100, 128]))
    if N * max(k, 2) * d > 6e7:
        k = 8
    thr = float(rng.choice([0.001, 0.001, 0.05, 0.0]))
    eps = float(rng.choice([0.01, 0.01, 0.1, 0.001]))
    max_it = int(rng.integers(1, 9))
    tag += f" k={k} thr={thr} eps={eps} max_it={max_it}"
    if k > N:
        try:
            train_lbg(eng, blocks, k, eps, thr, max_it, log=None)
        except ValueError:
            cases += 1
            continue
        raise AssertionError("k > N was accepted, " + tag)
    want = vq_oracle.lbg_train(want_blocks, k, eps, thr, max_it)
    got = train_lbg(eng, blocks, k, eps, thr, max_it, log=None)
    assert np.array_equal(got.codebook.cpu().numpy(), want["codebook"]), "trained codebook differs, " + tag
    assert list(got.iters_per_stage) == list(want["iters_per_stage"]), "iterations per stage differ, " + tag
    assert list(got.converged_iter) == list(want["converged_iter"]), "converged-at indices differ, " + tag
    assert np.array_equal(np.asarray(got.distortions, dtype=np.float64), want["distortions"], equal_nan=True), \
        "distortions differ, " + tag
    cb = want["codebook"]
    K = cb.shape[0]
    want_l, want_d = vq_oracle.encode_l1(want_blocks, cb, want_dist=True)
    tcb = torch.from_numpy(cb).to(dev)
    got_l, got_d = eng.encode_image(timg, bh, bw, tcb, pad, want_dist=True)
    assert np.array_equal(got_l.cpu().numpy(), want_l) and np.array_equal(got_d.cpu().numpy(), want_d), "search differs, " + tag
    nr, nc = -(-H // bh), -(-W // bw)
    dec = eng.decode(got_l.view(nr, nc), tcb, bh, bw, C)
    assert np.array_equal(dec.cpu().numpy(), vq_oracle.decode_u8(want_l.reshape(nr, nc), cb, bh, bw, C)), "decode differs, " + tag
    assert np.array_equal(eng.pack_labels(got_l, K).cpu().numpy(), vq_oracle.pack_labels(want_l, K)), "packed labels differ, " + tag
    # per-cluster sums / counts for given labels, the centroid update and the split on their own
    ws, wc = vq_oracle.accumulate(want_blocks, want_l, K)
    acc = eng.accum_labels(blocks, got_l, K)
    acc_h = acc.cpu().numpy().astype(np.uint64)
    assert np.array_equal(acc_h[: K * d].reshape(K, d), ws) and np.array_equal(acc_h[K * d:], wc), "sums for given labels differ, " + tag
    assert np.array_equal(eng.lbg_finalize(acc, tcb).cpu().numpy(), vq_oracle.finalize(ws, wc, cb)), "centroid update differs, " + tag
    assert np.array_equal(eng.lbg_split(tcb, eps).cpu().numpy(), vq_oracle.split(cb, eps)), "split differs, " + tag
    assert np.array_equal(eng.colsum(blocks).cpu().numpy().astype(np.uint64), vq_oracle.colsum_u8(want_blocks)), "column sums differ, " + tag
    assert float(eng.numpy_sum(got_d).item()) == vq_oracle.numpy_sum(want_d), "numpy-order sum differs, " + tag
    cases += 1
    if cases % 50 == 0:
        print(f"{cases} cases ok, last: {tag}", flush=True)
print(f"oracle fuzz ok: {cases} cases in {budget:.0f} s")
