#!/usr/bin/env python3
"""Randomised shapes and data through the emulated kernels (tests/emu) against the oracle.
Not collected by pytest; run by hand:  python tests/emu/fuzz.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from emu_engine import EmuEngine, emulated_kernels, _EMU_DEVICE as GUARDED  # noqa: E402
from oracle import vq_oracle as o  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
DIMS = [1, 2, 3, 4, 5, 8, 11, 12, 15, 16, 17, 24, 27, 31, 32, 33, 48, 64, 75, 96, 128, 192, 200]


def make_case():
    d = int(rng.choice(DIMS))
    K = int(rng.choice([1, 2, 3, 7, 16, 31, 32, 33, 64, 100, 255, 256, 257, 500, 1024, 2048, 2049]))
    if K * d > 60000:
        K = max(1, 60000 // d)
    N = int(rng.integers(1, 4000)) if rng.random() < 0.8 else int(rng.integers(8000, 9500))
    style = rng.choice(["structured", "iid", "constant", "dups", "wide"])
    if style == "structured":
        cb = np.clip(rng.random((K, 1)) * 255 + rng.normal(0, 25, (K, d)), -5, 300)
        blocks = np.clip(cb[rng.integers(0, K, N)] + rng.normal(0, 12, (N, d)), 0, 255).astype(np.uint8)
    elif style == "iid":
        cb = rng.random((K, d)) * 255
        blocks = rng.integers(0, 256, size=(N, d), dtype=np.uint8)
    elif style == "constant":
        cb = np.full((K, d), float(rng.integers(0, 256))) * (1 + 0.01 * rng.standard_normal((K, 1)))
        blocks = np.full((N, d), int(rng.integers(0, 256)), dtype=np.uint8)
    elif style == "dups":
        base = rng.random((max(1, K // 3), d)) * 255
        cb = base[rng.integers(0, len(base), K)]
        blocks = rng.integers(0, 256, size=(N, d), dtype=np.uint8)
        blocks[: N // 2] = np.clip(np.rint(cb[rng.integers(0, K, N // 2)]), 0, 255).astype(np.uint8)   # exact hits, ties
    else:
        cb = rng.random((K, d)) * 2000 - 500
        blocks = rng.integers(0, 256, size=(N, d), dtype=np.uint8)
    return style, blocks, cb


def T(a):
    """Host array -> tensor that ends at a protected page (see emu_engine): kernels that read past the
    end of an input are caught, not only those that write past the end of an output."""
    src = torch.from_numpy(np.ascontiguousarray(a))
    t = torch.empty(tuple(src.shape), dtype=src.dtype, device=GUARDED)
    t.copy_(src)
    return t


t_end = time.time() + budget
n_cases = 0
with emulated_kernels():
    eng = EmuEngine()
    while time.time() < t_end:
        style, blocks, cb = make_case()
        N, d = blocks.shape
        K = cb.shape[0]
        use_sorted = bool(rng.random() < 0.6)
        what = rng.choice(["encode", "lbg"])
        tag = f"case {n_cases}: {what} style={style} K={K} d={d} N={N} sorted={use_sorted}"
        b = T(blocks)
        srt = eng.sort_blocks(b) if use_sorted else None
        want_l, want_d = o.encode_l1(blocks, cb, want_dist=True)
        if what == "encode":
            got_l, got_d = eng.encode(b, T(cb), want_dist=True, mode=1, sorted=srt)
            assert np.array_equal(got_l.numpy(), want_l), tag
            assert np.array_equal(got_d.numpy(), want_d), tag
        else:
            acc, dist, labels = eng.lbg_accum(b, T(cb), want_labels=True, mode=1, sorted=srt)
            sums, counts = o.accumulate(blocks, want_l, K)
            a = acc.numpy().astype(np.uint64)
            assert np.array_equal(labels.numpy(), want_l), tag
            assert np.array_equal(a[: K * d].reshape(K, d), sums), tag
            assert np.array_equal(a[K * d:], counts), tag
            assert dist.item() == o.numpy_sum(want_d), tag
        n_cases += 1
        if n_cases % 10 == 0:
            print(f"{n_cases} cases ok (last: {tag})", flush=True)
print(f"done: {n_cases} cases, all equal to the oracle")
