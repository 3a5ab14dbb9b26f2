#!/usr/bin/env python3
"""The fused LBG kernel on a freshly split codebook against the same codebook after the stage's iterations: kernel
time (events around the launch), share of (vector, codeword) pairs evaluated, share of vectors re-checked in fp64.
Usage: python3 tools/time_after_split.py [K ...]   (16384 x 16384 gray, 4x4 blocks: 16.7 M vectors)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vector_quantization_amd import HipEngine, train_lbg  # noqa: E402

dev = torch.device("cuda:0")
eng = HipEngine(dev)
img = bench.structured_image(16384, 16384, 7, dev)
srt, (N, d) = eng.sort_image(img, 4, 4)
crop = eng.blocks_from_image(bench.structured_image(2048, 2048, 0, dev), 4, 4)
for K in [int(a) for a in sys.argv[1:]] or [32, 64, 256]:
    half = train_lbg(eng, crop, K // 2, log=None).codebook
    cb_split = eng.lbg_split(half, 0.01)
    cb_done = eng.lbg_stage(None, cb_split, N, 0.001, 20, sorted=srt, shape=(N, d))[0]
    for name, cb in (("split", cb_split), ("after the stage", cb_done)):
        ms = eng.search_kernel_ms(lambda: eng.lbg_stage(None, cb, N, -1.0, 1, sorted=srt, shape=(N, d)), 10)
        queued, pairs = eng.search_statistics((cb, eng._workspace(K, d, N)), N)
        print(f"K={K} {name}: kernel {ms * 1e3:.0f} us, pairs evaluated {pairs / (N * K):.3f}, fp64 re-checks {queued / N:.4f}", flush=True)
