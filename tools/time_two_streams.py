#!/usr/bin/env python3
"""Throughput of the image -> labels -> image step when consecutive steps alternate between S streams (each with its
own engine, sorted buffer, labels and output), against one stream.  Usage: python3 tools/time_two_streams.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vector_quantization_amd import HipEngine, train_lbg, _lib  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
H = W = 8192
img = bench.structured_image(H, W, 1000, dev)
eng0 = HipEngine(dev)
crop = eng0.blocks_from_image(bench.structured_image(2048, 2048, 0, dev), 4, 4)
cb = train_lbg(eng0, crop, 256, log=None).codebook
nr, nc = H // 4, W // 4
N, d = nr * nc, 16
for S in (1, 2, 3, 4, 6, 8):
    engs = [HipEngine(dev) for _ in range(S)]
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    labels = [torch.empty(N, dtype=torch.int32, device=dev) for _ in range(S)]
    scratch = [torch.empty(_lib.lib.vq_sorted_bytes(d, N), dtype=torch.uint8, device=dev) for _ in range(S)]

    def step(i):
        k = i % S
        with torch.cuda.stream(streams[k]):
            engs[k].encode_image(img, 4, 4, cb, scratch=scratch[k], labels=labels[k])
            return engs[k].decode(labels[k].view(nr, nc), cb, 4, 4, 1)

    for i in range(10):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all(torch.equal(labels[0], l) for l in labels)
    print(f"{S} stream(s): {dt / steps * 1e3:.4f} ms per step, {N * steps / dt:.3e} vectors/s, labels equal {ok}")
