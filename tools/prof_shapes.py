#!/usr/bin/env python3
"""Profiling target: the kernels of the shapes beside the headline workload -- d = 192 with K = 4096 (LDS-tiled
pruned search + shortlist fix-up + three-launch sort), RGB 4x4 (d = 48, K = 256: scalar-load pruned search, sort in
registers) and K = 1024 at d = 16 (search + k_accum_dist iteration).  Run under rocprofv3 --kernel-trace / --pmc.
Usage: python3 tools/prof_shapes.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import structured_image  # noqa: E402
from vector_quantization_amd import HipEngine, train_lbg, lbg_iteration  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
eng = HipEngine(dev)


def search(img, bh, bw, cb):
    srt, (N, d) = eng.sort_image(img, bh, bw)
    prepared = eng.prepare_codebook(cb, N)
    labels = torch.empty(N, dtype=torch.int32, device=dev)
    for _ in range(reps):
        eng.sort_image(img, bh, bw, out=srt)
        eng.encode_prepared(None, prepared, labels=labels, sorted=srt, shape=(N, d))
    return srt, N, d


# d = 192, K = 4096: one GPU's share of an 8192 x 8192 RGB image at 8 GPUs
rgb8 = structured_image(4096, 4096, 5, dev, channels=3)
cb192 = train_lbg(eng, eng.blocks_from_image(rgb8[:1024, :1024].contiguous(), 8, 8), 4096, log=None).codebook
torch.cuda.synchronize()
search(rgb8, 8, 8, cb192)
del rgb8
# RGB 4x4, d = 48, K = 256
rgb = structured_image(4096, 4096, 3, dev, channels=3)
cb48 = train_lbg(eng, eng.blocks_from_image(rgb[:1024, :1024].contiguous(), 4, 4), 256, log=None).codebook
torch.cuda.synchronize()
search(rgb, 4, 4, cb48)
del rgb
# K = 1024, d = 16: search and one LBG iteration (search + k_accum_dist)
img = structured_image(8192, 8192, 1000, dev)
cb1k = train_lbg(eng, eng.blocks_from_image(structured_image(2048, 2048, 0, dev), 4, 4), 1024, log=None).codebook
torch.cuda.synchronize()
srt, N, d = search(img, 4, 4, cb1k)
blocks = eng.blocks_from_image(img, 4, 4)
cur = cb1k
for _ in range(reps):
    cur, _ = lbg_iteration(eng, blocks, cur, N, sorted=srt)
torch.cuda.synchronize()
print("done")
