#!/usr/bin/env python3
"""LBG training to K = 1024 on a 16384 x 16384 image (the size of configs[2]) and one K = 1024 iteration, for one library build."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vector_quantization_amd import HipEngine, train_lbg
dev = torch.device("cuda:0")
eng = HipEngine(dev)
img = bench.structured_image(16384, 16384, 2000, dev)
blocks = eng.blocks_from_image(img, 4, 4)
N, d = blocks.shape
srt = eng.sort_blocks(blocks)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = train_lbg(eng, blocks, 1024, log=None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
t1 = time.perf_counter()
eng.lbg_stage(blocks, res.codebook, N, -1.0, 10, sorted=srt)
torch.cuda.synchronize(); it = (time.perf_counter() - t1) / 10
print(os.path.basename(os.environ.get("VQ_HIP_LIB", "default")), f"training {dt * 1e3:.1f} ms, {sum(res.iters_per_stage)} iterations; K=1024 iteration {it * 1e3:.3f} ms")
