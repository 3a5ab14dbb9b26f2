#!/usr/bin/env python3
"""Times vq_codebook_prepare alone for a few (K, d)."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantization_amd import HipEngine
dev = torch.device("cuda:0")
eng = HipEngine(dev)
g = torch.Generator(device=dev).manual_seed(5)
for K, d in [(256, 16), (1024, 16), (256, 48), (1024, 48), (4096, 16), (4096, 192), (512, 192), (64, 4)]:
    cb = torch.rand(K, d, device=dev, generator=g, dtype=torch.float64) * 255.0
    for _ in range(3):
        eng.prepare_codebook(cb, 0)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); eng.prepare_codebook(cb, 0); b.record()
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) * 1e3 for a, b in ev]
    print(f"K={K:5d} d={d:4d} prepare med {statistics.median(t):8.1f} min {min(t):8.1f} us", flush=True)
