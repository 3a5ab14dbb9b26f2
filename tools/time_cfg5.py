#!/usr/bin/env python3
"""Search timings at d = 192, K = 4096 (8x8 RGB blocks) and d = 48, K = 256 for one library build."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vector_quantization_amd import HipEngine, train_lbg
dev = torch.device("cuda:0")
eng = HipEngine(dev)
out = []
for name, side, bh, K, seed in (("d192 K4096 seed7", 4096, 8, 4096, 7), ("d192 K4096 seed5", 4096, 8, 4096, 5)):
    rgb = bench.structured_image(side, side, seed, dev, channels=3)
    cb = train_lbg(eng, eng.blocks_from_image(rgb[:1024, :1024].contiguous(), bh, bh), K, log=None).codebook
    prof = bench.search_profile(eng, rgb, bh, bh, cb, 5, dev)[0]
    out.append(f"{name}: filter {prof['filter_kernel_ms'] * 1e3:8.1f} us fixup {prof['fixup_kernel_ms'] * 1e3:7.1f} us pairs {prof['evaluated_pair_frac']:.3f} queued {prof['queued_for_fp64_frac']:.4f}")
print(os.path.basename(os.environ.get("VQ_HIP_LIB", "default")), " | ".join(out))
