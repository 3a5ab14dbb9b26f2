#!/usr/bin/env python3
"""Filter-kernel time on the structured and the iid config-2 image and at K = 1024, for one library build."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vector_quantization_amd import HipEngine, train_lbg
dev = torch.device("cuda:0")
eng = HipEngine(dev)
crop = eng.blocks_from_image(bench.structured_image(2048, 2048, 0, dev), 4, 4)
out = []
for name, img, K in (("structured K=256", bench.structured_image(8192, 8192, 1000, dev), 256),
                     ("iid K=256", bench.iid_image(8192, 8192, 1, dev), 256),
                     ("structured K=1024", bench.structured_image(8192, 8192, 1000, dev), 1024)):
    cb = train_lbg(eng, crop, K, log=None).codebook
    prof = bench.search_profile(eng, img, 4, 4, cb, 30, dev)[0]
    out.append(f"{name}: filter {prof['filter_kernel_ms'] * 1e3:7.1f} us, pairs {prof['evaluated_pair_frac']:.3f}")
print(os.path.basename(os.environ.get("VQ_HIP_LIB", "default")), " | ".join(out))
