#!/usr/bin/env python3
"""Profiling target: the kernels of the headline workload (8192 x 8192 gray, 4x4 blocks, K=256) only (no host
baseline, few launches).  Run under rocprofv3 --kernel-trace / --pmc.  Usage: python3 tools/prof_encode.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import structured_image  # noqa: E402
from vector_quantization_amd import HipEngine, train_lbg  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
eng = HipEngine(dev)
img = structured_image(8192, 8192, 1000, dev)
blocks = eng.blocks_from_image(img, 4, 4)
N = blocks.shape[0]
crop = eng.blocks_from_image(structured_image(2048, 2048, 0, dev), 4, 4)
cb = train_lbg(eng, crop, 256, log=None).codebook
prepared = eng.prepare_codebook(cb, N)
labels = torch.empty(N, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
srt, _ = eng.sort_image(img, 4, 4)
for _ in range(reps):
    eng.encode_prepared(blocks, prepared, labels=labels)                 # full search
for _ in range(reps):
    eng.encode_image(img, 4, 4, cb, scratch=srt, labels=labels)          # image sort + prepare + pruned search + fix-up
    eng.decode(labels.view(2048, 2048), cb, 4, 4, 1)
eng.lbg_stage(None, cb, N, -1.0, reps, sorted=srt, shape=(N, 16))       # LBG iterations as train_lbg runs them (fused pass)
for _ in range(reps):
    eng.blocks_from_image(img, 4, 4)
torch.cuda.synchronize()
print("done")
