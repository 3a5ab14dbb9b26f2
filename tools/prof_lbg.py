#!/usr/bin/env python3
"""Profiling target: LBG iterations of the headline workload only (one stage of forced iterations)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import structured_image
from vector_quantization_amd import HipEngine, train_lbg
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
eng = HipEngine(dev)
img = structured_image(8192, 8192, 1000, dev)
blocks = eng.blocks_from_image(img, 4, 4)
N = blocks.shape[0]
crop = eng.blocks_from_image(structured_image(2048, 2048, 0, dev), 4, 4)
cb = train_lbg(eng, crop, 256, log=None).codebook
srt, _ = eng.sort_image(img, 4, 4)
torch.cuda.synchronize()
eng.lbg_stage(blocks, cb, N, -1.0, reps, sorted=srt)
torch.cuda.synchronize()
print("done")
