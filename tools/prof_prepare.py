#!/usr/bin/env python3
"""Runs vq_codebook_prepare a few times (config 2 codebook) -- with a library built with -DVQ_PREP_STAMPS the
kernel prints the time it spent in each of its phases."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import structured_image
from vector_quantization_amd import HipEngine, train_lbg
dev = torch.device("cuda:0")
eng = HipEngine(dev)
crop = eng.blocks_from_image(structured_image(2048, 2048, 0, dev), 4, 4)
cb = train_lbg(eng, crop, int(sys.argv[1]) if len(sys.argv) > 1 else 256, log=None).codebook
torch.cuda.synchronize()
print("---- timed calls ----", flush=True)
for _ in range(4):
    eng.prepare_codebook(cb, 0)
    torch.cuda.synchronize()
