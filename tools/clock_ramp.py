#!/usr/bin/env python3
"""How long the card takes to reach its sustained clock: the encode step issued back to back after an idle second,
timed in groups of ten (HIP events).  Usage: python3 tools/clock_ramp.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vector_quantization_amd import HipEngine, train_lbg  # noqa: E402
from vector_quantization_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
eng = HipEngine(dev)
img = bench.structured_image(8192, 8192, 1000, dev)
crop = eng.blocks_from_image(bench.structured_image(2048, 2048, 0, dev), 4, 4)
cb = train_lbg(eng, crop, 256, log=None).codebook
N = 2048 * 2048
labels = torch.empty(N, dtype=torch.int32, device=dev)
scratch = torch.empty(_lib.lib.vq_sorted_bytes(16, N), dtype=torch.uint8, device=dev)


def step():
    eng.encode_image(img, 4, 4, cb, scratch=scratch, labels=labels)
    return eng.decode(labels.view(2048, 2048), cb, 4, 4, 1)


for idle in (1.0, 0.05, 10.0):
    step()
    torch.cuda.synchronize()
    time.sleep(idle)
    groups = 40
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(groups + 1)]
    ev[0].record()
    for g in range(groups):
        for _ in range(10):
            step()
        ev[g + 1].record()
    torch.cuda.synchronize()
    ms = [ev[g].elapsed_time(ev[g + 1]) / 10 for g in range(groups)]
    print(f"after {idle} s idle, ms per step in groups of ten: " + " ".join(f"{m:.3f}" for m in ms), flush=True)
