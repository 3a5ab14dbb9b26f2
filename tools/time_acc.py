#!/usr/bin/env python3
"""Splits the cost of the fused LBG kernel: search alone, + distances, + accumulation."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import structured_image
from vector_quantization_amd import HipEngine, train_lbg
dev = torch.device("cuda:0")
eng = HipEngine(dev)
def timeit(fn, n=30):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) * 1e3 for a, b in ev]
    return f"med {statistics.median(t):7.1f} min {min(t):7.1f} us"
img = structured_image(8192, 8192, 1000, dev)
blocks = eng.blocks_from_image(img, 4, 4)
N, d = blocks.shape
cb = train_lbg(eng, eng.blocks_from_image(structured_image(2048, 2048, 0, dev), 4, 4), 256, log=None).codebook
srt, _ = eng.sort_image(img, 4, 4)
prepared = eng.prepare_codebook(cb, N)
labels = torch.empty(N, dtype=torch.int32, device=dev)
print("filter only                ", timeit(lambda: eng.encode_prepared(None, prepared, labels=labels, sorted=srt, shape=(N, d), phases=1)))
print("filter + fixup             ", timeit(lambda: eng.encode_prepared(None, prepared, labels=labels, sorted=srt, shape=(N, d))))
print("filter + dist              ", timeit(lambda: eng.encode_prepared(None, prepared, labels=labels, sorted=srt, shape=(N, d), phases=1, want_dist=True)))
print("filter + fixup + dist      ", timeit(lambda: eng.encode_prepared(None, prepared, labels=labels, sorted=srt, shape=(N, d), want_dist=True)))
print("lbg_accum (fused)          ", timeit(lambda: eng.lbg_accum(None, cb, sorted=srt, shape=(N, d))))
print("lbg_stage 20 it (per iter) ", timeit(lambda: eng.lbg_stage(None, cb, N, -1.0, 20, sorted=srt, shape=(N, d)), n=5), "/20")
