#!/usr/bin/env python3
"""Times encode_prepared (config 2 shape) for one library build; the library is chosen with
VQ_HIP_LIB, so a shell loop over variant builds compares tuning knobs on one device.
Prints: name, median / min microseconds of the encode launch pair over 30 calls."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import structured_image
from vector_quantization_amd import HipEngine, train_lbg
dev = torch.device("cuda:0")
eng = HipEngine(dev)
img = structured_image(8192, 8192, 1000, dev)
blocks = eng.blocks_from_image(img, 4, 4)
N = blocks.shape[0]
crop = eng.blocks_from_image(structured_image(2048, 2048, 0, dev), 4, 4)
cb = train_lbg(eng, crop, 256, log=None).codebook
prepared = eng.prepare_codebook(cb, N)
labels = torch.empty(N, dtype=torch.int32, device=dev)
for _ in range(5):
    eng.encode_prepared(blocks, prepared, labels=labels)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
for a, b in ev:
    a.record(); eng.encode_prepared(blocks, prepared, labels=labels); b.record()
torch.cuda.synchronize()
t = [a.elapsed_time(b) * 1e3 for a, b in ev]
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
for a, b in ev:
    a.record(); eng.lbg_accum(blocks, cb); b.record()
torch.cuda.synchronize()
t2 = [a.elapsed_time(b) * 1e3 for a, b in ev]
srt = eng.sort_blocks(blocks)
def timeit(fn, n=30):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) * 1e3 for a, b in ev]
    return statistics.median(t), min(t)
lab2 = torch.empty_like(labels)
p_enc = timeit(lambda: eng.encode_prepared(blocks, prepared, labels=lab2, sorted=srt))
p_sort = timeit(lambda: eng.sort_blocks(blocks, out=srt))
p_lbg = timeit(lambda: eng.lbg_accum(blocks, cb, sorted=srt))
ok = bool(torch.equal(lab2, labels))
print(f"PRUNED encode med {p_enc[0]:7.1f} min {p_enc[1]:7.1f} | sort med {p_sort[0]:6.1f} | lbg_accum med {p_lbg[0]:7.1f} min {p_lbg[1]:7.1f} same={ok} || ", end="")
print(f"{os.environ.get('VQ_HIP_LIB', 'default'):60s} encode med {statistics.median(t):7.1f} min {min(t):7.1f} us | lbg_accum med {statistics.median(t2):7.1f} min {min(t2):7.1f} us  checksum {int(labels.sum().item())}")
