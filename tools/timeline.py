#!/usr/bin/env python3
"""Wave timeline of the search kernel (diagnostic builds with -DVQ_TIMELINE only; VQ_HIP_LIB names the build):
per wave its start / end time, the SIMD it ran on and the codebook rows it searched.  Prints the mean number of
resident waves per SIMD over the kernel's duration in 20 slices, wave lifetimes, and the spread of work per wave.
Usage: VQ_HIP_LIB=build/variants/libvq_timeline.so python3 tools/timeline.py [cfg2|cfg5|rgb48|iid]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vector_quantization_amd import HipEngine, train_lbg, _lib  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dev = torch.device("cuda:0")
eng = HipEngine(dev)
if which == "cfg2":
    img = bench.structured_image(8192, 8192, 1000, dev)
    bh, K = 4, 256
    crop = bench.structured_image(2048, 2048, 0, dev)
elif which == "iid":
    img = bench.iid_image(8192, 8192, 1, dev)
    bh, K = 4, 256
    crop = img[:1024, :1024].contiguous()
elif which == "g64":
    img = bench.structured_image(8192, 8192, 7, dev)
    bh, K = 8, 256
    crop = img[:2048, :2048].contiguous()
elif which == "rgb48":
    img = bench.structured_image(4096, 4096, 3, dev, channels=3)
    bh, K = 4, 256
    crop = img[:1024, :1024].contiguous()
else:
    img = bench.structured_image(4096, 4096, 5, dev, channels=3)
    bh, K = 8, 4096
    crop = img[:1024, :1024].contiguous()
cb = train_lbg(eng, eng.blocks_from_image(crop, bh, bh), K, log=None).codebook
srt, (N, d) = eng.sort_image(img, bh, bh)
prepared = eng.prepare_codebook(cb, N)
labels = torch.empty(N, dtype=torch.int32, device=dev)
mode = os.environ.get("TL_MODE", "search")
for _ in range(3):
    if mode == "lbg":
        eng.lbg_accum(None, cb, sorted=srt, shape=(N, d))
    else:
        eng.encode_prepared(None, prepared, labels=labels, sorted=srt, shape=(N, d), phases=1)
torch.cuda.synchronize()
R = int(os.environ.get("TL_R", "2"))
nwaves = min(1 << 18, -(-N // (64 * R)))
buf = np.zeros(8 * nwaves, dtype=np.uint32)
fn = _lib.lib.vq_debug_timeline
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert fn(buf.ctypes.data, buf.size) == 0
t = buf.reshape(-1, 8)
t0, t1, hw, rows = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2], t[:, 3]
start = t0.min()
t0 -= start
t1 -= start
span = t1.max()
life = (t1 - t0)
print(f"{which} {mode}: {nwaves} waves, kernel span {span / 100:.1f} us; wave life mean {life.mean() / 100:.2f} us "
      f"p10 {np.percentile(life, 10) / 100:.2f} p50 {np.percentile(life, 50) / 100:.2f} p90 {np.percentile(life, 90) / 100:.2f} "
      f"max {life.max() / 100:.2f}")
ph = [t[:, 4].astype(np.int64) - start - t0, t[:, 5].astype(np.int64) - t[:, 4].astype(np.int64),
      t[:, 6].astype(np.int64) - t[:, 5].astype(np.int64), t1 + start - t[:, 6].astype(np.int64)]
print("phases (mean us): load %.2f search %.2f resolve %.2f write %.2f" % tuple(p.mean() / 100 for p in ph))
print(f"rows searched per wave: mean {rows.mean():.1f} p10 {np.percentile(rows, 10):.0f} p50 {np.percentile(rows, 50):.0f} "
      f"p90 {np.percentile(rows, 90):.0f} max {rows.max()}")
print(f"last wave starts at {t0.max() / 100:.1f} us; mean resident waves/SIMD over the span {life.sum() / span / 1024:.2f}")
nb = 20
edges = np.linspace(0, span, nb + 1)
occ = []
for i in range(nb):
    lo, hi = edges[i], edges[i + 1]
    ov = np.clip(np.minimum(t1, hi) - np.maximum(t0, lo), 0, None).sum()
    occ.append(ov / (hi - lo) / 1024)
print("resident waves per SIMD by slice of the span:", " ".join(f"{o:.1f}" for o in occ))
# per SIMD: busy span (first start to last end) and idle tail
xcc = hw >> 28
se = (hw >> 13) & 7
cu = (hw >> 8) & 15
simd = (hw >> 4) & 3
sid = ((xcc.astype(np.int64) * 8 + se) * 16 + cu) * 4 + simd
ends = np.zeros(sid.max() + 1, dtype=np.int64)
np.maximum.at(ends, sid, t1)
work = np.zeros(sid.max() + 1, dtype=np.int64)
np.add.at(work, sid, rows.astype(np.int64))
used = ends > 0
print(f"SIMDs seen {used.sum()}; last end per SIMD: min {ends[used].min() / 100:.1f} p50 {np.percentile(ends[used], 50) / 100:.1f} "
      f"max {ends[used].max() / 100:.1f} us; rows per SIMD: min {work[used].min()} mean {work[used].mean():.0f} max {work[used].max()}")
os.makedirs("gpurun_out", exist_ok=True)
np.save(f"gpurun_out/timeline_{which}_{mode}.npy", t)
