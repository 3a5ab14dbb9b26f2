#!/usr/bin/env python3
"""Filter-kernel timings (HIP events, median of 30) of the pruned search for one library build (VQ_HIP_LIB):
configs cfg2 / iid / rgb48 / cfg5 / k1024 as named on the command line, each checked against the all-fp64 search
on a slice.  Also the fused LBG pass at cfg2 (lbg).  Usage: python3 tools/time_search.py cfg2 rgb48 ..."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vector_quantization_amd import HipEngine, train_lbg, VQ_MODE_EXACT  # noqa: E402

dev = torch.device("cuda:0")
eng = HipEngine(dev)


def med(fn, n=30):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return statistics.median(t), t[0]


def setup(which):
    if which in ("cfg2", "lbg", "k1024", "k4096"):
        img = bench.structured_image(8192, 8192, 1000, dev)
        bh, K = 4, {"k1024": 1024, "k4096": 4096}.get(which, 256)
        crop = bench.structured_image(2048, 2048, 0, dev)
    elif which == "iid":
        img = bench.iid_image(8192, 8192, 1, dev)
        bh, K = 4, 256
        crop = img[:1024, :1024].contiguous()
    elif which in ("rgb48", "rgb48big"):
        side = 8192 if which == "rgb48big" else 4096
        img = bench.structured_image(side, side, 3, dev, channels=3)
        bh, K = 4, 256
        crop = img[:1024, :1024].contiguous()
    elif which == "g64":                                       # grayscale 8x8 blocks: 64-byte rows
        img = bench.structured_image(8192, 8192, 7, dev)
        bh, K = 8, 256
        crop = img[:2048, :2048].contiguous()
    elif which in ("cfg5", "cfg5big"):
        side = 8192 if which == "cfg5big" else 4096
        img = bench.structured_image(side, side, 5, dev, channels=3)
        bh, K = 8, 4096
        crop = img[:1024, :1024].contiguous()
    else:
        raise SystemExit(f"unknown shape {which}")
    cb = train_lbg(eng, eng.blocks_from_image(crop, bh, bh), K, log=None).codebook
    return img, bh, cb


out = []
for which in sys.argv[1:] or ["cfg2"]:
    img, bh, cb = setup(which)
    srt, (N, d) = eng.sort_image(img, bh, bh)
    prepared = eng.prepare_codebook(cb, N)
    labels = torch.empty(N, dtype=torch.int32, device=dev)
    if which == "lbg":
        m, lo = med(lambda: eng.lbg_accum(None, cb, sorted=srt, shape=(N, d)))
        m2, lo2 = med(lambda: eng.lbg_stage(None, cb, N, -1.0, 20, sorted=srt, shape=(N, d)), n=5)
        out.append(f"lbg_accum {m:.1f} us (min {lo:.1f}); stage iteration {m2 / 20:.1f} us")
        continue
    m, lo = med(lambda: eng.encode_prepared(None, prepared, labels=labels, sorted=srt, shape=(N, d), phases=1))
    mb, lob = med(lambda: eng.encode_prepared(None, prepared, labels=labels, sorted=srt, shape=(N, d)))
    queued, pairs = eng.search_statistics(prepared, N)
    eng.encode_prepared(None, prepared, labels=labels, sorted=srt, shape=(N, d))
    # all-fp64 cross-check on a slice of the vectors
    blocks = eng.blocks_from_image(img, bh, bh)
    ns = min(N, 1 << 18 if d <= 48 else 1 << 14)
    ref = eng.encode(blocks[:ns], cb, mode=VQ_MODE_EXACT)
    ok = bool(torch.equal(ref, labels[:ns]))
    frac = 2.0 * d * pairs / (m * 1e-6) / 1e12 / bench.PEAK_SAD_TOPS
    out.append(f"{which}: filter {m:.1f} us (min {lo:.1f}) filter+fixup {mb:.1f} us pairs {pairs / (N * cb.shape[0]):.3f} "
               f"frac {frac:.3f} near-ties {queued / N:.4f} labels_ok {ok}")
    del img, blocks
print(os.path.basename(os.environ.get("VQ_HIP_LIB", "default")) + "\n  " + "\n  ".join(out))
