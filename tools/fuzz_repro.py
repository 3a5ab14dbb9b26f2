#!/usr/bin/env python3
"""Re-runs the case tools/gpu_fuzz.py saved last (gpurun_out/fuzz_last_case.npz) and says where the paths differ."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantization_amd import HipEngine
dev = torch.device("cuda:0")
eng = HipEngine(dev)
z = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fuzz_last_case.npz")
img, cb, bh, bw = z["img"], z["cb"], int(z["bh"]), int(z["bw"])
timg, tcb = torch.from_numpy(img).to(dev), torch.from_numpy(cb).to(dev)
blocks = eng.blocks_from_image(timg, bh, bw)
N, d = blocks.shape
K = cb.shape[0]
print(f"img {img.shape} {bh}x{bw} N={N} d={d} K={K}")
want_l, want_d = eng.encode(blocks, tcb, want_dist=True, mode=0)
srt2, _ = eng.sort_image(timg, bh, bw)
srt = eng.sort_blocks(blocks)
for name, s, b in (("image-sorted", srt2, None), ("block-sorted", srt, None), ("unsorted", None, blocks)):
    if s is None and b is None:
        continue
    for rep in range(2):
        acc, dist, lab = eng.lbg_accum(b, tcb, want_labels=True, sorted=s, shape=(N, d))
        bad = (lab != want_l).nonzero().flatten()
        sums = torch.zeros(K, d, dtype=torch.int64, device=dev).index_add_(0, want_l.long(), blocks.long())
        counts = torch.bincount(want_l.long(), minlength=K)
        print(name, rep, "labels differ at", bad.numel(), bad[:10].tolist(), "sums equal", torch.equal(acc[: K * d].view(K, d), sums),
              "counts equal", torch.equal(acc[K * d:], counts), "total", float(dist), float(eng.numpy_sum(want_d)))
        if bad.numel():
            i = int(bad[0])
            print("   vector", i, "got", int(lab[i]), "want", int(want_l[i]), "window", i // 8192, "of", (N + 8191) // 8192,
                  "bad per window", torch.bincount(bad // 8192, minlength=(N + 8191) // 8192).tolist())
    got_l, got_d = eng.encode(blocks, tcb, want_dist=True, mode=1, sorted=s) if s is not None else eng.encode(blocks, tcb, want_dist=True, mode=1)
    print(name, "encode: labels differ", int((got_l != want_l).sum()), "dist differ", int((got_d != want_d).sum()))
