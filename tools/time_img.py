#!/usr/bin/env python3
"""Times the image-fused sort / encode against extract + sort + search (config 2 shape)."""
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
for (H, W, C, bh, bw, K) in [(8192, 8192, 1, 4, 4, 256), (4096, 4096, 3, 4, 4, 256), (8192, 8192, 1, 2, 2, 64)]:
    img = structured_image(H, W, 1000, dev)
    if C == 3:
        img = torch.stack([img, img.flip(0), img.flip(1)], dim=-1).contiguous()
    blocks = eng.blocks_from_image(img, bh, bw)
    N, d = blocks.shape
    crop = eng.blocks_from_image(img[:1024, :1024].contiguous(), bh, bw)
    cb = train_lbg(eng, crop, K, log=None).codebook
    srt = eng.sort_blocks(blocks)
    srt2, _ = eng.sort_image(img, bh, bw)
    scratch = torch.empty(srt.numel(), dtype=torch.uint8, device=dev)
    labels = torch.empty(N, dtype=torch.int32, device=dev)
    print(f"H={H} W={W} C={C} {bh}x{bw} K={K} N={N} d={d}")
    print("  blocks_from_image ", timeit(lambda: eng.blocks_from_image(img, bh, bw)))
    print("  sort_blocks       ", timeit(lambda: eng.sort_blocks(blocks, out=srt)))
    print("  sort_image        ", timeit(lambda: eng.sort_image(img, bh, bw, out=srt2)))
    print("  encode(sorted)    ", timeit(lambda: eng.encode(blocks, cb, sorted=srt)))
    print("  encode_image      ", timeit(lambda: eng.encode_image(img, bh, bw, cb, scratch=scratch, labels=labels)))
    print("  decode            ", timeit(lambda: eng.decode(labels.view(H // bh, W // bw), cb, bh, bw, C)))
