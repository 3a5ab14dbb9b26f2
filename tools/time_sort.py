#!/usr/bin/env python3
"""Times the window sort (from the image and from the block matrix) for rows longer than 16 bytes."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import structured_image
from vector_quantization_amd import HipEngine
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
    return statistics.median(t)
for (H, W, C, bh, bw) in [(4096, 4096, 3, 4, 4), (4096, 4096, 3, 8, 8), (8192, 8192, 1, 8, 8), (8192, 8192, 3, 4, 4), (4090, 4090, 3, 4, 4)]:
    img = structured_image(H, W, 1000, dev, channels=C)
    blocks = eng.blocks_from_image(img, bh, bw)
    N, d = blocks.shape
    srt = eng.sort_blocks(blocks)
    srt2, _ = eng.sort_image(img, bh, bw)
    t_x = timeit(lambda: eng.blocks_from_image(img, bh, bw))
    t_b = timeit(lambda: eng.sort_blocks(blocks, out=srt))
    t_i = timeit(lambda: eng.sort_image(img, bh, bw, out=srt2))
    gb = N * (2 * d + 2) / 1e9
    print(f"H={H} W={W} C={C} {bh}x{bw} N={N} d={d}: extract {t_x:6.1f} us  sort_blocks {t_b:7.1f} us ({gb / t_b * 1e6:6.0f} GB/s)  sort_image {t_i:7.1f} us ({gb / t_i * 1e6:6.0f} GB/s)")
