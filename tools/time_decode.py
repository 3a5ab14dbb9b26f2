#!/usr/bin/env python3
"""Times block extraction and decode for several image / block shapes (HBM-bound byte movers), with their rates
against the algorithmic bytes (2 * d per vector for extraction, d + 4 for decode)."""
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
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)
for (H, W, C, bh, bw, K) in [(8192, 8192, 1, 4, 4, 256), (4096, 4096, 3, 4, 4, 256), (4090, 4090, 3, 4, 4, 256), (8192, 8192, 3, 4, 4, 256),
                             (4096, 4096, 3, 8, 8, 4096), (8192, 8192, 1, 8, 8, 256), (8192, 8192, 1, 2, 2, 16), (4096, 4096, 3, 2, 2, 16)]:
    img = structured_image(H, W, 1000, dev, channels=C)
    nr, nc = -(-H // bh), -(-W // bw)
    N, d = nr * nc, bh * bw * C
    cb = torch.rand(K, d, dtype=torch.float64, device=dev) * 255
    labels = torch.randint(0, K, (nr, nc), dtype=torch.int32, device=dev)
    out = torch.empty((nr * bh, nc * bw, C), dtype=torch.uint8, device=dev)
    t_x = timeit(lambda: eng.blocks_from_image(img, bh, bw))
    t_d = timeit(lambda: eng.decode(labels, cb, bh, bw, C, out=out))
    print(f"H={H} W={W} C={C} {bh}x{bw} K={K} N={N} d={d}: extract {t_x:6.1f} us ({2 * N * d / t_x / 1e6:5.2f} TB/s)  "
          f"decode {t_d:6.1f} us ({N * (d + 4) / t_d / 1e6:5.2f} TB/s)")
