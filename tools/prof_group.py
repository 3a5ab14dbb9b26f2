#!/usr/bin/env python3
"""Profiling target: LBG iterations through the multi-GPU code path (vq_lbg_iterate_local, RCCL all-gather,
vq_lbg_iterate_finish) with one rank on one GPU, at the headline shape.  Run under rocprofv3 --kernel-trace and read
the trace with tools/gaps.py.  Usage: python3 tools/prof_group.py [iterations]"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import structured_image  # noqa: E402
from vector_quantization_amd import HipEngine, train_lbg  # noqa: E402
from vector_quantization_amd.lbg import shards_chunk_aligned  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
eng = HipEngine(dev)
img = structured_image(8192, 8192, 1000, dev)
crop = eng.blocks_from_image(structured_image(2048, 2048, 0, dev), 4, 4)
cb = train_lbg(eng, crop, 256, log=None).codebook
srt, (N, d) = eng.sort_image(img, 4, 4)
aligned, sizes = shards_chunk_aligned(N, dist.group.WORLD, dev)
for n in (3, iters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.lbg_stage_group(None, cb, N, -1.0, n, dist.group.WORLD, sizes, aligned, sorted=srt, shape=(N, d))
    torch.cuda.synchronize()
    print(f"{n} iterations through the group path: {(time.perf_counter() - t0) / n * 1e3:.4f} ms each")
for n in (3, iters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.lbg_stage(None, cb, N, -1.0, n, sorted=srt, shape=(N, d))
    torch.cuda.synchronize()
    print(f"{n} iterations through vq_lbg_iterate: {(time.perf_counter() - t0) / n * 1e3:.4f} ms each")
dist.destroy_process_group()
