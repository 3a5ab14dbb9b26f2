#!/usr/bin/env python3
"""Diagnostic: how the fix-up kernel's time depends on the number of re-checked vectors."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import structured_image
from vector_quantization_amd import HipEngine, train_lbg
from vector_quantization_amd._lib import lib
dev = torch.device("cuda:0")
eng = HipEngine(dev)
img = structured_image(8192, 8192, 1000, dev)
blocks = eng.blocks_from_image(img, 4, 4)
N = blocks.shape[0]
crop = eng.blocks_from_image(structured_image(2048, 2048, 0, dev), 4, 4)
cb = train_lbg(eng, crop, 256, log=None).codebook
prepared = eng.prepare_codebook(cb, N)
ws = prepared[1]
qoff = lib.vq_workspace_bytes(256, 16, N) - ((N * 4 + 255) // 256) * 256 - 256
labels = torch.empty(N, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for _ in range(3):
    eng.encode_prepared(blocks, prepared, labels=labels)
torch.cuda.synchronize()
print("queued (real data):", int(ws[qoff:qoff + 4].view(torch.int32).item()), "of", N)
easy = cb.floor().clamp(0, 255).to(torch.uint8)[labels.long()].contiguous()
lab2 = torch.empty_like(labels)
for _ in range(3):
    eng.encode_prepared(easy, prepared, labels=lab2)
torch.cuda.synchronize()
print("queued (easy data):", int(ws[qoff:qoff + 4].view(torch.int32).item()))
hdr = ws[:16].view(torch.int32).tolist()
print("shift, emax, cmax:", hdr[:3])
