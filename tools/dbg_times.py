#!/usr/bin/env python3
"""Debug: per-wave timestamps of the pruned search (library built with -DVQ_DBG_TIMES)."""
import os, sys
import numpy as np, torch
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
labels = torch.empty(N, dtype=torch.int32, device=dev)
srt = eng.sort_blocks(blocks)
for _ in range(3):
    eng.encode_prepared(blocks, prepared, labels=labels, sorted=srt)
torch.cuda.synchronize()
ws = prepared[1]
total = lib.vq_workspace_bytes(256, 16, N)
qoff = total - (N * 4 + 255) // 256 * 256
nw = N // 256
raw = ws[qoff + 2 * N: qoff + 2 * N + nw * 32].cpu().numpy().view(np.uint64).reshape(nw, 4)
t0 = raw[:, 0].min()
start = (raw[:, 0] - t0).astype(np.float64); loop = (raw[:, 1] - t0).astype(np.float64); end = (raw[:, 2] - t0).astype(np.float64)
print("clock span (ticks):", end.max(), " waves:", nw)
print("per-wave total  mean %.0f  p50 %.0f p90 %.0f p99 %.0f max %.0f" % (np.mean(end - start), *np.percentile(end - start, [50, 90, 99]), (end - start).max()))
print("per-wave loop   mean %.0f  p50 %.0f p90 %.0f p99 %.0f max %.0f" % (np.mean(loop - start), *np.percentile(loop - start, [50, 90, 99]), (loop - start).max()))
print("per-wave epilog mean %.0f  p50 %.0f p90 %.0f p99 %.0f max %.0f" % (np.mean(end - loop), *np.percentile(end - loop, [50, 90, 99]), (end - loop).max()))
pro = (raw[:, 3] >> np.uint64(32)).astype(np.float64); rounds = ((raw[:, 3] >> np.uint64(16)) & np.uint64(0xffff)).astype(np.float64)
ncw = (raw[:, 3] & np.uint64(0xffff)).astype(np.float64)
print("prologue cycles mean %.0f p50 %.0f p90 %.0f max %.0f" % (pro.mean(), *np.percentile(pro, [50, 90]), pro.max()))
print("rounds mean %.2f max %.0f | codewords evaluated mean %.1f p90 %.0f max %.0f" % (rounds.mean(), rounds.max(), ncw.mean(), np.percentile(ncw, 90), ncw.max()))
search = loop - start - pro
print("search cycles mean %.0f -> per codeword %.1f cycles (4 vectors x 11 instr x 4 = 176 if alone on the SIMD)" % (search.mean(), (search / ncw).mean()))
for lo_, hi_ in [(0, 48), (48, 80), (80, 128), (128, 257)]:
    m = (ncw >= lo_) & (ncw < hi_)
    if m.sum():
        print("  ncw in [%d,%d): %5d waves, search/ncw %.1f, rounds %.2f, total %.0f" % (lo_, hi_, m.sum(), (search[m] / ncw[m]).mean(), rounds[m].mean(), (end - start)[m].mean()))
np.save("gpurun_out/dbg_times.npy", raw)
