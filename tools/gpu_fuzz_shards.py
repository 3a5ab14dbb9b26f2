#!/usr/bin/env python3
"""Random shardings of one LBG iteration on one GPU: vq_lbg_iterate_local on 2 .. 4 shards one after the other (each
with its own workspace and state, some of them empty, cuts on and off the 8192-vector chunk boundaries), the exchange
done by hand in both layouts, vq_lbg_iterate_finish -- against vq_lbg_iterate on all vectors: integer sums and the
codebook bit for bit, the distortion bit for bit when the shards are chunk-aligned (else to 1e-12).
Usage: python3 tools/gpu_fuzz_shards.py [seconds] [seed]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantization_amd import HipEngine  # noqa: E402
from vector_quantization_amd._lib import lib, check  # noqa: E402

dev = torch.device("cuda:0")
eng = HipEngine(dev)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
CH = 8192


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def state_buf():
    n = lib.vq_lbg_state_bytes(4)
    st = torch.empty(n, dtype=torch.uint8, device=dev)
    check(lib.vq_lbg_stage_begin(P(st), n, 4, stream), "stage_begin")
    return st, n


t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    d = int(rng.choice([1, 3, 4, 8, 12, 16, 16, 16, 24, 27, 48, 64, 192]))
    K = int(rng.choice([2, 8, 16, 32, 64, 100, 256, 512, 1024, 2048]))
    world = int(rng.integers(2, 5))
    N = int(rng.integers(max(K, 64), 90000)) if rng.random() < 0.8 else int(rng.integers(200000, 700000))
    if d >= 64:
        N = min(N, 60000)
    if rng.random() < 0.5:                       # chunk-aligned cuts (what the driver of a sharded run arranges)
        cuts = sorted(int(c) * CH for c in rng.integers(0, N // CH + 1, world - 1))
    else:
        cuts = sorted(int(c) for c in rng.integers(0, N + 1, world - 1))
    bounds = [0] + cuts + [N]
    sizes = [bounds[i + 1] - bounds[i] for i in range(world)]
    aligned = all(n % CH == 0 for n in [s for s in sizes if s > 0][:-1])
    layout = "gathered" if rng.random() < 0.5 else "reduced"
    tag = f"case {cases}: K={K} d={d} N={N} sizes={sizes} aligned={aligned} layout={layout}"
    cbase = np.clip(rng.random((K, 1)) * 255.0 + rng.normal(0, 20, (K, d)), 0, 255)
    blocks_h = np.clip(cbase[rng.integers(0, K, N)] + rng.normal(0, 9, (N, d)), 0, 255).astype(np.uint8)
    cb_h = np.clip(cbase + rng.normal(0, 0.3, (K, d)), 0, 255)
    blocks = torch.from_numpy(blocks_h).to(dev)
    # the single-process iteration
    srt = eng.sort_blocks(blocks)
    use_sorted = srt is not None and eng.prune_supported(K, d)
    cb1 = torch.from_numpy(cb_h).to(dev)
    st1, nst = state_buf()
    nacc = K * d + K
    acc1 = torch.empty(nacc, dtype=torch.int64, device=dev)
    dist1 = torch.empty(N, dtype=torch.float64, device=dev)
    lab1 = torch.empty(N, dtype=torch.int32, device=dev)
    ws1 = torch.empty(lib.vq_workspace_bytes(K, d, N), dtype=torch.uint8, device=dev)
    check(lib.vq_lbg_iterate(P(blocks), P(srt) if use_sorted else None, srt.numel() if use_sorted else 0, N, d, P(cb1), K, N, 0.001,
                             0, P(acc1[:K * d]), P(acc1[K * d:]), P(lab1), P(dist1), 1, P(st1), nst, P(ws1), ws1.numel(),
                             stream), "vq_lbg_iterate")
    # the sharded iteration
    width = max(1, max(-(-n // CH) for n in sizes)) if aligned else 1
    gathered = torch.zeros((world, nacc + width), dtype=torch.int64, device=dev)
    scratch_chunks = torch.zeros(-(-N // CH) + 1, dtype=torch.float64, device=dev)
    accs, tails = [], []
    for r in range(world):
        n = sizes[r]
        sh = blocks[bounds[r]:bounds[r + 1]].contiguous()
        srt_r = eng.sort_blocks(sh) if (n and use_sorted) else None
        cb_r = torch.from_numpy(cb_h).to(dev)
        st_r, _ = state_buf()
        acc = torch.empty(nacc, dtype=torch.int64, device=dev)
        chunk = torch.zeros(-(-n // CH) + 1, dtype=torch.float64, device=dev)
        total = torch.zeros(1, dtype=torch.float64, device=dev)
        dist_r = torch.empty(max(n, 1), dtype=torch.float64, device=dev)
        lab_r = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        ws = torch.empty(lib.vq_workspace_bytes(K, d, n), dtype=torch.uint8, device=dev)
        nsrt = srt_r.numel() if srt_r is not None else 0
        if layout == "gathered":
            row = gathered[r]
            row_tail = row[nacc:].view(torch.float64)
            check(lib.vq_lbg_iterate_local(P(sh) if n else None, P(srt_r), nsrt, n, d, P(cb_r), K, 0, P(row[:K * d]),
                                           P(row[K * d:nacc]), P(row_tail) if aligned else P(scratch_chunks),
                                           None if aligned else P(row_tail), P(lab_r), P(dist_r), 1, P(st_r), nst, P(ws),
                                           ws.numel(), stream), "vq_lbg_iterate_local")
            accs.append(row[:nacc].clone())
            continue
        check(lib.vq_lbg_iterate_local(P(sh) if n else None, P(srt_r), nsrt, n, d, P(cb_r), K, 0, P(acc[:K * d]), P(acc[K * d:]),
                                       P(chunk), P(total), P(lab_r), P(dist_r), 1, P(st_r), nst, P(ws), ws.numel(), stream),
              "vq_lbg_iterate_local")
        accs.append(acc)
        t = torch.zeros(width, dtype=torch.float64, device=dev)
        if aligned:
            t[: -(-n // CH)] = chunk[: -(-n // CH)]
        else:
            t[0] = total[0]
        tails.append(t)
    acc_total = torch.stack(accs).sum(dim=0)
    counts = (ctypes.c_int * world)(*[(-(-n // CH) if aligned else 1) for n in sizes])
    cb2 = torch.from_numpy(cb_h).to(dev)
    st2, _ = state_buf()
    if layout == "gathered":
        stride = nacc + width
        check(lib.vq_lbg_iterate_finish(P(gathered[0, :K * d]), P(gathered[0, K * d:nacc]), world, stride, P(gathered[0, nacc:]),
                                        world, stride, counts, P(cb2), K, d, N, 0.001, 0, P(st2), nst, None, 0, stream),
              "vq_lbg_iterate_finish")
    else:
        chunks_all = torch.stack(tails).contiguous()
        check(lib.vq_lbg_iterate_finish(P(acc_total[:K * d]), P(acc_total[K * d:]), 1, 0, P(chunks_all), world, width, counts,
                                        P(cb2), K, d, N, 0.001, 0, P(st2), nst, None, 0, stream), "vq_lbg_iterate_finish")
    torch.cuda.synchronize(dev)
    assert torch.equal(acc_total, acc1), "sums / counts differ, " + tag
    assert torch.equal(cb2, cb1), "codebook differs, " + tag
    rec1 = st1[64:80].cpu().numpy().view(np.float64)[0]
    rec2 = st2[64:80].cpu().numpy().view(np.float64)[0]
    if aligned:
        assert rec1 == rec2, f"distortion differs ({rec1!r} / {rec2!r}), " + tag
    else:
        assert np.isclose(rec1, rec2, rtol=1e-12, atol=0), f"distortion differs ({rec1!r} / {rec2!r}), " + tag
    cases += 1
    if cases % 20 == 0:
        print(f"{cases} cases ok, last: {tag}", flush=True)
print(f"shard fuzz ok: {cases} cases in {budget:.0f} s")
