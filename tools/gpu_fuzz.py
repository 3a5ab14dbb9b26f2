#!/usr/bin/env python3
"""Random shapes on the GPU: FAST search (plain, pruned, image-fused), the fused LBG pass and a three-iteration training
stage against the all-fp64 EXACT search of the same library (itself pinned to the CPU oracle by the test-suite).
Usage: python3 tools/gpu_fuzz.py [seconds] [seed]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantization_amd import HipEngine, train_lbg
dev = torch.device("cuda:0")
eng = HipEngine(dev)
eng_exact = HipEngine(dev, mode="exact")
from vector_quantization_amd._lib import lib, last_error, VQ_E_UNSUPPORTED  # noqa: E402
from vector_quantization_amd.lbg import shards_chunk_aligned  # noqa: E402
import ctypes  # noqa: E402


def P(t):
    return ctypes.c_void_p(t.data_ptr())


def STREAM():
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


group = None
if os.environ.get("VQ_FUZZ_GROUP") == "1":       # also the multi-GPU form of the training stage, on a group of one rank
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1)
    group = dist.group.WORLD
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    C = int(rng.choice([1, 1, 3]))
    bh, bw = (int(v) for v in rng.choice([1, 2, 3, 4, 8], size=2))
    d = bh * bw * C
    if d > 192:
        continue
    K = int(rng.choice([1, 2, 7, 16, 31, 32, 33, 64, 100, 255, 256, 257, 400, 512, 513, 1000, 1024, 1500, 4096, 4097, 5000]))
    H = int(rng.integers(bh, 40 * bh)) if rng.random() < 0.5 else int(rng.integers(1, 200))
    W = int(rng.integers(bw, 60 * bw)) if rng.random() < 0.5 else int(rng.integers(1, 300))
    u = rng.random()
    if u < 0.28:                                 # several sort windows, the last one partly filled
        H, W = int(rng.integers(100 * bh, 260 * bh)), int(rng.integers(100 * bw, 260 * bw))
    elif u < 0.32:                               # 0.1 .. 0.6 M vectors of any length, any codebook size
        H, W = int(rng.integers(300 * bh, 780 * bh)), int(rng.integers(300 * bw, 780 * bw))
    elif u < 0.36 and d <= 48:                   # up to a few million vectors: launches of many rounds of workgroups
        H, W = int(rng.integers(400 * bh, 1800 * bh)), int(rng.integers(400 * bw, 1800 * bw))
        K = min(K, 1500)
    if rng.random() < 0.4:                       # aligned sizes take the fast paths
        H, W = (H // (4 * bh) + 1) * 4 * bh, (W // 16 + 1) * 16
    kind = rng.integers(0, 3)
    if kind == 0:
        img = rng.integers(0, 256, size=(H, W, C), dtype=np.uint8)
    else:
        yy, xx = np.mgrid[0:H, 0:W]
        base = 128 + 100 * np.sin(xx / 17.0 + rng.random() * 6) * np.cos(yy / 11.0) + rng.normal(0, 4 if kind == 1 else 0.5, (H, W))
        img = np.clip(np.stack([base + 9 * c for c in range(C)], -1), 0, 255).astype(np.uint8)
    timg = torch.from_numpy(img).to(dev)
    pad = int(rng.integers(0, 2))                # edge replication (VQ_Full.py:23-33) or zeros (old/CodeBook_VQ.py)
    blocks = eng.blocks_from_image(timg, bh, bw, pad)
    # the block matrix itself, against plain tensor indexing
    nr, nc = -(-H // bh), -(-W // bw)
    padded = torch.nn.functional.pad(timg.permute(2, 0, 1)[None].float(), (0, nc * bw - W, 0, nr * bh - H),
                                     mode="replicate" if pad == 0 else "constant")[0].permute(1, 2, 0).to(torch.uint8)
    ref_blocks = padded.view(nr, bh, nc, bw, C).permute(0, 2, 1, 3, 4).reshape(nr * nc, d)
    assert torch.equal(blocks, ref_blocks), f"blocks differ, H={H} W={W} C={C} {bh}x{bw} pad={pad}"
    N = blocks.shape[0]
    b_np = blocks.cpu().numpy()
    pick = rng.integers(0, N, K)
    cb = b_np[pick].astype(np.float64) + rng.choice([0.0, 0.0, 0.25, -0.5, 0.001], size=(K, 1)) + rng.normal(0, 0.3, (K, d)) * (rng.random() < 0.5)
    if K > 3:
        cb[K // 2] = cb[0]                       # duplicate row
    tcb = torch.from_numpy(cb).to(dev)
    want_l, want_d = eng.encode(blocks, tcb, want_dist=True, mode=0)
    tag = f"case {cases}: H={H} W={W} C={C} {bh}x{bw} d={d} K={K} N={N} kind={kind} pad={pad}"
    got_l, got_d = eng.encode(blocks, tcb, want_dist=True, mode=1)
    assert torch.equal(got_l, want_l) and torch.equal(got_d, want_d), "plain FAST differs, " + tag
    srt = eng.sort_blocks(blocks)
    if srt is not None:
        got_l, got_d = eng.encode(blocks, tcb, want_dist=True, mode=1, sorted=srt)
        assert torch.equal(got_l, want_l) and torch.equal(got_d, want_d), "pruned FAST differs, " + tag
    got_l, got_d = eng.encode_image(timg, bh, bw, tcb, pad, want_dist=True)
    assert torch.equal(got_l, want_l) and torch.equal(got_d, want_d), "image-fused differs, " + tag
    srt2, _ = eng.sort_image(timg, bh, bw, pad)
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez("gpurun_out/fuzz_last_case.npz", img=img, cb=cb, bh=bh, bw=bw)   # what to look at after a failure
    acc, dist, lab = eng.lbg_accum(None if (srt2 is not None and eng.prune_supported(K, d)) else blocks, tcb, want_labels=True,
                                   sorted=srt2, shape=(N, d))
    assert torch.equal(lab, want_l), "LBG labels differ, " + tag
    sums = torch.zeros(K, d, dtype=torch.int64, device=dev).index_add_(0, want_l.long(), blocks.long())
    counts = torch.bincount(want_l.long(), minlength=K)
    assert torch.equal(acc[: K * d].view(K, d), sums) and torch.equal(acc[K * d:], counts), "LBG sums differ, " + tag
    dec = eng.decode(want_l.view(-(-H // bh), -(-W // bw)), tcb, bh, bw, C)
    ref = tcb.trunc().to(torch.int64).remainder(256).to(torch.uint8)[want_l.long()].view(-(-H // bh), -(-W // bw), bh, bw, C)
    ref = ref.permute(0, 2, 1, 3, 4).reshape(dec.shape)
    assert torch.equal(dec, ref), "decode differs, " + tag
    # the bit-packed label stream (VQ_Full.py:134-147) and back
    packed = eng.pack_labels(want_l, K)
    bits = 0 if K <= 1 else int(K - 1).bit_length()
    if bits:
        lab_np = want_l.cpu().numpy().astype(np.uint64)
        stream_bits = ((lab_np[:, None] >> np.arange(bits - 1, -1, -1, dtype=np.uint64)[None, :]) & 1).astype(np.uint8).reshape(-1)
        assert np.array_equal(packed.cpu().numpy(), np.packbits(stream_bits)), "packed labels differ, " + tag
    else:
        assert packed.numel() == 0, "packed labels of K = 1 are not empty, " + tag
    assert torch.equal(eng.unpack_labels(packed, N, K), want_l), "unpacked labels differ, " + tag
    # the entry point that allocates nothing
    lab_s = torch.full((N,), -7, dtype=torch.int32, device=dev)
    dist_s = torch.full((N,), -1.0, dtype=torch.float64, device=dev)
    rc = lib.vq_encode_l1_static(P(blocks), N, d, P(tcb), K, P(lab_s), P(dist_s), 1, STREAM())
    if rc == 0:
        assert torch.equal(lab_s, want_l) and torch.equal(dist_s, want_d), "static-scratch search differs, " + tag
    else:
        assert rc == VQ_E_UNSUPPORTED, f"vq_encode_l1_static: {last_error()}, " + tag
    # a training stage: FAST on the sorted copy against the all-fp64 search on the block matrix, three iterations
    if K >= 2 and N >= K:
        cb0 = torch.from_numpy(b_np[pick].astype(np.float64) * (1.0 + 0.01 * rng.choice([-1.0, 1.0], size=(K, 1)))).to(dev)
        want_cb, want_D, want_conv = eng.lbg_stage(blocks, cb0, N * d, 0.001, 3, mode=0)
        got_cb, got_D, got_conv = eng.lbg_stage(None if (srt2 is not None and eng.prune_supported(K, d)) else blocks, cb0, N * d,
                                                0.001, 3, sorted=srt2, shape=(N, d), mode=1)
        assert torch.equal(got_cb, want_cb) and got_D == want_D and got_conv == want_conv, "training stage differs, " + tag
        if group is not None:                    # the two-halves form of the iteration, collectives of a one-rank group between
            aligned, sizes = shards_chunk_aligned(N, group, dev)
            for limit_bytes in (1 << 20, 0):     # both exchange routes
                got_cb, got_D, got_conv = eng.lbg_stage_group(blocks, cb0, N * d, 0.001, 3, group, sizes, aligned, sorted=srt2,
                                                              shape=(N, d), mode=1, allreduce_from_bytes=limit_bytes)
                assert torch.equal(got_cb, want_cb) and got_D == want_D and got_conv == want_conv, \
                    f"training stage in two halves (route {limit_bytes}) differs, " + tag
    # now and then a whole training run (VQ_Full.py:60-98): FAST engine against the all-fp64 engine
    if cases % 7 == 0 and N >= 8:
        k = int(rng.choice([2, 3, 8, 20, 64, 130])) if N < 200000 else int(rng.choice([8, 64]))
        k = min(k, N)
        got = train_lbg(eng, blocks, k, max_iterations=6, log=None)
        want = train_lbg(eng_exact, blocks, k, max_iterations=6, log=None)
        assert torch.equal(got.codebook, want.codebook) and got.distortions == want.distortions and \
            got.iters_per_stage == want.iters_per_stage, "training run differs, " + tag + f" k={k}"
    cases += 1
    if cases % 20 == 0:
        print(f"{cases} cases ok, last: {tag}", flush=True)
print(f"fuzz ok: {cases} cases in {budget:.0f} s")
if group is not None:
    torch.distributed.destroy_process_group()
