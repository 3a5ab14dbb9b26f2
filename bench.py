#!/usr/bin/env python3
"""Benchmark of the VQ hot path on MI355X.

Workload (the headline configuration): one synthetic 8192 x 8192 grayscale image per GPU, 4x4 blocks
(N = 4 194 304 block vectors of dim 16), K = 256 codebook trained by LBG on a 2048 x 2048 crop.
One step = encode every block vector (nearest codeword, exact labels) + decode the labels back
to an image, inputs resident in HBM.  `value` = block vectors through that step per second,
whole job.  Extra keys report the encode kernel alone, decode alone, and LBG iterations/s
(fused assign + accumulate, all-reduce of sums/counts when N > 1, centroid update).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 8192
BH = BW = 4
K_CODEBOOK = 256
PEAK_VALU_TOPS = 78.6       # MI355X fp32 vector peak 157.3 TFLOP/s counts FMA as 2; non-FMA ops/s = half
PEAK_HBM_GBS = 8000.0


def structured_image(h, w, seed, device):
    """Smooth waves plus Gaussian noise, clipped to uint8 (the 'structured' synthetic image)."""
    g = torch.Generator(device=device).manual_seed(seed)
    y = torch.arange(h, device=device, dtype=torch.float32)[:, None]
    x = torch.arange(w, device=device, dtype=torch.float32)[None, :]
    v = (128 + 64 * torch.sin(2 * np.pi * x / 257) * torch.cos(2 * np.pi * y / 181)
         + 32 * torch.sin(2 * np.pi * (x + y) / 1021) + 16 * torch.randn((h, w), device=device, generator=g))
    return v.clamp(0, 255).to(torch.uint8)


def cpu_baseline(blocks_sample, cb, n_decode):
    """Reference arithmetic on the host: scipy cdist('cityblock') + argmin in 65 536-vector chunks
    (VQ_Full.py:124-125) and the decode loop restated in C (VQ_Full.py:157-169), one core."""
    from oracle import vq_oracle as o
    t0 = time.perf_counter()
    labels = o.scipy_encode_chunked(blocks_sample, cb, chunk=65536)
    t_enc = time.perf_counter() - t0
    side = int(np.sqrt(n_decode))
    grid = labels[: side * side].astype(np.int32).reshape(side, side)
    t0 = time.perf_counter()
    o.decode_u8(grid, cb, BH, BW, 1)
    t_dec = time.perf_counter() - t0
    n = blocks_sample.shape[0]
    per_vec = t_enc / n + t_dec / (side * side)
    return dict(value=1.0 / per_vec, unit="block-vectors/s", cores=1, kind="port",
                sample=f"{n} of the workload's block vectors: scipy cdist+argmin in 65536-vector chunks "
                       f"({t_enc:.2f} s) + C restatement of the decode loop on {side * side} labels ({t_dec:.3f} s)",
                encode_only=n / t_enc), labels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=1 << 21, help="block vectors timed on the host (0 = skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if args.gpus != world:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    group = None
    if distributed:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)
        group = dist.group.WORLD

    from vector_quantization_amd import HipEngine, train_lbg, lbg_iteration
    from vector_quantization_amd.lbg import shards_chunk_aligned
    eng = HipEngine(device)

    def barrier():
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)

    # ---- untimed set-up: image, block vectors, a trained codebook ------------------------------
    img = structured_image(H, W, seed=1000 + rank, device=device)
    blocks = eng.blocks_from_image(img, BH, BW)
    N, d = blocks.shape
    crop = eng.blocks_from_image(structured_image(2048, 2048, seed=0, device=device), BH, BW)
    cb = train_lbg(eng, crop, K_CODEBOOK, log=None).codebook          # same codebook on every rank (seed 0)
    prepared = eng.prepare_codebook(cb, blocks.shape[0])
    labels = torch.empty(N, dtype=torch.int32, device=device)
    nr, nc = H // BH, W // BW
    sorted_buf = eng.sort_blocks(blocks)                                 # allocated once, refilled every step

    def encode():
        # the window sort is part of the encode: the step starts from the plain block matrix
        srt = eng.sort_blocks(blocks, out=sorted_buf)
        eng.encode_prepared(blocks, prepared, labels=labels, sorted=srt)

    def step():
        encode()
        return eng.decode(labels.view(nr, nc), cb, BH, BW, 1)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if distributed:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t.item())
    value = world * N * args.steps / elapsed

    # ---- the dominant kernel alone, HIP events on the launch stream ------------------------------
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in ev:
        a.record()
        encode()
        b.record()
    torch.cuda.synchronize(device)
    enc_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    labels_pruned = labels.clone()
    # the two parts of the encode on their own: window sort (HBM-bound), pruned search + fix-up (VALU-bound)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in ev:
        a.record()
        eng.sort_blocks(blocks, out=sorted_buf)
        b.record()
    torch.cuda.synchronize(device)
    sort_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in ev:
        a.record()
        eng.encode_prepared(blocks, prepared, labels=labels, sorted=sorted_buf)
        b.record()
    torch.cuda.synchronize(device)
    search_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in ev:
        a.record()
        eng.encode_prepared(blocks, prepared, labels=labels)              # every codeword for every vector
        b.record()
    torch.cuda.synchronize(device)
    enc_full_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    full_equals_pruned = bool(torch.equal(labels, labels_pruned))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in ev:
        a.record()
        eng.decode(labels.view(nr, nc), cb, BH, BW, 1)
        b.record()
    torch.cuda.synchronize(device)
    dec_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    ops = 2.0 * K_CODEBOOK * d * N                  # subtract + |.|-accumulate per (vector, codeword, element)
    achieved = ops / (search_ms * 1e-3) / 1e12
    # HBM bytes per launch of the same kernels on the same workload, from separate rocprofv3 --pmc passes
    # (FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE as is); see profiles/
    traffic, pmc, pmc_note = None, None, "no PMC summary found under profiles/"
    pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_encode_summary.json")
    if os.path.isfile(pmc_path):
        with open(pmc_path) as f:
            pmc = json.load(f)
        by_kernel = pmc["hbm_bytes_by_kernel"]
        traffic = sum(v for k, v in by_kernel.items() if not k.startswith("k_window_sort"))
        pmc_note = ("traffic (search + fix-up launches), valu_busy_frac and evaluated_pair_frac from profiles/r01_pmc_*.csv "
                    f"(rocprofv3 --pmc, same workload); algorithmic bytes per launch = {N * (d + 4 + 2)} "
                    "(row + label + 2-byte origin per vector)")
    search_alg_bytes = N * (d + 4 + 2)
    roofline = dict(bound="valu", achieved=achieved, peak=PEAK_VALU_TOPS, unit="TFLOP/s",
                    frac=achieved / PEAK_VALU_TOPS, traffic=traffic,
                    kernel="k_assign_fast<8,4,packed,pruned> + k_fixup (the two launches of vq_encode_l1_prepared "
                           "on window-sorted vectors)",
                    kernel_ms=search_ms, full_search_ms=enc_full_ms,
                    full_search_achieved=ops / (enc_full_ms * 1e-3) / 1e12,
                    full_search_frac=ops / (enc_full_ms * 1e-3) / 1e12 / PEAK_VALU_TOPS,
                    full_search_labels_equal=full_equals_pruned,
                    algorithmic_ops_per_vector=2 * K_CODEBOOK * d,
                    algorithmic_bytes_per_vector=d + 4 + 2,
                    valu_busy_frac=(pmc["full_search"]["valu_busy_frac"] if pmc else None),
                    evaluated_pair_frac=(pmc["pruned_search"]["share_of_full_search_valu"] if pmc else None),
                    note="ops = one subtract + one |.|-accumulate per (vector, codeword, element), counted for ALL K*N "
                         "pairs; peak = the fp32 vector peak counted without FMA (157.3 / 2). Two effects put frac "
                         "above 1: the integer v_sad_u16 retires two such pairs per lane-instruction, and the search "
                         "skips every codeword whose element sum is provably too far from the vector's "
                         "(|sum x - sum c| <= L1 distance; vectors sorted by element sum inside windows of 8192, "
                         "codebook sorted by element sum), so it executes only evaluated_pair_frac of the "
                         "instructions of the full search while returning the same labels. full_search_* is the "
                         "same launch pair without that pruning and valu_busy_frac the share of SIMD issue cycles "
                         "its VALU instructions occupy. " + pmc_note)
    sort_alg_bytes = N * (2 * d + 2)
    sort = dict(bound="hbm", kernel="k_window_sort (vq_blocks_sort)", kernel_ms=sort_ms,
                achieved=sort_alg_bytes / (sort_ms * 1e-3) / 1e9, peak=PEAK_HBM_GBS, unit="GB/s",
                frac=sort_alg_bytes / (sort_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                traffic=(sum(v for k, v in pmc["hbm_bytes_by_kernel"].items() if k.startswith("k_window_sort"))
                         if pmc else None),
                algorithmic_bytes_per_vector=2 * d + 2,
                note="reads every row once, writes it once at its sorted place plus a 2-byte origin; the scattered "
                     "16-byte stores are the write amplification seen in traffic")
    decode_gbs = (N * (d + 4)) / (dec_ms * 1e-3) / 1e9

    # ---- LBG iterations (fused assign + accumulate, exchange, centroid update) ----------------------
    n_total = world * N
    cur = cb
    # every rank's shard size, exchanged once (train_lbg does the same before its loop)
    shard_info = shards_chunk_aligned(N, group) if distributed else None
    for _ in range(2):
        cur, _ = lbg_iteration(eng, blocks, cur, n_total, group, shard_info, sorted=sorted_buf)
    barrier()
    t0 = time.perf_counter()
    lbg_steps = max(5, args.steps // 2)
    for _ in range(lbg_steps):
        # the vectors do not change between iterations: the sort made above serves all of them
        cur, _ = lbg_iteration(eng, blocks, cur, n_total, group, shard_info, sorted=sorted_buf)
    barrier()
    lbg_elapsed = time.perf_counter() - t0
    t = torch.tensor([lbg_elapsed], dtype=torch.float64, device=device)
    if distributed:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    lbg_elapsed = float(t.item())

    result = {
        "metric": "block-vectors encoded/sec (encode+decode step; LBG iter/sec in `lbg`) at K=256,dim=16",
        "value": value,
        "unit": "block-vectors/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u16 fixed-point SAD filter + f64 re-check (labels identical to the all-f64 search)",
        "data": "synthetic",
        "config": {"workload": "configs[1]: synthetic 8192x8192 grayscale per GPU, 4x4 blocks (dim 16), K=256, "
                               "encode+decode", "vectors_per_gpu": N, "K": K_CODEBOOK, "dim": d,
                   "parallelism": f"dp{world} (block vectors sharded by image, codebook replicated)"},
        "encode_vectors_per_s_per_gpu": N / (enc_ms * 1e-3),
        "decode_vectors_per_s_per_gpu": N / (dec_ms * 1e-3),
        "decode_gbs": decode_gbs,
        "decode_frac_of_hbm_peak": decode_gbs / PEAK_HBM_GBS,
        "lbg": {"iters_per_s": lbg_steps / lbg_elapsed, "vectors_per_s": n_total * lbg_steps / lbg_elapsed,
                "ms_per_iter": lbg_elapsed / lbg_steps * 1e3, "vectors_total": n_total,
                "exchange": "one all-gather of int64 [K*d sums | K counts | fp64 per-chunk distance sums] per iteration" if distributed
                else "none (1 GPU)"},
        "encode_ms": enc_ms,
        "roofline": roofline,
        "sort": sort,
    }

    if rank == 0 and world == 1 and args.cpu_sample > 0:
        n_s = min(args.cpu_sample, N)
        sample = blocks[:n_s].cpu().numpy()
        cbh = cb.cpu().numpy()
        base, cpu_labels = cpu_baseline(sample, cbh, min(n_s, 1 << 20))
        result["cpu_baseline"] = base
        result["cpu_labels_match_gpu"] = bool(np.array_equal(cpu_labels, labels[:n_s].cpu().numpy().astype(np.int64)))
    else:
        result["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(result))
    if distributed:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
