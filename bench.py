#!/usr/bin/env python3
"""Benchmark of the VQ hot path on MI355X.

Workload (the headline configuration, configs[1] of the baseline description): one synthetic 8192 x 8192 grayscale image
per GPU, 4x4 blocks (N = 4 194 304 block vectors of dim 16), K = 256 codebook trained by LBG on a
2048 x 2048 crop.  One step starts from the IMAGE resident in HBM: image -> window-sorted block vectors
(padding + block extraction + sort, one pass), codebook preparation, pruned nearest-codeword search with
its near-ties re-checked in fp64 inside the same kernel (exact labels), decode of the labels back to an image.  `value` = block vectors
through that step per second, whole job.

Extra keys: the dominant kernel's roofline (executed SAD work against the v_sad_u16 issue ceiling, never
above 1; the pruning gain is a separate key), the HBM-bound kernels against the HBM roofline, LBG
iterations/s (fused assign + accumulate, exchange of sums/counts when N > 1, centroid update), a full
LBG training at the size of configs[2], other shapes (iid worst case, K = 1024, RGB d = 48, d = 192 with
K = 4096), and the reference's CPU arithmetic timed on this box's host cores (1 core and P processes).

    python bench.py --gpus 1 --steps 200 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 8192
BH = BW = 4
K_CODEBOOK = 256
# v_sad_u16 issues once per 4 cycles per SIMD for a 64-lane wave and retires two |x - c| accumulations per
# lane (= 4 algorithmic ops: 2 subtractions, 2 absolute-value accumulations):
# 256 CUs x 4 SIMDs x 64 lanes x 2.4 GHz / 4 cycles x 4 ops = 157.3 T ops/s
PEAK_SAD_TOPS = 256 * 4 * 64 * 2.4e9 / 4 * 4 / 1e12
PEAK_HBM_GBS = 8000.0


# ---------------------------------------------------------------------------------------------------
# CPU baseline (runs in a child process that never touches the GPU)
# ---------------------------------------------------------------------------------------------------
def _cpu_shard(args):
    from oracle import vq_oracle as o
    blocks, cb = args
    return o.scipy_encode_chunked(blocks, cb, chunk=65536)


def _cpu_lbg_shard(args):
    """Per-cluster integer sums / counts and the distance partial of one shard (the reference arithmetic of
    VQ_Full.py:77-90 up to the division)."""
    from oracle import vq_oracle as o
    blocks, cb = args
    labels, mind = o.scipy_encode_chunked(blocks, cb, chunk=65536, want_dist=True)
    K, d = cb.shape
    sums = np.zeros((K, d), dtype=np.int64)
    np.add.at(sums, labels, blocks.astype(np.int64))
    counts = np.bincount(labels, minlength=K).astype(np.int64)
    return sums, counts, float(mind.sum())


def cpu_lbg_main(path):
    """One LBG iteration of the reference arithmetic (chunked cdist + argmin + per-cluster mean, VQ_Full.py:77-90)
    on the sample: one core, then sharded over P worker processes with the integer sums reduced on the parent."""
    import multiprocessing as mp
    from oracle import vq_oracle as o
    z = np.load(path)
    blocks, cb = z["blocks"], z["cb"]
    n = blocks.shape[0]
    t0 = time.perf_counter()
    new_cb, D = o.scipy_lbg_iter_chunked(blocks, cb, chunk=65536)
    t1 = time.perf_counter() - t0
    procs = max(1, min(16, os.cpu_count() or 1))
    shards = [(blocks[i * n // procs:(i + 1) * n // procs], cb) for i in range(procs)]
    with mp.get_context("fork").Pool(procs) as pool:
        pool.map(_cpu_lbg_shard, [(blocks[:1024], cb)] * procs)
        t0 = time.perf_counter()
        parts = pool.map(_cpu_lbg_shard, shards)
        sums = sum(p[0] for p in parts)
        counts = sum(p[1] for p in parts)
        par_cb = np.where(counts[:, None] > 0, sums / np.maximum(counts, 1)[:, None], cb)
        tp = time.perf_counter() - t0
    np.save(path + ".newcb.npy", new_cb)
    print(json.dumps(dict(
        value=n / t1, unit="block-vectors/s per LBG iteration", iterations_per_s_at_sample=1.0 / t1, cores=1, kind="port",
        seconds=t1, distortion=float(D),
        sample=f"one iteration over {n} of the workload's block vectors, K={cb.shape[0]}, d={cb.shape[1]}: scipy cdist + "
               f"argmin in 65536-vector chunks, per-cluster np.mean, np.mean of the winning distances; one core",
        all_cores=dict(value=n / tp, unit="block-vectors/s per LBG iteration", cores=procs, seconds=tp,
                       codebook_equal_one_core=bool(np.array_equal(par_cb, new_cb)),
                       sample=f"the same vectors sharded over {procs} worker processes, integer sums / counts reduced "
                              f"on the parent, one divide"))))


def cpu_encode_sample_main(path):
    """Encode-only figure on a sample of another shape (K = 1024 at d = 16, K = 4096 at d = 192): one core."""
    from oracle import vq_oracle as o
    z = np.load(path)
    blocks, cb = z["blocks"], z["cb"]
    n = blocks.shape[0]
    t0 = time.perf_counter()
    labels = o.scipy_encode_chunked(blocks, cb, chunk=8192 if cb.shape[0] > 1024 else 65536)
    t1 = time.perf_counter() - t0
    np.save(path + ".labels.npy", labels)
    print(json.dumps(dict(value=n / t1, unit="block-vectors/s (encode only)", cores=1, kind="port", seconds=t1,
                          sample=f"{n} block vectors, K={cb.shape[0]}, d={cb.shape[1]}: scipy cdist + argmin, one core; "
                                 f"throughput is per vector, so larger images scale linearly")))


def cpu_baseline_main(path):
    """Reference arithmetic on the host: scipy cdist('cityblock') + argmin in 65 536-vector chunks
    (VQ_Full.py:124-125) and the decode loop restated in C (VQ_Full.py:157-169); one core, then the same
    encode sharded over P worker processes."""
    import multiprocessing as mp
    from oracle import vq_oracle as o
    z = np.load(path)
    blocks, cb = z["blocks"], z["cb"]
    n = blocks.shape[0]
    t0 = time.perf_counter()
    labels = o.scipy_encode_chunked(blocks, cb, chunk=65536)
    t_enc = time.perf_counter() - t0
    side = int(np.sqrt(min(n, 1 << 20)))
    grid = labels[: side * side].astype(np.int32).reshape(side, side)
    t0 = time.perf_counter()
    o.decode_u8(grid, cb, BH, BW, 1)
    t_dec = time.perf_counter() - t0
    per_vec = t_enc / n + t_dec / (side * side)
    procs = max(1, min(16, os.cpu_count() or 1))
    shards = [(blocks[i * n // procs:(i + 1) * n // procs], cb) for i in range(procs)]
    with mp.get_context("fork").Pool(procs) as pool:
        pool.map(_cpu_shard, [(blocks[:1024], cb)] * procs)          # start the workers, import scipy
        t0 = time.perf_counter()
        parts = pool.map(_cpu_shard, shards)
        t_par = time.perf_counter() - t0
    same = bool(np.array_equal(np.concatenate(parts), labels))
    np.save(path + ".labels.npy", labels)
    print(json.dumps(dict(
        value=1.0 / per_vec, unit="block-vectors/s", cores=1, kind="port",
        sample=f"{n} of the workload's block vectors: scipy cdist+argmin in 65536-vector chunks ({t_enc:.2f} s) "
               f"+ C restatement of the decode loop on {side * side} labels ({t_dec:.3f} s), one core",
        encode_only=n / t_enc,
        all_cores=dict(value=n / t_par, unit="block-vectors/s (encode only)", cores=procs, seconds=t_par,
                       labels_equal_one_core=same,
                       sample=f"the same {n} vectors sharded over {procs} worker processes (reference arithmetic, "
                              f"P = {procs} processes; this box exposes {os.cpu_count()} CPUs)"))))


def run_cpu_baseline(blocks_h, cb_h, leg="encode"):
    """leg: "encode" (encode + decode, the headline metric), "lbg" (one LBG iteration) or "sample" (encode only, other
    shapes).  Returns (the child's JSON, labels or new codebook as numpy)."""
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "cpu_sample.npz")
        np.savez(path, blocks=blocks_h, cb=cb_h)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-file", path, "--cpu-leg", leg],
                             check=True, capture_output=True, text=True)
        base = json.loads(out.stdout.strip().splitlines()[-1])
        extra = np.load(path + (".newcb.npy" if leg == "lbg" else ".labels.npy"))
    return base, extra


# ---------------------------------------------------------------------------------------------------
# GPU side
# ---------------------------------------------------------------------------------------------------
def structured_image(h, w, seed, device, channels=1):
    """Smooth waves plus Gaussian noise, clipped to uint8 (the 'structured' synthetic image); with three
    channels the waves are phase-shifted per channel."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    y = torch.arange(h, device=device, dtype=torch.float32)[:, None]
    x = torch.arange(w, device=device, dtype=torch.float32)[None, :]
    planes = []
    for c in range(channels):
        ph = 0.7 * c
        v = (128 + 64 * torch.sin(2 * np.pi * x / 257 + ph) * torch.cos(2 * np.pi * y / 181 - ph)
             + 32 * torch.sin(2 * np.pi * (x + y) / 1021 + 2 * ph)
             + 16 * torch.randn((h, w), device=device, generator=g))
        planes.append(v.clamp(0, 255).to(torch.uint8))
    return planes[0] if channels == 1 else torch.stack(planes, dim=-1).contiguous()


def iid_image(h, w, seed, device):
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    return torch.randint(0, 256, (h, w), device=device, generator=g, dtype=torch.uint8)


PREROLL_MS = 25.0


def preroll(fn, device, ms=PREROLL_MS):
    """Untimed set-up before a measurement: fn() back to back for `ms` of wall time.  After any pause the card needs
    about 13 ms of continuous work to reach its sustained clock (tools/clock_ramp.py: the encode step takes 0.18 - 0.22 ms
    right after a pause of 50 ms or more and 0.169 ms from the 80th consecutive step on); a measurement of a few
    milliseconds taken cold reads 5 % low."""
    import torch
    t_end = time.perf_counter() + ms * 1e-3
    while time.perf_counter() < t_end:
        for _ in range(16):
            fn()
        torch.cuda.synchronize(device)


def event_ms(fn, reps, device):
    import torch
    preroll(fn, device)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize(device)
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))


def _nbytes_sorted(eng, d, N):
    from vector_quantization_amd import _lib
    return _lib.lib.vq_sorted_bytes(int(d), int(N))


def search_profile(eng, img, bh, bw, cb, reps, device, check_vectors=1 << 18):
    """Timings and statistics of the search on one (image, codebook): the filter kernel alone (HIP events on
    its stream), the fix-up alone, the evaluated share of all (vector, codeword) pairs."""
    import torch
    C = 1 if img.dim() == 2 else img.shape[2]
    K, d = cb.shape
    srt, (N, d2) = eng.sort_image(img, bh, bw)
    assert d2 == d
    blocks = None
    pruned = srt is not None and eng.prune_supported(K, d)
    if not pruned:
        blocks = eng.blocks_from_image(img, bh, bw)
        srt = None
    prepared = eng.prepare_codebook(cb, N)
    labels = torch.empty(N, dtype=torch.int32, device=device)
    kw = dict(labels=labels, sorted=srt, shape=(N, d))
    for _ in range(2):
        eng.encode_prepared(blocks, prepared, **kw)
    # the filter kernel alone: HIP events recorded by the library right around that launch, on its stream
    preroll(lambda: eng.encode_prepared(blocks, prepared, phases=1, **kw), device)
    filter_ms = eng.search_kernel_ms(lambda: eng.encode_prepared(blocks, prepared, phases=1, **kw), reps)
    queued, pairs_eval = eng.search_statistics(prepared, N)
    in_filter = bool(pruned and eng.prune_resolves_in_filter(K, d))     # near-ties settled inside the filter kernel
    fixup_ms = 0.0 if in_filter else event_ms(lambda: eng.encode_prepared(blocks, prepared, phases=2, **kw), reps, device)
    both_ms = event_ms(lambda: eng.encode_prepared(blocks, prepared, **kw), reps, device)
    sort_ms = event_ms(lambda: eng.sort_image(img, bh, bw, out=srt if pruned else None), reps, device) if pruned else None
    prep_ms = event_ms(lambda: eng.prepare_codebook(cb, 0), reps, device)
    e2e_scratch = torch.empty(max(int(_nbytes_sorted(eng, d, N)), N * d, 16), dtype=torch.uint8, device=device)
    e2e_ms = event_ms(lambda: eng.encode_image(img, bh, bw, cb, scratch=e2e_scratch, labels=labels), reps, device)
    # labels against the all-fp64 search (every distance in sequential fp64 on the device) on a sample of the vectors
    from vector_quantization_amd import VQ_MODE_EXACT
    eng.encode_prepared(blocks, prepared, **kw)
    bl = blocks if blocks is not None else eng.blocks_from_image(img, bh, bw)
    ns = min(N, check_vectors)
    labels_equal = bool(torch.equal(eng.encode(bl[:ns], cb, mode=VQ_MODE_EXACT), labels[:ns]))
    if blocks is None:
        del bl
    pairs_all = float(N) * K
    pairs_eval = float(pairs_eval)
    ops_exec = 2.0 * d * pairs_eval
    out = dict(K=K, d=d, N=N, pruned_search=bool(pruned), near_ties_settled_in_filter=in_filter,
               filter_kernel_ms=filter_ms, fixup_kernel_ms=fixup_ms,
               search_ms=both_ms, vectors_per_s=N / (both_ms * 1e-3), image_sort_ms=sort_ms, codebook_prepare_ms=prep_ms,
               image_to_labels_ms=e2e_ms, image_to_labels_vectors_per_s=N / (e2e_ms * 1e-3),
               queued_for_fp64_frac=queued / N,
               evaluated_pair_frac=pairs_eval / pairs_all,
               executed_tops=ops_exec / (filter_ms * 1e-3) / 1e12,
               frac_of_sad_ceiling=ops_exec / (filter_ms * 1e-3) / 1e12 / PEAK_SAD_TOPS,
               algorithmic_tops=2.0 * d * pairs_all / (filter_ms * 1e-3) / 1e12,
               labels_equal=labels_equal, labels_checked=f"first {ns} vectors against the all-fp64 search")
    return out, labels, srt, prepared, blocks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--lanes", type=int, default=6,
                    help="HIP streams that consecutive steps alternate between (1 = every step on one stream)")
    ap.add_argument("--cpu-sample", type=int, default=1 << 22, help="block vectors timed on the host (0 = skip)")
    ap.add_argument("--no-shapes", action="store_true", help="skip the other-shape and full-training measurements")
    ap.add_argument("--cpu-baseline-file", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-leg", default="encode", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_file:
        {"encode": cpu_baseline_main, "lbg": cpu_lbg_main, "sample": cpu_encode_sample_main}[args.cpu_leg](args.cpu_baseline_file)
        return

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # (VQ_BENCH_FORCE_GROUP=1 under torchrun with one process: the multi-GPU code path -- RCCL collectives between the
    # two halves of every LBG iteration -- on a single GPU, to measure what the exchange machinery itself costs)
    distributed = world > 1 or os.environ.get("VQ_BENCH_FORCE_GROUP") == "1"
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    group = None
    if distributed:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)
        group = dist.group.WORLD

    from vector_quantization_amd import HipEngine, EncodePipeline, train_lbg, lbg_iteration
    from vector_quantization_amd import _lib
    from vector_quantization_amd.lbg import shards_chunk_aligned
    eng = HipEngine(device)

    def barrier():
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)

    # ---- untimed set-up: the image and a trained codebook ---------------------------------------------
    img = structured_image(H, W, seed=1000 + rank, device=device)
    crop = eng.blocks_from_image(structured_image(2048, 2048, seed=0, device=device), BH, BW)
    cb = train_lbg(eng, crop, K_CODEBOOK, log=None).codebook          # same codebook on every rank (seed 0)
    nr, nc = H // BH, W // BW
    N, d = nr * nc, BH * BW
    labels = torch.empty(N, dtype=torch.int32, device=device)
    scratch = torch.empty(_lib.lib.vq_sorted_bytes(d, N), dtype=torch.uint8, device=device)

    # A step: image -> sorted block vectors -> prepared codebook -> pruned search (near-ties re-checked in fp64 inside
    # the same kernel) -> labels -> decoded image.  Consecutive steps go to different HIP streams (each lane with its own
    # workspace, sorted buffer, labels and output), so that the byte-moving kernels of one step and the start of its
    # search run in what the search of the step before leaves free; every step does all of its work inside the timed
    # region, which ends with a device-wide synchronisation.
    pipe = EncodePipeline(device, lanes=max(1, args.lanes))

    def step():
        return pipe.submit(img, BH, BW, cb, inputs_ready=True)    # (resident before the timed region starts)

    # untimed set-up: every lane allocates its buffers and runs once, and the card is brought to its sustained clock
    # (see preroll); then the W warm-up steps and the K timed ones
    preroll(step, device)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if distributed:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t.item())
    value = world * N * args.steps / elapsed
    labels_step = last.labels.clone()

    # the same steps one after the other on one stream (the latency of a step rather than the throughput of many)
    def step_sequential():
        eng.encode_image(img, BH, BW, cb, scratch=scratch, labels=labels)
        return eng.decode(labels.view(nr, nc), cb, BH, BW, 1)

    preroll(step_sequential, device)
    for _ in range(args.warmup):
        step_sequential()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_sequential()
    barrier()
    sequential_ms = (time.perf_counter() - t0) / args.steps * 1e3
    sequential_labels_equal = bool(torch.equal(labels, labels_step))

    # ---- the kernels of the step on their own, HIP events on the launch stream ---------------------------
    reps = max(20, min(args.steps, 200))
    prof, labels_p, srt, prepared, _ = search_profile(eng, img, BH, BW, cb, reps, device)
    step_labels_equal = bool(torch.equal(labels_p, labels_step))
    # the same launch pair without pruning (every codeword for every vector), as a cross-check of labels and as
    # the measure of what pruning saves
    blocks = eng.blocks_from_image(img, BH, BW)
    lab_full = torch.empty_like(labels)
    preroll(lambda: eng.encode_prepared(blocks, prepared, labels=lab_full, phases=1), device)
    full_ms = eng.search_kernel_ms(lambda: eng.encode_prepared(blocks, prepared, labels=lab_full, phases=1), reps)
    _, pairs_full = eng.search_statistics(prepared, N)
    eng.encode_prepared(blocks, prepared, labels=lab_full)
    full_equals_pruned = bool(torch.equal(lab_full, labels_step))
    full_ops = 2.0 * d * pairs_full
    sort_ms = event_ms(lambda: eng.sort_image(img, BH, BW, out=scratch), reps, device)
    prep_ms = event_ms(lambda: eng.prepare_codebook(cb, 0), reps, device)
    dec_ms = event_ms(lambda: eng.decode(labels.view(nr, nc), cb, BH, BW, 1), reps, device)
    enc_ms = event_ms(lambda: eng.encode_image(img, BH, BW, cb, scratch=scratch, labels=labels), reps, device)

    # HBM bytes per launch from separate rocprofv3 --pmc passes of this library build, when a summary made
    # from the same libvqhip.so is committed under profiles/ (FETCH_SIZE doubled per the gfx950 correction)
    traffic, pmc_note = None, "no PMC summary for this library build under profiles/"
    lib_sha = hashlib.sha256(open(_lib.library_path(), "rb").read()).hexdigest()[:16]
    pmc_path = os.path.join(ROOT, "profiles", "r03_pmc_summary.json")
    pmc = None
    if os.path.isfile(pmc_path):
        with open(pmc_path) as f:
            pmc = json.load(f)
        if pmc.get("library_sha256_16") != lib_sha:
            pmc_note = (f"profiles/r03_pmc_summary.json was made from library {pmc.get('library_sha256_16')}, this run "
                        f"uses {lib_sha}: not reported")
            pmc = None
        else:
            pmc_note = "HBM traffic per launch from profiles/r03_pmc_*.csv (rocprofv3 --pmc passes of this library build)"

    def pmc_bytes(role):
        r = (pmc or {}).get("roles", {}).get(role)
        return float(r["hbm_bytes"]) if r else None

    traffic = pmc_bytes("search_filter")
    roofline = dict(
        bound="valu", unit="Tops/s (integer subtract and |.|-accumulate operations)",
        kernel="k_assign_fast<DP2=8, R=2, packed keys, scalar-load codewords, PRUNE> (the filter launch of "
               "vq_encode_l1_prepared / vq_encode_l1_img on window-sorted vectors)",
        kernel_ms=prof["filter_kernel_ms"],
        achieved=prof["executed_tops"], peak=PEAK_SAD_TOPS, frac=prof["frac_of_sad_ceiling"], traffic=traffic,
        evaluated_pair_frac=prof["evaluated_pair_frac"],
        algorithmic_speedup=1.0 / prof["evaluated_pair_frac"],
        algorithmic_tops_all_pairs=prof["algorithmic_tops"],
        algorithmic_ops_per_vector=2 * K_CODEBOOK * d, algorithmic_bytes_per_vector=d + 4 + 2,
        queued_for_fp64_frac=prof["queued_for_fp64_frac"],
        fixup_kernel_ms=prof["fixup_kernel_ms"],
        full_search_kernel_ms=full_ms, full_search_executed_tops=full_ops / (full_ms * 1e-3) / 1e12,
        full_search_frac=full_ops / (full_ms * 1e-3) / 1e12 / PEAK_SAD_TOPS,
        full_search_labels_equal=full_equals_pruned, step_labels_equal_profiled=step_labels_equal,
        note="L1 distance has no GEMM form, so the bound is the vector ALU, not MFMA or HBM. achieved = EXECUTED "
             "subtract + |.|-accumulate ops (2*d per evaluated (vector, codeword) pair; pairs counted by the kernel "
             "itself: rows each wave evaluated x its 128 vectors) / the filter kernel's mean duration (HIP events "
             "recorded by the library right around that launch, on its stream). peak = the v_sad_u16 issue ceiling: one wave64 instruction per 4 cycles per "
             "SIMD (measured, profiles/r03_valu_ubench_forms1.txt and _forms2.txt, column ms; no operand form issues faster), two |x-c| accumulations "
             "per lane = 157.3 T ops/s. Besides its d/2 SADs every pair costs 3 more instructions of the same issue "
             "class per codeword pair (two keys, median, min3, min: 2.5 per codeword), so 8/10.5 = 0.76 is the ceiling of frac at d = 16; the kernel's time also "
             "holds the fp64 re-check of its near-ties (queued_for_fp64_frac of the vectors; there is no fix-up "
             "launch for this shape, fixup_kernel_ms = 0). The pruned search "
             "evaluates only evaluated_pair_frac of all N*K pairs (codewords whose element sum differs from the "
             "vector's by more than the best distance so far cannot win), with identical labels: that gain is "
             "algorithmic_speedup, not part of frac. " + pmc_note)
    sort_alg = N * (2 * d + 2)
    dec_alg = N * (d + 4)
    hbm_kernels = {
        "image_sort": dict(bound="hbm", kernel="k_window_sort_lds<4, image source> (vq_blocks_sort_img)", kernel_ms=sort_ms,
                           achieved=sort_alg / (sort_ms * 1e-3) / 1e9, peak=PEAK_HBM_GBS, unit="GB/s",
                           frac=sort_alg / (sort_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, traffic=pmc_bytes("image_sort"),
                           algorithmic_bytes_per_vector=2 * d + 2,
                           note="reads every pixel once, writes every row once at its sorted place plus a 2-byte origin"),
        "decode": dict(bound="hbm", kernel="k_decode_quads (vq_decode_u8)", kernel_ms=dec_ms,
                       achieved=dec_alg / (dec_ms * 1e-3) / 1e9, peak=PEAK_HBM_GBS, unit="GB/s",
                       frac=dec_alg / (dec_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, traffic=pmc_bytes("decode"),
                       algorithmic_bytes_per_vector=d + 4),
        "codebook_prepare": dict(kernel="k_prepare_small<4>", kernel_ms=prep_ms, note="K*d entries, one workgroup; latency-bound"),
    }

    # ---- LBG iterations (fused assign + accumulate, exchange, centroid update) ----------------------
    n_total = world * N
    cur = cb
    shard_info = shards_chunk_aligned(N, group, device) if distributed else None
    # (a stage costs about 0.14 ms besides its iterations -- state reset, the first codebook preparation, the read of
    # the last record --, so a short stage reads low: at least 100 iterations are timed whatever --steps says)
    lbg_steps = max(100, args.steps // 2)

    def lbg_run(cur, steps):
        if distributed:
            # several GPUs: the same stage with the exchange of every iteration enqueued between its two halves, on
            # device tensors (nothing returns to the host but the 16-byte records)
            aligned, sizes = shard_info
            return eng.lbg_stage_group(blocks, cur, n_total, -1.0, steps, group, sizes, aligned, sorted=srt)[0]
        # one GPU: what train_lbg runs -- a stage of `steps` iterations with the stop rule on the device
        # (threshold < 0: it never holds), iteration j + 1 enqueued before the outcome of iteration j is read
        return eng.lbg_stage(blocks, cur, n_total, -1.0, steps, sorted=srt)[0]

    cur = lbg_run(cur, 128)                                      # untimed: warm-up and the card's clock (see preroll)
    barrier()
    t0 = time.perf_counter()
    cur = lbg_run(cur, lbg_steps)
    barrier()
    lbg_elapsed = time.perf_counter() - t0
    t = torch.tensor([lbg_elapsed], dtype=torch.float64, device=device)
    if distributed:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    lbg_elapsed = float(t.item())
    # the dominant kernel of an iteration alone: the pruned search with the per-cluster accumulation and the winners'
    # fp64 distances fused in (HIP events recorded by the library around that launch)
    # (as a training stage launches it: no labels written, distances left in sorted order)
    lbg_run(cur, 128)
    acc_ms = eng.search_kernel_ms(lambda: eng.lbg_stage(None, cb, N, -1.0, 1, sorted=srt, shape=(N, d)), 20)
    _, acc_pairs = eng.search_statistics((cb, eng._workspace(K_CODEBOOK, d, N)), N)
    acc_ops = (2.0 * d) * float(acc_pairs) + float(d) * N        # SAD work of the evaluated pairs + d byte sums per vector
    lbg_roofline = dict(
        bound="valu", unit="Tops/s (integer subtract and |.|-accumulate operations)",
        kernel="k_assign_fast<DP2=8, R=2, packed keys, scalar-load codewords, PRUNE, ACC> (vq_lbg_iterate / vq_lbg_accum)",
        kernel_ms=acc_ms, achieved=acc_ops / (acc_ms * 1e-3) / 1e12, peak=PEAK_SAD_TOPS,
        frac=acc_ops / (acc_ms * 1e-3) / 1e12 / PEAK_SAD_TOPS, traffic=pmc_bytes("lbg_assign"),
        algorithmic_ops_per_vector=2 * K_CODEBOOK * d + d, algorithmic_bytes_per_vector=d,
        evaluated_pair_frac=float(acc_pairs) / (float(N) * K_CODEBOOK),
        note="executed ops = 2*d per evaluated (vector, codeword) pair + d accumulations per vector, against the v_sad_u16 "
             "issue ceiling; besides the search the kernel adds every vector's bytes into an LDS table (staged, "
             "transposed 64-bit LDS atomics) and evaluates the winner's fp64 distance from an LDS copy of the codebook")
    lbg = {"iters_per_s": lbg_steps / lbg_elapsed, "roofline": lbg_roofline, "vectors_per_s": n_total * lbg_steps / lbg_elapsed,
           "ms_per_iter": lbg_elapsed / lbg_steps * 1e3, "vectors_total": n_total, "timed_iterations": lbg_steps,
           "exchange": ("one collective per iteration on int64 [K*d sums | K counts] + fp64 per-chunk distance sums, "
                        "enqueued on device tensors between vq_lbg_iterate_local and vq_lbg_iterate_finish"
                        if distributed else "none (1 GPU)")}

    result = {
        "metric": "block-vectors encoded/sec (image -> labels -> decoded image; LBG iter/sec in `lbg`) at K=256,dim=16",
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
                               "encode+decode starting from the image", "vectors_per_gpu": N, "K": K_CODEBOOK, "dim": d,
                   "parallelism": f"dp{world} (block vectors sharded by image, codebook replicated)"},
        "step_issue": f"consecutive steps alternate between {max(1, args.lanes)} HIP streams (EncodePipeline); every step "
                      "runs all of its kernels; one stream: sequential_ms_per_step",
        "setup": f"before every timed region the same work runs untimed for {PREROLL_MS:.0f} ms (each pipeline lane allocates "
                 "and runs once; the card reaches its sustained clock after about 13 ms of continuous work, "
                 "tools/clock_ramp.py), then the W warm-up steps, then the K timed steps",
        "sequential_ms_per_step": sequential_ms,
        "sequential_vectors_per_s_per_gpu": N / (sequential_ms * 1e-3),
        "sequential_labels_equal": sequential_labels_equal,
        "encode_ms": enc_ms,
        "encode_vectors_per_s_per_gpu": N / (enc_ms * 1e-3),
        "decode_vectors_per_s_per_gpu": N / (dec_ms * 1e-3),
        "step_kernels_ms": {"image_sort": sort_ms, "codebook_prepare": prep_ms, "search_filter": prof["filter_kernel_ms"],
                            "search_fixup": prof["fixup_kernel_ms"], "decode": dec_ms},
        "lbg": lbg,
        "roofline": roofline,
        "hbm_kernels": hbm_kernels,
        "library_sha256_16": lib_sha,
    }

    if rank == 0 and world == 1 and not args.no_shapes:
        shapes = {}
        # worst case for pruning: uniform iid pixels (no spread of element sums to exploit)
        iid = iid_image(H, W, 1, device)
        cb_iid = train_lbg(eng, eng.blocks_from_image(iid[:1024, :1024].contiguous(), BH, BW), K_CODEBOOK, log=None).codebook
        shapes["cfg2_iid_K256_d16"] = search_profile(eng, iid, BH, BW, cb_iid, 20, device)[0]
        del iid
        # K = 1024, d = 16 (the codebook size of configs[2]) on the headline image
        cb1k = train_lbg(eng, crop, 1024, log=None).codebook
        shapes["K1024_d16"] = search_profile(eng, img, BH, BW, cb1k, 20, device)[0]
        # the reference's default colour path: RGB 4x4 blocks, d = 48
        rgb = structured_image(4096, 4096, 3, device, channels=3)
        cb48 = train_lbg(eng, eng.blocks_from_image(rgb[:1024, :1024].contiguous(), 4, 4), 256, log=None).codebook
        shapes["rgb4x4_K256_d48"] = search_profile(eng, rgb, 4, 4, cb48, 20, device)[0]
        del rgb
        # the same shape at four times the vectors: a launch of many rounds of workgroups instead of fewer than three
        # (what is left of a launch when the dispatcher runs out of workgroups weighs less)
        rgb = structured_image(8192, 8192, 3, device, channels=3)
        shapes["rgb4x4_K256_d48_4M_vectors"] = search_profile(eng, rgb, 4, 4, cb48, 10, device)[0]
        del rgb
        # configs[4]: RGB 8x8 blocks, d = 192, K = 4096 (one GPU's share: 4096 x 4096 x 3, N = 262 144)
        rgb8 = structured_image(4096, 4096, 5, device, channels=3)
        cb192 = train_lbg(eng, eng.blocks_from_image(rgb8[:1024, :1024].contiguous(), 8, 8), 4096, log=None).codebook
        shapes["cfg5_K4096_d192"] = search_profile(eng, rgb8, 8, 8, cb192, 5, device)[0]
        # four GPU shares in one call (N = 1 048 576): four workgroups per compute unit instead of one
        rgb8 = structured_image(8192, 8192, 5, device, channels=3)
        shapes["cfg5_K4096_d192_1M_vectors"] = search_profile(eng, rgb8, 8, 8, cb192, 3, device)[0]
        del rgb8
        result["shapes"] = shapes
        # configs[2]: full LBG training to K = 1024 on a 16384 x 16384 image (N = 16 777 216), one GPU
        big = structured_image(16384, 16384, 2, device)
        bblocks = eng.blocks_from_image(big, BH, BW)
        del big
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        res = train_lbg(eng, bblocks, 1024, log=None)
        torch.cuda.synchronize(device)
        t_train = time.perf_counter() - t0
        its = int(sum(res.iters_per_stage))
        result["lbg_full_training_cfg3"] = {
            "workload": "configs[2]: 16384x16384 grayscale, 4x4 blocks (N = 16 777 216, dim 16), LBG to K = 1024, "
                        "epsilon 0.01, threshold 0.001, max 100 iterations per stage",
            "seconds": t_train, "iterations": its, "iters_per_stage": list(res.iters_per_stage),
            "final_distortion": res.distortions[-1], "iterations_per_s": its / t_train,
            "vector_iterations_per_s": its * float(bblocks.shape[0]) / t_train}
        del bblocks

    if rank == 0 and world == 1 and args.cpu_sample > 0:
        n_s = min(args.cpu_sample, N)
        base, cpu_labels = run_cpu_baseline(blocks[:n_s].cpu().numpy(), cb.cpu().numpy())
        result["cpu_baseline"] = base
        result["cpu_labels_match_gpu"] = bool(np.array_equal(cpu_labels, labels_step[:n_s].cpu().numpy().astype(np.int64)))
        # the other half of the metric: one LBG iteration of the reference arithmetic on 2^20 of the vectors, with the
        # GPU's iteration on the same vectors beside it (codebooks compared bit for bit)
        n_l = min(1 << 20, n_s)
        lb, cpu_cb = run_cpu_baseline(blocks[:n_l].cpu().numpy(), cb.cpu().numpy(), leg="lbg")
        gpu_cb, gpu_D = lbg_iteration(eng, blocks[:n_l].contiguous(), cb, n_l)
        lb["codebook_equals_gpu"] = bool(np.array_equal(cpu_cb, gpu_cb.cpu().numpy()))
        lb["distortion_equals_gpu"] = bool(lb["distortion"] == gpu_D)
        lb["note"] = ("iterations/s at the full workload = value / vectors_per_gpu (the arithmetic is linear in the "
                      "number of vectors)")
        lb["iterations_per_s_at_workload"] = lb["value"] / N
        base["lbg"] = lb
        if not args.no_shapes:
            # encode-only figures for the codebook shapes of configs[2] and configs[4] on samples (one core)
            samples = {}
            for name, cbx, n_x in (("K1024_d16", cb1k, 1 << 18),):
                sb, sl = run_cpu_baseline(blocks[:n_x].cpu().numpy(), cbx.cpu().numpy(), leg="sample")
                sb["labels_equal_gpu"] = bool(np.array_equal(sl, eng.encode(blocks[:n_x], cbx).cpu().numpy().astype(np.int64)))
                samples[name] = sb
            b192 = eng.blocks_from_image(structured_image(1024, 1024, 5, device, channels=3), 8, 8)[:8192].contiguous()
            sb, sl = run_cpu_baseline(b192.cpu().numpy(), cb192.cpu().numpy(), leg="sample")
            sb["labels_equal_gpu"] = bool(np.array_equal(sl, eng.encode(b192, cb192).cpu().numpy().astype(np.int64)))
            samples["cfg5_K4096_d192"] = sb
            base["other_shapes"] = samples
    else:
        result["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(result))
    if distributed:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
