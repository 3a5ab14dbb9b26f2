"""LBG codebook training loop (host control flow of VQ_Full.py:60-98) over an engine.

The engine does the per-vector work on its GPU; this module owns the split schedule, the
stop rule and -- when the block vectors are sharded over several GPUs -- the per-iteration
exchange: one all-gather of every rank's integer [sums | counts] buffer together with its fp64
per-chunk distance sums (large codebooks: an integer all-reduce plus a gather of the distance sums);
every rank then adds the distance partials in rank order.  Integer sums make the n-GPU
codebook identical to the 1-GPU one; with 8192-aligned shards the distortion (numpy's summation
order) is identical too, so the stop rule fires at the same iteration.
"""
import math

import torch


class LBGResult:
    def __init__(self, codebook, iters_per_stage, converged_iter, distortions):
        self.codebook = codebook                  # (K, d) float64 tensor on the engine's device
        self.iters_per_stage = iters_per_stage    # iterations run in each split stage
        self.converged_iter = converged_iter      # the `i` the reference prints, or -1
        self.distortions = distortions            # every iteration's distortion, in order


def _group_size(group):
    import torch.distributed as dist
    return dist.get_world_size(group=group)


def _needs_host_staging(t, group):
    """gloo moves host memory; device tensors are staged through the host for it (RCCL takes them
    as they are)."""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _allreduce_sum_(t, group):
    import torch.distributed as dist
    if _needs_host_staging(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


NP_CHUNK = 8192   # elements per pairwise-summed chunk of numpy's reduction (and of vq_numpy_sum_f64)


# Above this many bytes of [sums | counts] per rank, times the number of ranks, the integer part travels by
# all-reduce (each rank moves about twice its own payload, whatever the rank count) and only the few fp64
# chunk sums are gathered; below it one all-gather of everything is a single latency-bound message.
ALLREDUCE_FROM_BYTES = 1 << 20


def _all_gather_into(src, world, group):
    """all_gather of equally sized 1-D tensors; returns a (world, n) tensor on src's device."""
    import torch.distributed as dist
    if dist.get_backend(group) == "gloo":                   # gloo moves host memory, one tensor per rank
        h = src.cpu()
        parts = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(parts, h, group=group)
        return torch.stack(parts).to(src.device)
    out = torch.empty((world, src.numel()), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out.view(-1), src.contiguous(), group=group)
    return out


def shards_chunk_aligned(n_local, group, device=None):
    """True when the ranks' shards, concatenated in rank order, keep numpy's 8192-element chunk
    boundaries inside shards: every rank but the last one holding data has a multiple of 8192.
    Returns (aligned, sizes); one small tensor all-gather."""
    import torch.distributed as dist
    world = dist.get_world_size(group=group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    sizes = [int(v) for v in _all_gather_into(mine, world, group).reshape(-1).tolist()]
    holders = [r for r in range(world) if sizes[r] > 0]
    return all(sizes[r] % NP_CHUNK == 0 for r in holders[:-1]), sizes


def _exchange(acc, dist_sum, chunks, sizes, aligned, group):
    """The exchange of an LBG iteration.  Every rank contributes its integer [sums | counts] and its fp64
    distance partials.  Small payloads go out as ONE all-gather of one int64 buffer (the fp64 partials
    bit-cast); large ones (K * (d + 1) * 8 * world beyond ALLREDUCE_FROM_BYTES) as an integer all-reduce plus
    an all-gather of the distance partials alone.  Returns (integer [sums | counts] added over the ranks, on
    acc's device; the global sum of winning distances as a Python float).

    With chunk-aligned shards the distance total is the left-to-right sum of every rank's per-chunk
    sums in rank order -- bit for bit what one process computes on the concatenated data; otherwise
    the per-rank sums added in rank order.  Integer sums do not depend on the order of addition, so both
    routes give identical codebooks."""
    import numpy as np
    import torch.distributed as dist
    world = dist.get_world_size(group=group)
    nacc = acc.numel()
    width = max(1, max(-(-n // NP_CHUNK) for n in sizes)) if aligned else 1
    tail = torch.zeros(width, dtype=torch.float64, device=acc.device)
    if aligned:
        tail[: chunks.numel()] = chunks
    else:
        tail[0] = dist_sum.reshape(-1)[0]
    if nacc * 8 * world > ALLREDUCE_FROM_BYTES:
        acc_total = _allreduce_sum_(acc.clone(), group)
        host = _all_gather_into(tail, world, group).cpu().numpy()
    else:
        payload = torch.cat([acc, tail.view(torch.int64)])
        gathered = _all_gather_into(payload, world, group)             # (world, nacc + width)
        acc_total = gathered[:, :nacc].sum(dim=0)                      # integers: any order gives the same sums
        host = gathered[:, nacc:].contiguous().view(torch.float64).cpu().numpy()
    if aligned:
        seq = np.concatenate([host[r, : -(-sizes[r] // NP_CHUNK)] for r in range(world)])
        total = float(np.cumsum(seq)[-1]) if seq.size else 0.0     # cumsum adds strictly left to right
    else:
        total = 0.0
        for r in range(world):
            total += float(host[r, 0])
    return acc_total, total


def initial_centroid(engine, blocks, group=None):
    """np.mean(self.blocks, axis=0) of VQ_Full.py:67 as exact integer column sums / N."""
    sums = engine.colsum(blocks)
    n = torch.tensor([blocks.shape[0]], dtype=torch.int64, device=sums.device)
    packed = torch.cat([sums, n])            # laid out as a one-cluster [sums | count] buffer
    if group is not None:
        _allreduce_sum_(packed, group)
    n_total = int(packed[-1].item())
    if n_total == 0:
        raise ValueError("no block vectors")
    # one IEEE divide of an exactly representable integer by the count, as numpy does
    d = sums.numel()
    centroid = engine.lbg_finalize(packed, torch.zeros((1, d), dtype=torch.float64, device=sums.device))
    return centroid, n_total


def lbg_iteration(engine, blocks, cb, n_total, group=None, shard_info=None, sorted=None):
    """One pass of VQ_Full.py:77-90.  Returns (new codebook, distortion as a Python float).
    sorted: engine.sort_blocks(blocks), built once for all iterations (optional)."""
    if group is None:
        acc, dist = engine.lbg_accum(blocks, cb, sorted=sorted)
        new_cb = engine.lbg_finalize(acc, cb)      # enqueued before the host waits for the distance sum
        total = float(dist.item())
    else:
        if shard_info is None:
            shard_info = shards_chunk_aligned(blocks.shape[0], group, blocks.device)
        aligned, sizes = shard_info
        acc, dist, chunks = engine.lbg_accum(blocks, cb, want_chunks=True, sorted=sorted)
        acc, total = _exchange(acc, dist, chunks, sizes, aligned, group)
        new_cb = engine.lbg_finalize(acc, cb)
    return new_cb, total / float(n_total)


def train_lbg(engine, blocks, k, epsilon=0.01, threshold=0.001, max_iterations=100, group=None, log=print,
              iters_per_stage=None, on_converged=None, sorted=None, shape=None):
    """Follows VQ_Full.py:60-98.

    blocks: this rank's (N_local, d) uint8 shard on the engine's device.  group: a
    torch.distributed process group when the vectors are sharded (None = single GPU).
    iters_per_stage: optional list forcing the number of iterations of each stage (parity runs
    "given the same iteration count"); the stop rule is then not consulted.
    on_converged(size, i, distortion): called instead of `log` when a stage meets the stop rule.
    """
    centroid, n_total = initial_centroid(engine, blocks, group)
    if k > n_total:
        raise ValueError(
            f"Invalid quantization level k={k}: cannot exceed the total number of image blocks ({n_total})."
        )
    cb = centroid
    if sorted is None:
        sorted = engine.sort_blocks(blocks)       # the vectors do not change: one sort serves every iteration
    shard_info = shards_chunk_aligned(blocks.shape[0], group, blocks.device) if group is not None else None
    iters, conv, distortions = [], [], []
    stage = 0
    while cb.shape[0] < k:
        cb = engine.lbg_split(cb, epsilon)
        prev = float("inf")
        done, conv_i = 0, -1
        forced = None if iters_per_stage is None else int(iters_per_stage[stage])
        limit = max_iterations if forced is None else forced
        device_stage = (group is None and hasattr(engine, "lbg_stage")) or \
                       (group is not None and hasattr(engine, "lbg_stage_group") and not _needs_host_staging(cb, group))
        if device_stage:
            # the stage runs with the stop rule on the device, iterations enqueued ahead of the host (several GPUs: the
            # exchange of each iteration is enqueued between its two halves, on device tensors)
            n_d = None if blocks is not None else shape
            thr = -1.0 if forced is not None else threshold
            if group is None:
                cb, stage_d, conv_i = engine.lbg_stage(blocks, cb, n_total, thr, limit, sorted=sorted, shape=n_d)
            else:
                aligned, sizes = shard_info
                cb, stage_d, conv_i = engine.lbg_stage_group(blocks, cb, n_total, thr, limit, group, sizes, aligned,
                                                             sorted=sorted, shape=n_d,
                                                             allreduce_from_bytes=ALLREDUCE_FROM_BYTES)
            distortions.extend(stage_d)
            done = len(stage_d)
            if conv_i >= 0:
                if on_converged is not None:
                    on_converged(int(cb.shape[0]), conv_i, stage_d[-1])
                elif log is not None:
                    log(f"Converged at iter {conv_i}, distortion={stage_d[-1]:.3f}")
        else:
            for i in range(limit):
                cb, distortion = lbg_iteration(engine, blocks, cb, n_total, group, shard_info, sorted)
                distortions.append(distortion)
                done += 1
                if forced is None and prev != float("inf"):
                    # prev == 0 gives nan (or inf), which never satisfies the test -- as in the reference
                    change = abs(prev - distortion) / prev if prev != 0.0 else float("nan")
                    if change < threshold:
                        conv_i = i
                        if on_converged is not None:
                            on_converged(int(cb.shape[0]), i, distortion)
                        elif log is not None:
                            log(f"Converged at iter {i}, distortion={distortion:.3f}")
                        break
                prev = distortion
        iters.append(done)
        conv.append(conv_i)
        stage += 1
    return LBGResult(cb, iters, conv, distortions)


def final_codebook_size(k):
    """Codebook size the doubling loop of VQ_Full.py:70 ends with: the next power of two >= k."""
    return 1 if k <= 1 else 1 << math.ceil(math.log2(k))
