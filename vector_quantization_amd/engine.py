"""HipEngine: the device side of the VQ hot path, one object per GPU.

PyTorch-ROCm tensors are used only as device buffers (data_ptr) and for the current HIP stream;
every computation is a call through the C ABI of libvqhip.so.  uint64 sums and counts live in
int64 tensors (values stay far below 2^63).
"""
import ctypes

import torch

from . import _lib
from ._lib import lib, check, VQ_MODE_EXACT, VQ_MODE_FAST, VQ_PAD_EDGE, VQ_PAD_ZERO  # noqa: F401


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _nbytes(t):
    return t.numel() * t.element_size() if t is not None else 0


class HipEngine:
    """Launches the HIP kernels on one device.  Not thread-safe per instance (it caches its
    workspace); use one engine per host thread, or one per GPU in a multi-process job."""

    def __init__(self, device=None, mode="fast"):
        if not torch.cuda.is_available():
            raise RuntimeError(
                "vector_quantization_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                "there is no CPU fallback")
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.type != "cuda":
            raise ValueError(f"HipEngine needs a cuda/ROCm device, got {self.device}")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if mode not in ("fast", "exact"):
            raise ValueError("mode must be 'fast' or 'exact'")
        self.mode = VQ_MODE_FAST if mode == "fast" else VQ_MODE_EXACT
        self._ws = None
        self._scratch_n = -1
        self._labels_scratch = None
        self._dist_scratch = None
        self._side = None
        self._pinned = None

    # -- plumbing ----------------------------------------------------------------------------
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _workspace(self, K, d, N=0):
        need = lib.vq_workspace_bytes(int(K), int(d), int(N))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=self.device)
        return self._ws

    def _dev(self, t, dtype):
        if t.device != self.device or t.dtype != dtype or not t.is_contiguous():
            t = t.to(device=self.device, dtype=dtype).contiguous()
        return t

    def synchronize(self):
        torch.cuda.synchronize(self.device)

    # -- block extraction (VQ_Full.py:23-33, 51-57) ----------------------------------------------
    def blocks_from_image(self, img, bh, bw, pad_mode=VQ_PAD_EDGE):
        img = self._dev(img, torch.uint8)
        if img.dim() == 2:
            img = img.unsqueeze(-1)
        H, W, C = img.shape
        if bh <= 0 or bw <= 0:
            raise ValueError("Block height and width must be positive integers.")
        nr, nc = -(-H // bh), -(-W // bw)
        blocks = torch.empty((nr * nc, bh * bw * C), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.vq_blocks_u8(_ptr(img), H, W, C, bh, bw, pad_mode, _ptr(blocks), self._stream()),
                  "vq_blocks_u8")
        return blocks

    # -- initial centroid sums (VQ_Full.py:67) -----------------------------------------------------
    def colsum(self, blocks):
        N, d = blocks.shape
        sums = torch.empty(d, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.vq_colsum_u8(_ptr(blocks), N, d, _ptr(sums), self._stream()), "vq_colsum_u8")
        return sums

    # -- nearest codeword (VQ_Full.py:124-125) -----------------------------------------------------
    def sort_blocks(self, blocks, out=None):
        """Window-sorted copy of the block vectors for the pruned search (a second layout of
        self.blocks, VQ_Full.py:51-57); pass the result as `sorted=` to encode / encode_prepared /
        lbg_accum together with the same blocks.  Returns None for dimensions the fast kernels do
        not serve."""
        N, d = blocks.shape
        if d > 192 or N == 0:
            return None
        need = lib.vq_sorted_bytes(int(d), int(N))
        if out is None or out.numel() < need:
            out = torch.empty(need, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.vq_blocks_sort(_ptr(blocks), N, d, _ptr(out), out.numel(), self._stream()), "vq_blocks_sort")
        return out

    def sort_image(self, img, bh, bw, pad_mode=VQ_PAD_EDGE, out=None):
        """Image -> window-sorted block vectors in one pass (padding, block extraction and the sort of
        sort_blocks fused; VQ_Full.py:23-33, 51-57): the same buffer sort_blocks(blocks_from_image(img))
        fills, without building the block matrix.  Returns (buffer, (N, d)), or (None, (N, d)) for
        dimensions the fast kernels do not serve."""
        img = self._dev(img, torch.uint8)
        if img.dim() == 2:
            img = img.unsqueeze(-1)
        H, W, C = img.shape
        if bh <= 0 or bw <= 0:
            raise ValueError("Block height and width must be positive integers.")
        N, d = (-(-H // bh)) * (-(-W // bw)), bh * bw * C
        if d > 192 or N == 0:
            return None, (N, d)
        need = lib.vq_sorted_bytes(int(d), int(N))
        if out is None or out.numel() < need:
            out = torch.empty(need, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.vq_blocks_sort_img(_ptr(img), H, W, C, bh, bw, pad_mode, _ptr(out), out.numel(), self._stream()),
                  "vq_blocks_sort_img")
        return out, (N, d)

    def encode_image(self, img, bh, bw, cb, pad_mode=VQ_PAD_EDGE, want_dist=False, scratch=None, labels=None):
        """Labels of an image's block vectors straight from the image (VQ_Full.py:23-57 + :124-125)."""
        img = self._dev(img, torch.uint8)
        if img.dim() == 2:
            img = img.unsqueeze(-1)
        H, W, C = img.shape
        cb = self._dev(cb, torch.float64)
        K, d = cb.shape
        if d != bh * bw * C:
            raise ValueError(f"codebook rows have {d} entries, blocks have {bh * bw * C}")
        N = (-(-H // bh)) * (-(-W // bw))
        need = max(int(lib.vq_sorted_bytes(int(d), int(N))), N * d, 16)
        if scratch is None or scratch.numel() < need:
            scratch = torch.empty(need, dtype=torch.uint8, device=self.device)
        if labels is None:
            labels = torch.empty(N, dtype=torch.int32, device=self.device)
        mind = torch.empty(N, dtype=torch.float64, device=self.device) if want_dist else None
        ws = self._workspace(K, d, N)
        with torch.cuda.device(self.device):
            check(lib.vq_encode_l1_img(_ptr(img), H, W, C, bh, bw, pad_mode, _ptr(cb), K, _ptr(labels), _ptr(mind),
                                       _ptr(scratch), scratch.numel(), _ptr(ws), ws.numel(), self._stream()),
                  "vq_encode_l1_img")
        return (labels, mind) if want_dist else labels

    @staticmethod
    def prune_supported(K, d):
        return bool(lib.vq_prune_supported(int(K), int(d)))

    def prune_resolves_in_filter(self, K, d):
        """Whether the pruned search of this shape settles near-ties inside its filter kernel (no fix-up launch)."""
        return bool(lib.vq_prune_resolves_in_filter(int(K), int(d)))

    def encode(self, blocks, cb, want_dist=False, mode=None, sorted=None, shape=None):
        """blocks may be None when `sorted` serves the codebook's shape (prune_supported); `shape` = (N, d)
        then says what the sorted copy holds."""
        N, d = blocks.shape if blocks is not None else shape
        cb = self._dev(cb, torch.float64)
        K = cb.shape[0]
        labels = torch.empty(N, dtype=torch.int32, device=self.device)
        mind = torch.empty(N, dtype=torch.float64, device=self.device) if want_dist else None
        ws = self._workspace(K, d, N)
        use_mode = self.mode if mode is None else mode
        if sorted is not None and use_mode == VQ_MODE_FAST and N > 0 and self.prune_supported(K, d):
            with torch.cuda.device(self.device):
                check(lib.vq_codebook_prepare(_ptr(cb), K, d, _ptr(ws), ws.numel(), self._stream()),
                      "vq_codebook_prepare")
                check(lib.vq_encode_l1_prepared(_ptr(blocks), _ptr(sorted), _nbytes(sorted), N, d, _ptr(cb), K, _ptr(labels), _ptr(mind),
                                                _ptr(ws), ws.numel(), self._stream()), "vq_encode_l1_prepared")
            return (labels, mind) if want_dist else labels
        if blocks is None:
            raise ValueError("the block matrix is needed: the sorted copy does not serve this codebook shape or mode")
        with torch.cuda.device(self.device):
            check(lib.vq_encode_l1(_ptr(blocks), N, d, _ptr(cb), K, _ptr(labels), _ptr(mind),
                                   self.mode if mode is None else mode, _ptr(ws), ws.numel(), self._stream()),
                  "vq_encode_l1")
        return (labels, mind) if want_dist else labels

    def prepare_codebook(self, cb, max_vectors):
        """Fill a private workspace for cb; returns a handle for encode_prepared (many batches of up
        to max_vectors block vectors, one codebook)."""
        cb = self._dev(cb, torch.float64)
        K, d = cb.shape
        need = lib.vq_workspace_bytes(int(K), int(d), int(max_vectors))
        ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.vq_codebook_prepare(_ptr(cb), K, d, _ptr(ws), ws.numel(), self._stream()),
                  "vq_codebook_prepare")
        return (cb, ws)

    def encode_prepared(self, blocks, prepared, labels=None, want_dist=False, sorted=None, shape=None, phases=3):
        """phases: 1 = filter launch only, 2 = fix-up launch only (of the last filter pass), 3 = both."""
        cb, ws = prepared
        N, d = blocks.shape if blocks is not None else shape
        K = cb.shape[0]
        if labels is None:
            labels = torch.empty(N, dtype=torch.int32, device=self.device)
        mind = torch.empty(N, dtype=torch.float64, device=self.device) if want_dist else None
        with torch.cuda.device(self.device):
            check(lib.vq_encode_l1_prepared_phases(_ptr(blocks), _ptr(sorted), _nbytes(sorted), N, d, _ptr(cb), K,
                                                   _ptr(labels), _ptr(mind), int(phases), _ptr(ws), ws.numel(),
                                                   self._stream()), "vq_encode_l1_prepared_phases")
        return (labels, mind) if want_dist else labels

    def search_kernel_ms(self, fn, reps=20):
        """Mean duration in ms of the search kernel alone (HIP events recorded by the library right around that
        launch, on the launch stream) over `reps` calls of fn(), each of which must launch it exactly once."""
        stream = torch.cuda.current_stream(self.device)
        pairs = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)                                   # (creates the handles)
            b.record(stream)
            lib.vq_profile_next_search(ctypes.c_void_p(a.cuda_event), ctypes.c_void_p(b.cuda_event))
            fn()
            pairs.append((a, b))
        lib.vq_profile_next_search(None, None)
        torch.cuda.synchronize(self.device)
        return sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)

    def search_statistics(self, prepared, N):
        """(vectors re-checked in fp64 -- queued for the fix-up kernel, or settled inside the filter kernel by the
        variants that do that --, (vector, codeword) pairs evaluated) of the last filter pass that used this
        prepared workspace on N vectors."""
        cb, ws = prepared
        K, d = cb.shape
        off = lib.vq_workspace_counters_offset(int(K), int(d), int(N))
        slots = ws[off: off + 64 * 64].clone()
        queued = slots.view(torch.int32).reshape(64, 16)[:, :2].to(torch.int64).sum()
        pairs = slots.view(torch.int64).reshape(64, 8)[:, 1].sum()
        return int(queued.item()), int(pairs.item())

    # -- fused LBG assignment + accumulation (VQ_Full.py:77-90) -------------------------------------
    def lbg_accum(self, blocks, cb, want_labels=False, mode=None, want_chunks=False, sorted=None, shape=None):
        """Returns (acc, dist_sum[, labels][, chunk_sums]); acc is one int64 buffer
        [K*d sums | K counts] so that a multi-GPU job can all-reduce it in one collective.  dist_sum
        is the sum of the winning distances in numpy's summation order."""
        N, d = blocks.shape if blocks is not None else shape
        cb = self._dev(cb, torch.float64)
        K = cb.shape[0]
        acc = torch.zeros(K * d + K, dtype=torch.int64, device=self.device)
        dist = torch.empty(1, dtype=torch.float64, device=self.device)
        if self._scratch_n < N:
            self._labels_scratch = torch.empty(N, dtype=torch.int32, device=self.device)
            self._dist_scratch = torch.empty(N, dtype=torch.float64, device=self.device)
            self._scratch_n = N
        labels = torch.empty(N, dtype=torch.int32, device=self.device) if want_labels else self._labels_scratch
        chunks = torch.empty(-(-N // 8192) + 1, dtype=torch.float64, device=self.device) if want_chunks else None
        ws = self._workspace(K, d, N)
        sums, counts = acc[: K * d], acc[K * d:]
        with torch.cuda.device(self.device):
            check(lib.vq_lbg_accum(_ptr(blocks), _ptr(sorted), _nbytes(sorted), N, d, _ptr(cb), K, _ptr(sums), _ptr(counts), _ptr(labels),
                                   _ptr(self._dist_scratch), _ptr(dist), _ptr(chunks),
                                   self.mode if mode is None else mode, _ptr(ws), ws.numel(), self._stream()),
                  "vq_lbg_accum")
        out = [acc, dist]
        if want_labels:
            out.append(labels)
        if want_chunks:
            out.append(chunks[: -(-N // 8192)])
        return tuple(out)

    # -- one whole split stage with the stop rule on the device (VQ_Full.py:76-98) ----------------------
    def _mark(self):
        """An event on the engine's stream, recorded now."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return ev

    def _read_bytes(self, buf, off, nbytes, mark):
        """nbytes of a device uint8 buffer as numpy, as they are once the work before `mark` has finished;
        work enqueued after the mark keeps running (the copy goes through a side stream)."""
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
            self._pinned = torch.empty(256, dtype=torch.uint8).pin_memory()
        with torch.cuda.stream(self._side):
            self._side.wait_event(mark)
            self._pinned[:nbytes].copy_(buf[off: off + nbytes], non_blocking=True)
            self._side.synchronize()
        return self._pinned[:nbytes].numpy().copy()

    def lbg_stage(self, blocks, cb, n_total, threshold, limit, sorted=None, shape=None, mode=None):
        """All iterations of one split stage.  Iteration j + 1 is enqueued before the outcome of iteration j is
        read, so the GPU never waits for the host; the stop rule is evaluated on the device and gates the
        iterations enqueued past it.  threshold < 0: run exactly `limit` iterations.
        Returns (codebook after the stage, [distortion of every iteration run], index of the iteration that met
        the rule or -1)."""
        import numpy as np
        N, d = blocks.shape if blocks is not None else shape
        cb = self._dev(cb, torch.float64).clone()              # updated in place by every iteration
        K = cb.shape[0]
        if limit < 1:
            return cb, [], -1
        nstate = lib.vq_lbg_state_bytes(int(limit))
        state = torch.empty(nstate, dtype=torch.uint8, device=self.device)
        acc = torch.empty(K * d + K, dtype=torch.int64, device=self.device)
        if self._scratch_n < N:
            self._labels_scratch = torch.empty(N, dtype=torch.int32, device=self.device)
            self._dist_scratch = torch.empty(N, dtype=torch.float64, device=self.device)
            self._scratch_n = N
        ws = self._workspace(K, d, N)
        use_mode = self.mode if mode is None else mode
        # the fused pass writes labels only when given a buffer for them: training does not need any
        lab = None if (sorted is not None and use_mode == VQ_MODE_FAST and lib.vq_lbg_pass_fused(int(K), int(d))) \
            else self._labels_scratch
        with torch.cuda.device(self.device):
            check(lib.vq_lbg_stage_begin(_ptr(state), nstate, int(limit), self._stream()), "vq_lbg_stage_begin")

        def enqueue(j):
            with torch.cuda.device(self.device):
                check(lib.vq_lbg_iterate(_ptr(blocks), _ptr(sorted), _nbytes(sorted), N, d, _ptr(cb), K, int(n_total),
                                         float(threshold), j, _ptr(acc[: K * d]), _ptr(acc[K * d:]),
                                         _ptr(lab), _ptr(self._dist_scratch), use_mode, _ptr(state),
                                         nstate, _ptr(ws), ws.numel(), self._stream()), "vq_lbg_iterate")
            return self._mark()

        distortions, conv = self._drive_stage(enqueue, state, limit)
        return cb, distortions, conv

    def _drive_stage(self, enqueue, state, limit):
        """Enqueues iteration j + 1 before reading the record of iteration j (through the side stream), until the
        record says the stop rule held or the iteration was gated off.  Returns (distortions, converged index or -1)."""
        import numpy as np
        distortions, conv = [], -1
        mark = enqueue(0)
        for j in range(limit):
            nxt = enqueue(j + 1) if j + 1 < limit else None
            rec = self._read_bytes(state, 64 + 16 * j, 16, mark)
            mark = nxt
            D = float(np.frombuffer(rec[:8].tobytes(), dtype=np.float64)[0])
            converged, valid = (int(v) for v in np.frombuffer(rec[8:16].tobytes(), dtype=np.uint32))
            if not valid:
                break
            distortions.append(D)
            if converged:
                conv = j
                break
        return distortions, conv

    def lbg_stage_group(self, blocks, cb, n_total, threshold, limit, group, sizes, aligned, sorted=None, shape=None,
                        mode=None, allreduce_from_bytes=1 << 20):
        """lbg_stage for block vectors sharded over the ranks of `group` (RCCL): per iteration this rank's
        vq_lbg_iterate_local, one exchange on device tensors (all-gather of [sums | counts | chunk sums] while that is
        small, else an integer all-reduce plus an all-gather of the chunk sums), vq_lbg_iterate_finish -- all enqueued
        ahead of the host, nothing copied to the host but the 16-byte record of each iteration (side stream).  Every
        rank computes the same records, so every rank stops at the same iteration and issues the same collectives.
        sizes: vectors per rank; aligned: every rank but the last holding data has a multiple of 8192."""
        import torch.distributed as dist
        N, d = blocks.shape if blocks is not None else shape
        cb = self._dev(cb, torch.float64).clone()
        K = cb.shape[0]
        if limit < 1:
            return cb, [], -1
        world = dist.get_world_size(group=group)
        nstate = lib.vq_lbg_state_bytes(int(limit))
        state = torch.empty(nstate, dtype=torch.uint8, device=self.device)
        nacc = K * d + K
        nch = -(-N // 8192)
        width = max(1, max(-(-n // 8192) for n in sizes)) if aligned else 1
        counts_h = (ctypes.c_int * world)(*[(-(-n // 8192) if aligned else 1) for n in sizes])
        # This rank's record [sums | counts | chunk sums (or the local total)]: the local half writes it in place, an
        # all-gather moves it as it is, the finishing half reads every rank's record where the all-gather left it.
        row = nacc + width
        record = torch.zeros(row, dtype=torch.int64, device=self.device)
        tail = record[nacc:].view(torch.float64)
        unaligned_chunks = None if aligned else torch.zeros(nch + 1, dtype=torch.float64, device=self.device)
        small = nacc * 8 * world <= allreduce_from_bytes
        if small:
            gathered = torch.empty((world, row), dtype=torch.int64, device=self.device)
        else:
            chunks_all = torch.empty((world, width), dtype=torch.float64, device=self.device)
        if self._scratch_n < N:
            self._labels_scratch = torch.empty(N, dtype=torch.int32, device=self.device)
            self._dist_scratch = torch.empty(N, dtype=torch.float64, device=self.device)
            self._scratch_n = N
        ws = self._workspace(K, d, N)
        use_mode = self.mode if mode is None else mode
        lab = None if (sorted is not None and use_mode == VQ_MODE_FAST and lib.vq_lbg_pass_fused(int(K), int(d))) \
            else self._labels_scratch
        with torch.cuda.device(self.device):
            check(lib.vq_lbg_stage_begin(_ptr(state), nstate, int(limit), self._stream()), "vq_lbg_stage_begin")

        def enqueue(j):
            with torch.cuda.device(self.device):
                check(lib.vq_lbg_iterate_local(_ptr(blocks), _ptr(sorted), _nbytes(sorted), N, d, _ptr(cb), K, j,
                                               _ptr(record[: K * d]), _ptr(record[K * d: nacc]),
                                               _ptr(tail) if aligned else _ptr(unaligned_chunks),
                                               None if aligned else _ptr(tail), _ptr(lab),
                                               _ptr(self._dist_scratch), use_mode, _ptr(state), nstate, _ptr(ws), ws.numel(),
                                               self._stream()), "vq_lbg_iterate_local")
                if small:
                    dist.all_gather_into_tensor(gathered.view(-1), record, group=group)
                    check(lib.vq_lbg_iterate_finish(_ptr(gathered[0, : K * d]), _ptr(gathered[0, K * d: nacc]), world, row,
                                                    _ptr(gathered[0, nacc:]), world, row, counts_h, _ptr(cb), K, d,
                                                    int(n_total), float(threshold), j, _ptr(state), nstate, _ptr(ws),
                                                    ws.numel(), self._stream()), "vq_lbg_iterate_finish")
                else:
                    dist.all_gather_into_tensor(chunks_all.view(-1), tail, group=group)
                    dist.all_reduce(record[:nacc], op=dist.ReduceOp.SUM, group=group)   # integers: any order, same sums
                    check(lib.vq_lbg_iterate_finish(_ptr(record[: K * d]), _ptr(record[K * d: nacc]), 1, 0, _ptr(chunks_all),
                                                    world, width, counts_h, _ptr(cb), K, d, int(n_total), float(threshold),
                                                    j, _ptr(state), nstate, _ptr(ws), ws.numel(), self._stream()),
                          "vq_lbg_iterate_finish")
            return self._mark()

        distortions, conv = self._drive_stage(enqueue, state, limit)
        return cb, distortions, conv

    def numpy_sum(self, v):
        """Sum of a float64 vector in numpy's summation order (np.add.reduce), on the device."""
        v = self._dev(v, torch.float64).reshape(-1)
        N = v.numel()
        out = torch.empty(1, dtype=torch.float64, device=self.device)
        chunks = torch.empty(-(-N // 8192) + 1, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.vq_numpy_sum_f64(_ptr(v), N, _ptr(out), _ptr(chunks), self._stream()), "vq_numpy_sum_f64")
        return out

    def accum_labels(self, blocks, labels, K):
        N, d = blocks.shape
        acc = torch.zeros(K * d + K, dtype=torch.int64, device=self.device)
        labels = self._dev(labels, torch.int32)
        with torch.cuda.device(self.device):
            check(lib.vq_accum_labels(_ptr(blocks), N, d, _ptr(labels), K, _ptr(acc[: K * d]), _ptr(acc[K * d:]),
                                      self._stream()), "vq_accum_labels")
        return acc

    # -- centroid update / split (VQ_Full.py:81-86, 71-73) -------------------------------------------
    def lbg_finalize(self, acc, cb_old):
        cb_old = self._dev(cb_old, torch.float64)
        K, d = cb_old.shape
        cb_new = torch.empty_like(cb_old)
        with torch.cuda.device(self.device):
            check(lib.vq_lbg_finalize(_ptr(acc[: K * d]), _ptr(acc[K * d:]), _ptr(cb_old), K, d, _ptr(cb_new),
                                      self._stream()), "vq_lbg_finalize")
        return cb_new

    def lbg_split(self, cb, eps):
        cb = self._dev(cb, torch.float64)
        K, d = cb.shape
        cb2 = torch.empty((2 * K, d), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.vq_lbg_split(_ptr(cb), K, d, float(eps), _ptr(cb2), self._stream()), "vq_lbg_split")
        return cb2

    # -- decode (VQ_Full.py:157-169) -------------------------------------------------------------------
    def decode(self, labels, cb, bh, bw, C, out=None):
        labels = self._dev(labels, torch.int32)
        nr, nc = labels.shape
        cb = self._dev(cb, torch.float64).reshape(-1, bh * bw * C)
        K = cb.shape[0]
        shape = (nr * bh, nc * bw, C)
        if out is not None and (tuple(out.shape) != shape or out.dtype != torch.uint8 or out.device != self.device or
                                not out.is_contiguous()):
            raise ValueError(f"out must be a contiguous uint8 tensor of shape {shape} on {self.device}")
        img = out if out is not None else torch.empty(shape, dtype=torch.uint8, device=self.device)
        ws = self._workspace(K, bh * bw * C)
        with torch.cuda.device(self.device):
            check(lib.vq_decode_u8(_ptr(labels), nr, nc, _ptr(cb), K, bh, bw, C, _ptr(img), _ptr(ws), ws.numel(),
                                   self._stream()), "vq_decode_u8")
        return img

    # -- label bit stream (VQ_Full.py:134-147) -----------------------------------------------------------
    def pack_labels(self, labels, K):
        labels = self._dev(labels, torch.int32).reshape(-1)
        N = labels.numel()
        nbytes = lib.vq_packed_bytes(N, int(K))
        out = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.vq_pack_labels(_ptr(labels), N, int(K), _ptr(out), self._stream()), "vq_pack_labels")
        return out

    def unpack_labels(self, packed, N, K):
        packed = self._dev(packed, torch.uint8)
        need = lib.vq_packed_bytes(int(N), int(K))
        if packed.numel() < need:
            raise ValueError(f"packed stream has {packed.numel()} bytes, {need} needed for N={N}, K={K}")
        labels = torch.empty(int(N), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib.vq_unpack_labels(_ptr(packed), int(N), int(K), _ptr(labels), self._stream()),
                  "vq_unpack_labels")
        return labels
