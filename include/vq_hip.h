/*
 * vq_hip.h -- C ABI of libvqhip.so, the MI355X (gfx950) implementation of the image
 * vector-quantisation hot path of VQ_Full.py: block extraction, L1 nearest-codeword
 * search, LBG split / centroid update, decode.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer owned by the caller unless its name starts with h_.
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it and
 *     returns without synchronising.  No entry point allocates or frees device memory: all
 *     scratch comes in through the `ws` argument (size from vq_workspace_bytes()).
 *   - Return value: 0 on success, a negative VQ_E_* code on failure; vq_last_error()
 *     returns a thread-local message for the last failure on the calling thread.
 *   - The library may be called from any host thread.  The only state it keeps between calls are
 *     three mutex-guarded tables -- device properties (queried once per device), the buffers
 *     vq_blocks_sort has filled (address -> N, d; at most 4096 records, see vq_sorted_forget), the
 *     per-device address of the static scratch -- and, per host thread, a note of what the last
 *     vq_lbg_iterate / vq_lbg_iterate_local left in a workspace (see there).
 *   - Block vectors are rows of an (N, d) uint8 matrix, d = block_h * block_w * C, element
 *     order (row-in-block, col-in-block, channel), vectors in row-major block-grid order.
 *     Codebooks are (K, d) IEEE doubles, row-major.
 *   - `mode`: VQ_MODE_FAST and VQ_MODE_EXACT return identical results (same labels, same
 *     fp64 distances).  FAST filters with a 16-bit fixed-point SAD and re-checks every
 *     near-tie in sequential fp64; EXACT evaluates every distance in sequential fp64.
 */
#ifndef VQ_HIP_H
#define VQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQ_OK             0
#define VQ_E_INVALID     -1   /* bad argument (null pointer, non-positive size, k > N ...) */
#define VQ_E_HIP         -2   /* a HIP runtime call failed; message holds hipGetErrorString */
#define VQ_E_WORKSPACE   -3   /* ws is null or smaller than vq_workspace_bytes(K, d) */
#define VQ_E_UNSUPPORTED -4   /* shape outside what the kernels handle */

#define VQ_MODE_EXACT 0
#define VQ_MODE_FAST  1

#define VQ_PAD_EDGE 0   /* replicate last row / column (VQ_Full.py:28-33) */
#define VQ_PAD_ZERO 1   /* constant 0 (old/CodeBook_VQ.py:26) */

/* Bytes of the static device array behind vq_encode_l1_static (one copy per GPU). */
#define VQ_STATIC_SCRATCH_BYTES (8u << 20)

/* ABI version of this header; bumped on any signature change. */
int vq_abi_version(void);

/* Thread-local message of the last failing call on this thread ("" if none). */
const char *vq_last_error(void);

/* Bytes of device scratch for a (K, d) codebook and calls on up to N block vectors (N sizes the
 * queue of vectors that need the fp64 re-check and the per-chunk distance sums; at most 2^32 - 16
 * vectors per call).  vq_codebook_prepare and vq_decode_u8 only need N = 0.  Pure host arithmetic. */
size_t vq_workspace_bytes(int K, int d, int64_t N);

/* Block extraction.  Replaces Codebook.__init__'s padding (VQ_Full.py:23-33) and
 * Codebook.image_to_blocks (VQ_Full.py:51-57): img is (H, W, C) uint8, padded on the bottom /
 * right to multiples of (bh, bw) with pad_mode, and cut into
 * ceil(H/bh) * ceil(W/bw) vectors of d = bh*bw*C bytes written to blocks. */
int vq_blocks_u8(const uint8_t *img, int H, int W, int C, int bh, int bw, int pad_mode,
                 uint8_t *blocks, void *stream);

/* Column sums of the block matrix: sums[e] = sum_n blocks[n, e] (exact, uint64).  The initial
 * centroid np.mean(self.blocks, axis=0) of VQ_Full.py:67 is sums[e] / N.  sums is overwritten. */
int vq_colsum_u8(const uint8_t *blocks, int64_t N, int d, uint64_t *sums, void *stream);

/* Nearest codeword.  Replaces cdist(..., 'cityblock') + np.argmin of Codebook.compress
 * (VQ_Full.py:124-125): labels[n] = argmin_k sum_e |double(blocks[n,e]) - cb[k,e]| with the sum
 * taken left to right in fp64 and the lowest k winning ties.  labels holds N int32.  min_dist
 * (nullable, N doubles) receives the winning fp64 distance per vector (the values VQ_Full.py:89
 * gathers). */
int vq_encode_l1(const uint8_t *blocks, int64_t N, int d, const double *cb, int K,
                 int32_t *labels, double *min_dist, int mode, void *ws, size_t ws_bytes, void *stream);

/* vq_encode_l1 for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments
 * and results as vq_encode_l1 (VQ_Full.py:124-125), but the scratch is a static device array of
 * VQ_STATIC_SCRATCH_BYTES (8 MiB) inside the library instead of a caller-owned workspace, and the
 * call allocates nothing.  A (K, d, N) whose vq_workspace_bytes exceeds that size is refused with
 * VQ_E_UNSUPPORTED (at K = 256, d = 16 about 1.7 million vectors per call fit).  Every call on a GPU
 * shares that GPU's array: calls that overlap in time on one GPU (different streams or threads)
 * race on it and must be serialised by the caller.  The first call on a device looks the array's address up
 * (hipGetSymbolAddress) and a first call per kernel shape may set a function attribute: make one warm-up
 * call per device and shape outside any stream capture (vq_static_scratch_bytes() does the address
 * look-up for the current device as well). */
size_t vq_static_scratch_bytes(void);
int vq_encode_l1_static(const uint8_t *blocks, int64_t N, int d, const double *cb, int K,
                        int32_t *labels, double *min_dist, int mode, void *stream);

/* Window-sorted copy of the block vectors, the input of the pruned search.  No counterpart in the
 * reference: it is a second layout of self.blocks (VQ_Full.py:51-57), built once per image and
 * reused by every LBG iteration (VQ_Full.py:77) and by compress (VQ_Full.py:124).  Inside every window of 8192
 * consecutive vectors the rows are reordered by element sum; `sorted` (vq_sorted_bytes(d, N) bytes,
 * 16-byte aligned) receives the reordered rows and, per row, where in its window it came from.
 * Passing it to vq_encode_l1_prepared / vq_lbg_accum lets the search skip every codeword whose
 * element sum differs from the vector's by more than the best distance found (a lower bound of
 * the L1 distance), with identical results.  vq_prune_supported(K, d) tells whether a (K, d)
 * codebook is served by that search (else `sorted` is ignored): 32 <= K <= 4096 and d <= 192.
 * vq_blocks_sort ends the buffer with a (magic, N, d) header and records on the host which buffer it
 * filled for which (N, d).  A search that is handed a `sorted` argument it has no such record of, one
 * recorded for another N or d, or one smaller than vq_sorted_bytes(d, N), returns VQ_E_INVALID; a
 * buffer whose header was overwritten since the sort makes the search write label -1 for every
 * vector instead of following a stale origin map.  The record is keyed by the buffer's address: a copy of a
 * sorted buffer is not accepted (sort into the copy), and before a sorted buffer's memory is freed or reused
 * for something else the caller should drop its record with vq_sorted_forget (returns 1 if there was one);
 * the table holds at most 4096 records and starts over when it is full (a search then asks for a fresh sort). */
size_t vq_sorted_bytes(int d, int64_t N);
int vq_sorted_forget(const void *sorted);
int vq_prune_supported(int K, int d);
int vq_blocks_sort(const uint8_t *blocks, int64_t N, int d, void *sorted, size_t sorted_bytes, void *stream);

/* Image-fused forms.  vq_blocks_sort_img does the work of vq_blocks_u8 followed by vq_blocks_sort in one
 * pass -- padding (VQ_Full.py:23-33), block extraction (VQ_Full.py:51-57) and the window sort -- reading
 * the image once and never writing the (N, d) block matrix; `sorted` (vq_sorted_bytes(d, N) bytes with
 * N = ceil(H/bh) * ceil(W/bw), d = bh*bw*C <= 192) comes out exactly as those two calls would fill it.
 * vq_encode_l1_img is Codebook.__init__'s block extraction plus Codebook.compress's search
 * (VQ_Full.py:23-57, :124-125) on an image: labels (and min_dist, nullable) as vq_encode_l1 gives them for
 * the image's block vectors.  `scratch` (at least vq_sorted_bytes(d, N) bytes, 16-byte aligned) receives
 * the window-sorted rows when the pruned search serves (K, d), else the plain block matrix. */
int vq_blocks_sort_img(const uint8_t *img, int H, int W, int C, int bh, int bw, int pad_mode, void *sorted,
                       size_t sorted_bytes, void *stream);
int vq_encode_l1_img(const uint8_t *img, int H, int W, int C, int bh, int bw, int pad_mode, const double *cb, int K,
                     int32_t *labels, double *min_dist, void *scratch, size_t scratch_bytes, void *ws,
                     size_t ws_bytes, void *stream);

/* The same search split in two, for callers that encode many batches with one codebook (a batch of
 * images, repeated benchmark steps): vq_codebook_prepare fills ws from cb once (one small launch),
 * vq_encode_l1_prepared then runs VQ_MODE_FAST's two kernels (filter, fp64 fix-up) per batch.  ws must not
 * be handed to any other call in between; cb must still hold the same codebook.  Only dimensions
 * d <= 192 (the SAD filter kernels) are served; others return VQ_E_UNSUPPORTED.  `sorted` is null
 * or the buffer vq_blocks_sort / vq_blocks_sort_img filled from the same blocks (sorted_bytes: its
 * size).  When `sorted` is given and vq_prune_supported(K, d), `blocks` is not read and may be null. */
int vq_codebook_prepare(const double *cb, int K, int d, void *ws, size_t ws_bytes, void *stream);
int vq_encode_l1_prepared(const uint8_t *blocks, const void *sorted, size_t sorted_bytes, int64_t N, int d,
                          const double *cb, int K, int32_t *labels, double *min_dist, void *ws, size_t ws_bytes,
                          void *stream);

/* For profiling and benchmarks: vq_encode_l1_prepared with its two launches selectable, phases = 1 the
 * filter (queue reset + SAD search; labels / min_dist of the vectors it settles), 2 the fp64 fix-up of the
 * vectors the last filter pass queued in this ws, 3 both (= vq_encode_l1_prepared).  After a filter pass
 * the workspace holds two statistics per sub-queue slot q (q = 0 .. 63) at byte offset
 * vq_workspace_counters_offset(K, d, N) + 64 * q: at byte 0 a uint32 count of vectors queued for the
 * fix-up, at byte 4 a uint32 count of near-ties the filter kernel settled in fp64 itself (the pruned search
 * with scalar-loaded codewords does that and queues nothing: its phase 2 is empty), at byte 8 a uint64 count
 * of the (vector, codeword) pairs whose distance the filter evaluated (N * K rounded up to the kernel's batch
 * without pruning). */
int vq_encode_l1_prepared_phases(const uint8_t *blocks, const void *sorted, size_t sorted_bytes, int64_t N, int d,
                                 const double *cb, int K, int32_t *labels, double *min_dist, int phases, void *ws,
                                 size_t ws_bytes, void *stream);
size_t vq_workspace_counters_offset(int K, int d, int64_t N);
/* For profiling and benchmarks: the next launch of the search kernel (the filter of VQ_MODE_FAST, with or without the
 * fused LBG accumulation) made by the calling thread is bracketed by hipEventRecord(ev_start) / hipEventRecord(ev_stop)
 * on the call's stream -- that kernel alone, whatever else the entry point enqueues.  One-shot; two hipEvent_t handles
 * created with timing enabled (null, null cancels). */
void vq_profile_next_search(void *ev_start, void *ev_stop);
/* 1 when the pruned search of a (K, d) codebook (vq_prune_supported, `sorted` given) settles its near-ties inside
 * the filter kernel, so that its phase 2 launches nothing; 0 when it queues them for the fix-up kernel. */
int vq_prune_resolves_in_filter(int K, int d);

/* One LBG assignment pass, fused.  Replaces VQ_Full.py:77-90 up to the division: assigns every
 * vector as vq_encode_l1 does (labels: N int32, min_dist: N doubles, both overwritten), ADDS into
 * sums[k*d+e] the byte sums of cluster k and into counts[k] its member count, and OVERWRITES
 * dist_sum[0] with the sum of min_dist taken in numpy's order (see vq_numpy_sum_f64), so that
 * dist_sum / N is bit for bit the np.mean of VQ_Full.py:90.  sums and counts are accumulated
 * into, so the caller zeroes them first (and may all-reduce them across GPUs afterwards).
 * chunk_sums (nullable) receives the ceil(N / 8192) per-chunk sums dist_sum is the left-to-right
 * total of; ranks that hold consecutive, 8192-aligned shards can concatenate them to reproduce
 * the single-process value.  `sorted` is null or the buffer vq_blocks_sort / vq_blocks_sort_img filled
 * from the same blocks (used by VQ_MODE_FAST only; when it is given and vq_prune_supported(K, d),
 * `blocks` is not read and may be null). */
int vq_lbg_accum(const uint8_t *blocks, const void *sorted, size_t sorted_bytes, int64_t N, int d,
                 const double *cb, int K, uint64_t *sums, uint64_t *counts, int32_t *labels, double *min_dist,
                 double *dist_sum, double *chunk_sums, int mode, void *ws, size_t ws_bytes, void *stream);

/* A whole LBG iteration in one call, with the stop rule on the device.  Replaces one trip of the loop at
 * VQ_Full.py:76-98 on one GPU: vq_lbg_accum (sums / counts are zeroed here first and serve as scratch), then
 * the centroid update IN PLACE in cb (VQ_Full.py:81-88) and the rule of VQ_Full.py:89-98: the distortion
 * D = (numpy-order distance sum) / n_total is stored in the stage's state, and the iteration is marked
 * converged when |prev - D| / prev < threshold (never at a stage's first iteration).  The state lives in
 * device memory (vq_lbg_state_bytes(max_iterations) bytes; vq_lbg_stage_begin resets it at the start of
 * every split stage), so the host can enqueue iteration i + 1 before it has read the outcome of iteration
 * i: once an iteration has met the rule, the kernels of later iterations of that stage return at once
 * and leave cb and the records untouched.  Record i sits at byte 64 + 16 * i of the state:
 * { double distortion; uint32 converged; uint32 valid } -- valid = 0 marks an iteration that was skipped.
 * A negative threshold never converges (forced iteration counts).  Consecutive iterations of a stage issued by ONE
 * host thread with the same ws, state, N, K and d, and no other entry point given that ws by that thread in between,
 * skip re-zeroing the tables of ws (the previous iteration leaves them zeroed); any other call sequence of that
 * thread costs one extra fill.  The note is per host thread: a workspace must not be shared between host threads,
 * nor written by the caller, while a stage runs.  labels and min_dist are scratch of the iteration: what they hold
 * afterwards is unspecified (with a `sorted` buffer they stay in the order of the sorted vectors); use vq_encode_l1 /
 * vq_lbg_accum for labels and distances in the caller's order. */
/* 1 when vq_lbg_iterate in VQ_MODE_FAST with a `sorted` buffer runs this (K, d) as one fused pass (search and
 * accumulation in one kernel): labels may then be null (none are written, which saves their stores during
 * training) and min_dist is scratch whose contents are unspecified afterwards.  0: labels is required. */
int vq_lbg_pass_fused(int K, int d);
size_t vq_lbg_state_bytes(int max_iterations);
int vq_lbg_stage_begin(void *state, size_t state_bytes, int max_iterations, void *stream);
int vq_lbg_iterate(const uint8_t *blocks, const void *sorted, size_t sorted_bytes, int64_t N, int d, double *cb, int K,
                   int64_t n_total, double threshold, int iteration, uint64_t *sums, uint64_t *counts, int32_t *labels,
                   double *min_dist, int mode, void *state, size_t state_bytes, void *ws, size_t ws_bytes, void *stream);

/* The same iteration for block vectors sharded over several GPUs (one process per GPU), split around the exchange so
 * that nothing returns to the host.  No counterpart in the reference (VQ_Full.py:76-98 is one process).
 * vq_lbg_iterate_local: this rank's part of VQ_Full.py:77-90 -- sums / counts of its N vectors (overwritten),
 * chunk_sums[0 .. ceil(N / 8192)) = the per-chunk sums of its winning distances, local_total[0] (nullable) = their
 * left-to-right total.  The three outputs may point into one buffer that is then handed to an all-gather as it is.
 * vq_lbg_iterate_finish: given the sums / counts of all ranks -- `records` = 1: already added up (an all-reduce);
 * `records` = the number of ranks: one record per rank, record r at sums[r * record_stride] / counts[r * record_stride]
 * (64-bit words; what an all-gather of [sums | counts | ...] leaves), added here -- and every rank's chunk sums (rank
 * r's h_chunk_counts[r] <= chunk_stride values at chunk_sums_all[r * chunk_stride]; device memory; world <= 64), adds
 * the chunk sums left to right in rank order -- bit for bit the single-process distance sum when every rank but the
 * last holds a multiple of 8192 vectors; ranks whose shards are not aligned pass their local_total as a single value
 * -- then updates cb in place and applies the stop rule for `iteration` exactly as vq_lbg_iterate does.  ws (nullable):
 * the workspace the calling thread gave vq_lbg_iterate_local for this iteration; with it, for the shapes of the
 * one-workgroup preparation (d <= 16, K <= 512), the updated codebook is prepared for the next iteration in the same
 * launch and the next vq_lbg_iterate_local skips that step.  Both calls carry the stage's gate: enqueued for an
 * iteration past the one that converged they leave cb and the records untouched (the collectives between them
 * still run, on every rank alike).  labels as in vq_lbg_iterate (nullable when vq_lbg_pass_fused). */
int vq_lbg_iterate_local(const uint8_t *blocks, const void *sorted, size_t sorted_bytes, int64_t N, int d, const double *cb,
                         int K, int iteration, uint64_t *sums, uint64_t *counts, double *chunk_sums, double *local_total,
                         int32_t *labels, double *min_dist, int mode, void *state, size_t state_bytes, void *ws,
                         size_t ws_bytes, void *stream);
int vq_lbg_iterate_finish(const uint64_t *sums, const uint64_t *counts, int records, int64_t record_stride,
                          const double *chunk_sums_all, int world, int64_t chunk_stride, const int *h_chunk_counts, double *cb,
                          int K, int d, int64_t n_total, double threshold, int iteration, void *state, size_t state_bytes,
                          void *ws, size_t ws_bytes, void *stream);

/* Sum of an fp64 vector in the order numpy.add.reduce uses on a contiguous array (what np.mean
 * does before dividing, VQ_Full.py:90): pairwise summation -- runs of up to 128 elements with
 * eight interleaved partial sums, halved recursively -- inside chunks of 8192 elements, the chunk
 * results added left to right.  out[0] receives the sum, chunk_sums (ceil(N / 8192) + 1 doubles,
 * required) the per-chunk sums. */
int vq_numpy_sum_f64(const double *v, int64_t N, double *out, double *chunk_sums, void *stream);

/* Per-cluster byte sums and counts for a given labelling (the inputs of np.mean(members, axis=0),
 * VQ_Full.py:82-84).  Accumulates into sums / counts like vq_lbg_accum. */
int vq_accum_labels(const uint8_t *blocks, int64_t N, int d, const int32_t *labels, int K,
                    uint64_t *sums, uint64_t *counts, void *stream);

/* Centroid update.  Replaces VQ_Full.py:81-86: cb_new[k] = sums[k] / counts[k] (one IEEE fp64
 * divide of the exact integer sum) for a non-empty cluster, cb_old[k] for an empty one.
 * cb_new may alias cb_old. */
int vq_lbg_finalize(const uint64_t *sums, const uint64_t *counts, const double *cb_old, int K, int d,
                    double *cb_new, void *stream);

/* Split.  Replaces VQ_Full.py:71-73: cb2[0:K] = cb * (1 + eps), cb2[K:2K] = cb * (1 - eps);
 * cb2 holds 2*K*d doubles and must not overlap cb. */
int vq_lbg_split(const double *cb, int K, int d, double eps, double *cb2, void *stream);

/* Decode.  Replaces the loop of Codebook.decompress (VQ_Full.py:157-169):
 * img[r*bh+i, q*bw+j, ch] = (uint8) trunc(cb[labels[r*nc+q], i, j, ch]) with the conversion
 * wrapping modulo 256 as numpy's does; img is (nr*bh, nc*bw, C) uint8 (padded size, not cropped).
 * A label outside [0, K) is an error of the caller (the reference raises IndexError, VQ_Full.py:162, and the
 * Python wrapper checks labels before it calls): the kernels decode it as codeword K - 1 rather than read
 * outside the codebook. */
int vq_decode_u8(const int32_t *labels, int nr, int nc, const double *cb, int K, int bh, int bw, int C,
                 uint8_t *img, void *ws, size_t ws_bytes, void *stream);

/* Label bit-packer.  Replaces VQ_Full.py:134-147: labels in order, ceil(log2 K) bits each,
 * MSB first, last byte zero-padded on the right.  out holds vq_packed_bytes(N, K) bytes. */
size_t vq_packed_bytes(int64_t N, int K);
int vq_pack_labels(const int32_t *labels, int64_t N, int K, uint8_t *out, void *stream);

/* Inverse of vq_pack_labels (no counterpart in the reference, whose decoder reads the JSON
 * labels; provided so that the .bin stream can be decoded on the GPU). */
int vq_unpack_labels(const uint8_t *packed, int64_t N, int K, int32_t *labels, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VQ_HIP_H */
