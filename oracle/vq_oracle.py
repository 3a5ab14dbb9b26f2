"""Python face of the CPU oracle (oracle/vq_oracle.c) plus a SciPy restatement of the same path.

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by bench.py's
cpu_baseline leg -- never by the vector_quantization_amd package.

Parity status: pinned against outputs of the reference itself (tests/golden/*.npz, produced by
tests/golden/make_golden.py); see tests/test_oracle_golden.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvq_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "vq_oracle.c")
    if force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libvq_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        vp, i, i64, dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
        L.vqo_blocks_u8.argtypes = [vp, i, i, i, i, i, i, vp]
        L.vqo_blocks_u8.restype = None
        L.vqo_encode_l1.argtypes = [vp, i64, i, vp, i, vp, vp]
        L.vqo_encode_l1.restype = None
        L.vqo_pairwise_sum.argtypes = [vp, i64]
        L.vqo_pairwise_sum.restype = dbl
        L.vqo_numpy_sum.argtypes = [vp, i64]
        L.vqo_numpy_sum.restype = dbl
        L.vqo_colsum_u8.argtypes = [vp, i64, i, vp]
        L.vqo_colsum_u8.restype = None
        L.vqo_split.argtypes = [vp, i, i, dbl, vp]
        L.vqo_split.restype = None
        L.vqo_accumulate.argtypes = [vp, i64, i, vp, i, vp, vp]
        L.vqo_accumulate.restype = None
        L.vqo_finalize.argtypes = [vp, vp, vp, i, i, vp]
        L.vqo_finalize.restype = None
        L.vqo_lbg_iter.argtypes = [vp, i64, i, vp, i, vp, vp, vp, vp, vp]
        L.vqo_lbg_iter.restype = dbl
        L.vqo_lbg_train.argtypes = [vp, i64, i, i, dbl, dbl, i, vp, vp, vp, vp, vp, i, vp]
        L.vqo_lbg_train.restype = i
        L.vqo_decode_u8.argtypes = [vp, i, i, vp, i, i, i, i, vp]
        L.vqo_decode_u8.restype = None
        L.vqo_bits_needed.argtypes = [i]
        L.vqo_bits_needed.restype = i
        L.vqo_pack_labels.argtypes = [vp, i64, i, vp]
        L.vqo_pack_labels.restype = ctypes.c_size_t
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def blocks_u8(img, bh, bw, pad_mode=0):
    img = _c(img, np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    H, W, C = img.shape
    nr, nc = -(-H // bh), -(-W // bw)
    out = np.empty((nr * nc, bh * bw * C), dtype=np.uint8)
    lib().vqo_blocks_u8(_p(img), H, W, C, bh, bw, pad_mode, _p(out))
    return out


def encode_l1(blocks, cb, want_dist=False):
    blocks = _c(blocks, np.uint8)
    cb = _c(cb, np.float64)
    N, d = blocks.shape
    labels = np.empty(N, dtype=np.int32)
    mind = np.empty(N, dtype=np.float64)
    lib().vqo_encode_l1(_p(blocks), N, d, _p(cb), cb.shape[0], _p(labels), _p(mind))
    return (labels, mind) if want_dist else labels


def pairwise_sum(a):
    a = _c(a, np.float64)
    return lib().vqo_pairwise_sum(_p(a), a.size)


def numpy_sum(a):
    a = _c(a, np.float64)
    return lib().vqo_numpy_sum(_p(a), a.size)


def colsum_u8(blocks):
    blocks = _c(blocks, np.uint8)
    N, d = blocks.shape
    sums = np.empty(d, dtype=np.uint64)
    lib().vqo_colsum_u8(_p(blocks), N, d, _p(sums))
    return sums


def split(cb, eps):
    cb = _c(cb, np.float64)
    K, d = cb.shape
    out = np.empty((2 * K, d), dtype=np.float64)
    lib().vqo_split(_p(cb), K, d, eps, _p(out))
    return out


def accumulate(blocks, labels, K):
    blocks = _c(blocks, np.uint8)
    labels = _c(labels, np.int32)
    N, d = blocks.shape
    sums = np.empty((K, d), dtype=np.uint64)
    counts = np.empty(K, dtype=np.uint64)
    lib().vqo_accumulate(_p(blocks), N, d, _p(labels), K, _p(sums), _p(counts))
    return sums, counts


def finalize(sums, counts, cb_old):
    cb_old = _c(cb_old, np.float64)
    K, d = cb_old.shape
    sums = _c(sums, np.uint64)
    counts = _c(counts, np.uint64)
    out = np.empty((K, d), dtype=np.float64)
    lib().vqo_finalize(_p(sums), _p(counts), _p(cb_old), K, d, _p(out))
    return out


def lbg_iter(blocks, cb):
    """Returns dict(cb_out, labels, min_dist, sums, counts, distortion)."""
    blocks = _c(blocks, np.uint8)
    cb = _c(cb, np.float64)
    N, d = blocks.shape
    K = cb.shape[0]
    out = np.empty((K, d), dtype=np.float64)
    labels = np.empty(N, dtype=np.int32)
    mind = np.empty(N, dtype=np.float64)
    sums = np.empty((K, d), dtype=np.uint64)
    counts = np.empty(K, dtype=np.uint64)
    D = lib().vqo_lbg_iter(_p(blocks), N, d, _p(cb), K, _p(out), _p(labels), _p(mind), _p(sums), _p(counts))
    return dict(cb_out=out, labels=labels, min_dist=mind, sums=sums, counts=counts, distortion=D)


def lbg_train(blocks, k, eps=0.01, thr=0.001, max_iter=100):
    """Returns dict(codebook, iters_per_stage, converged_iter, distortions); raises ValueError if k > N."""
    blocks = _c(blocks, np.uint8)
    N, d = blocks.shape
    Kf = 1
    while Kf < k:
        Kf *= 2
    cb = np.empty((Kf, d), dtype=np.float64)
    iters = np.zeros(64, dtype=np.int32)
    conv = np.zeros(64, dtype=np.int32)
    nst = ctypes.c_int(0)
    nd = ctypes.c_int(0)
    cap = 64 * max(max_iter, 1)
    dist = np.zeros(cap, dtype=np.float64)
    K = lib().vqo_lbg_train(_p(blocks), N, d, k, eps, thr, max_iter, _p(cb), _p(iters), _p(conv),
                            ctypes.byref(nst), _p(dist), cap, ctypes.byref(nd))
    if K < 0:
        raise ValueError(f"Invalid quantization level k={k}: cannot exceed the total number of image blocks ({N}).")
    return dict(codebook=cb[:K], iters_per_stage=iters[: nst.value].copy(), converged_iter=conv[: nst.value].copy(),
                distortions=dist[: nd.value].copy())


def decode_u8(labels, cb, bh, bw, C):
    labels = _c(labels, np.int32)
    nr, nc = labels.shape
    cb = _c(cb, np.float64).reshape(-1, bh * bw * C)
    img = np.empty((nr * bh, nc * bw, C), dtype=np.uint8)
    lib().vqo_decode_u8(_p(labels), nr, nc, _p(cb), cb.shape[0], bh, bw, C, _p(img))
    return img


def bits_needed(K):
    return lib().vqo_bits_needed(int(K))


def pack_labels(labels, K):
    labels = _c(labels, np.int32).reshape(-1)
    bits = bits_needed(K)
    out = np.empty((labels.size * bits + 7) // 8, dtype=np.uint8)
    n = lib().vqo_pack_labels(_p(labels), labels.size, int(K), _p(out))
    assert n == out.size
    return out


# ---------------------------------------------------------------------------------------------------
# SciPy restatement: literally the calls the reference makes (cdist 'cityblock' + argmin, VQ_Full.py:
# 77-78 / 124-125), applied to N-chunks so that the N x K distance matrix stays small.  Used as a
# second opinion on the C restatement and as the benchmark's CPU baseline.
# ---------------------------------------------------------------------------------------------------
def scipy_encode_chunked(blocks, cb, chunk=65536, want_dist=False):
    from scipy.spatial.distance import cdist
    N = blocks.shape[0]
    labels = np.empty(N, dtype=np.int64)
    mind = np.empty(N, dtype=np.float64) if want_dist else None
    for lo in range(0, N, chunk):
        dist = cdist(blocks[lo:lo + chunk], cb, metric="cityblock")
        lab = np.argmin(dist, axis=1)
        labels[lo:lo + chunk] = lab
        if want_dist:
            mind[lo:lo + chunk] = dist[np.arange(len(dist)), lab]
    return (labels, mind) if want_dist else labels


def scipy_lbg_iter_chunked(blocks, cb, chunk=65536):
    """One iteration of VQ_Full.py:77-90 with the distance matrix chunked; returns (new_cb, distortion)."""
    labels, mind = scipy_encode_chunked(blocks, cb, chunk, want_dist=True)
    new_cb = np.zeros_like(cb)
    for idx in range(len(cb)):
        members = blocks[labels == idx]
        new_cb[idx] = np.mean(members, axis=0) if len(members) > 0 else cb[idx]
    return new_cb, np.mean(mind)
