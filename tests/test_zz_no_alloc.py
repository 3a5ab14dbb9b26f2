"""vq_encode_l1_static -- the entry point that allocates nothing (its scratch is a static device array
inside the library) -- against vq_encode_l1 with a caller-owned workspace, through the C ABI.

Collected last on purpose (file name): the round's end runs `pytest -x`.  Passes alone in a fresh
process and after the other GPU tests.
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _case(rng, N, K, d):
    cb = np.clip(rng.random((K, 1)) * 255.0 + rng.normal(0, 20, (K, d)), 0, 300)
    blocks = np.clip(cb[rng.integers(0, K, N)] + rng.normal(0, 10, (N, d)), 0, 255).astype(np.uint8)
    blocks[: N // 50] = 0                               # a run of identical vectors
    cb[K // 2] = cb[K // 3]                             # a duplicate codeword: exact ties
    return blocks, cb


def _both(lib, check, blocks_h, cb_h, mode):
    dev = torch.device("cuda:0")
    blocks = torch.from_numpy(blocks_h).to(dev)
    cb = torch.from_numpy(cb_h).to(dev)
    N, d = blocks.shape
    K = cb.shape[0]
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    # the existing entry point, caller-owned workspace
    need = lib.vq_workspace_bytes(K, d, N)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    lab_a = torch.full((N,), -7, dtype=torch.int32, device=dev)
    dist_a = torch.full((N,), -1.0, dtype=torch.float64, device=dev)
    check(lib.vq_encode_l1(_p(blocks), N, d, _p(cb), K, _p(lab_a), _p(dist_a), mode, _p(ws), need, stream), "vq_encode_l1")
    # the static-scratch entry point: outputs only, nothing else passed in
    lab_b = torch.full((N,), -7, dtype=torch.int32, device=dev)
    dist_b = torch.full((N,), -1.0, dtype=torch.float64, device=dev)
    # one warm-up call (module load, the look-up of the static array's address, function attributes), then the call
    # that is checked: the device's free memory as the driver reports it (hipMemGetInfo -- it sees any hipMalloc,
    # also one made behind torch's back) must be the same before and after
    warm = torch.empty_like(lab_b)
    check(lib.vq_encode_l1_static(_p(blocks), N, d, _p(cb), K, _p(warm), None, mode, stream), "vq_encode_l1_static")
    torch.cuda.synchronize(dev)
    free_before = torch.cuda.mem_get_info(dev)[0]
    check(lib.vq_encode_l1_static(_p(blocks), N, d, _p(cb), K, _p(lab_b), _p(dist_b), mode, stream), "vq_encode_l1_static")
    torch.cuda.synchronize(dev)
    assert torch.cuda.mem_get_info(dev)[0] == free_before
    return lab_a.cpu().numpy(), dist_a.cpu().numpy(), lab_b.cpu().numpy(), dist_b.cpu().numpy()


def test_static_scratch_entry_matches_workspace_entry(oracle):
    from vector_quantization_amd._lib import lib, check, last_error, VQ_E_UNSUPPORTED
    rng = np.random.default_rng(2024)
    # two calls in a row on the same GPU's scratch, the second smaller (fewer vectors, smaller codebook,
    # another dimension): whatever the first call left in the array must not leak into the second
    for N, K, d in [(150000, 256, 16), (9001, 40, 12)]:
        blocks, cb = _case(rng, N, K, d)
        la, da, lb, db = _both(lib, check, blocks, cb, mode=1)
        assert np.array_equal(la, lb) and np.array_equal(da, db)          # bit-exact: labels and fp64 distances
        want_l, want_d = oracle.encode_l1(blocks, cb, want_dist=True)
        assert np.array_equal(lb, want_l) and np.array_equal(db, want_d)
    # the all-fp64 mode goes through the same entry point (it needs no scratch)
    blocks, cb = _case(rng, 3000, 17, 16)
    la, da, lb, db = _both(lib, check, blocks, cb, mode=0)
    assert np.array_equal(la, lb) and np.array_equal(da, db)
    # beyond the static array: refused, with the limit in the message, before anything is launched
    dev = torch.device("cuda:0")
    big = torch.zeros((1 << 22, 16), dtype=torch.uint8, device=dev)
    cbt = torch.from_numpy(cb[:, :16].copy()).to(dev)
    lab = torch.empty(1 << 22, dtype=torch.int32, device=dev)
    rc = lib.vq_encode_l1_static(_p(big), 1 << 22, 16, _p(cbt), cbt.shape[0], _p(lab), None, 1, None)
    assert rc == VQ_E_UNSUPPORTED and "static array holds" in last_error()


def test_static_scratch_entry_under_stream_capture(oracle):
    """The caller the entry point is for: a HIP graph capture, where an allocation would fail.  After one warm-up call
    outside the capture the call is captured, the graph replayed twice, and the labels checked against the oracle."""
    from vector_quantization_amd._lib import lib, check
    rng = np.random.default_rng(7)
    blocks_h, cb_h = _case(rng, 60000, 128, 16)
    dev = torch.device("cuda:0")
    blocks = torch.from_numpy(blocks_h).to(dev)
    cb = torch.from_numpy(cb_h).to(dev)
    N, d = blocks.shape
    K = cb.shape[0]
    lab = torch.full((N,), -7, dtype=torch.int32, device=dev)
    dist = torch.full((N,), -1.0, dtype=torch.float64, device=dev)
    assert lib.vq_static_scratch_bytes() == 8 << 20
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    check(lib.vq_encode_l1_static(_p(blocks), N, d, _p(cb), K, _p(lab), _p(dist), 1, stream), "warm-up")
    torch.cuda.synchronize(dev)
    lab.fill_(-7)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        check(lib.vq_encode_l1_static(_p(blocks), N, d, _p(cb), K, _p(lab), _p(dist), 1, st), "captured call")
    want_l, want_d = oracle.encode_l1(blocks_h, cb_h, want_dist=True)
    for _ in range(2):
        lab.fill_(-7)
        g.replay()
        torch.cuda.synchronize(dev)
        assert np.array_equal(lab.cpu().numpy(), want_l) and np.array_equal(dist.cpu().numpy(), want_d)
