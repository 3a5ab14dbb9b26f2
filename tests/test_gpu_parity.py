"""Parity of the HIP path (through the C ABI, via HipEngine) against the CPU oracle and the
golden fixtures produced by the reference.  Bar: bit-exact everywhere -- labels, pixels, packed
bits, integer sums, fp64 codebooks, per-vector distances and the distortion (the device adds the
distances in numpy's own summation order).
"""
import numpy as np
import pytest
import torch

from vqtest_util import TRAIN_CASES, load_golden, stage_end_codebooks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from vector_quantization_amd import HipEngine
    return HipEngine("cuda:0", mode="fast")


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to("cuda:0")


# ---------------------------------------------------------------------------------------------------
# block extraction
# ---------------------------------------------------------------------------------------------------
def test_blocks_golden(eng, train_case):
    g = train_case
    b = eng.blocks_from_image(dev(g["img"]), int(g["bh"]), int(g["bw"]))
    assert np.array_equal(b.cpu().numpy(), g["blocks"])


@pytest.mark.parametrize("shape,bh,bw", [((7, 5, 3), 2, 2), ((33, 17, 1), 4, 4), ((64, 64, 3), 8, 8), ((9, 9, 1), 3, 3),
                                         ((5, 7, 3), 1, 1), ((16, 20, 2), 4, 2), ((3, 3, 1), 4, 4), ((130, 67, 3), 3, 5),
                                         # row strips (segments of whole words, 16-byte vectors, at least 64 blocks per
                                         # block row): several strips, unaligned rows, padding on both edges
                                         ((37, 1030, 3), 4, 4), ((64, 4090, 3), 4, 4), ((33, 520, 1), 8, 8),
                                         ((70, 515, 3), 8, 8), ((9, 301, 4), 4, 1), ((40, 2050, 1), 4, 4),
                                         # strips for segments that are not whole words
                                         ((9, 300, 1), 2, 2), ((11, 301, 1), 2, 2), ((10, 201, 3), 2, 2), ((10, 211, 1), 3, 3),
                                         ((5, 100, 3), 1, 1), ((4, 9001, 1), 2, 2), ((2049, 2051, 1), 2, 2), ((33, 1000, 3), 5, 3)])
@pytest.mark.parametrize("pad_mode", [0, 1])
def test_blocks_ragged(eng, oracle, shape, bh, bw, pad_mode):
    rng = np.random.default_rng(hash((shape, bh, bw)) & 0xffff)
    img = rng.integers(0, 256, size=shape, dtype=np.uint8)
    got = eng.blocks_from_image(dev(img), bh, bw, pad_mode).cpu().numpy()
    assert np.array_equal(got, oracle.blocks_u8(img, bh, bw, pad_mode))


def test_colsum(eng, oracle, train_case):
    g = train_case
    s = eng.colsum(dev(g["blocks"])).cpu().numpy().astype(np.uint64)
    assert np.array_equal(s, oracle.colsum_u8(g["blocks"]))


# ---------------------------------------------------------------------------------------------------
# nearest codeword
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", [0, 1])
def test_encode_golden(eng, train_case, mode):
    g = train_case
    labels, mind = eng.encode(dev(g["blocks"]), dev(g["codebook"]), want_dist=True, mode=mode)
    assert np.array_equal(labels.cpu().numpy().astype(np.int64).reshape(g["labels"].shape), g["labels"])
    if "it_cb_in" in g:
        labels, mind = eng.encode(dev(g["blocks"]), dev(g["it_cb_in"]), want_dist=True, mode=mode)
        assert np.array_equal(labels.cpu().numpy().astype(np.int64), g["it_labels"])
        assert np.array_equal(mind.cpu().numpy(), g["it_mind"])


@pytest.mark.parametrize("mode", [0, 1])
def test_encode_every_stage_codebook(eng, oracle, train_case, mode):
    g = train_case
    blocks = dev(g["blocks"])
    for cb in stage_end_codebooks(g):
        want_l, want_d = oracle.encode_l1(g["blocks"], cb, want_dist=True)
        got_l, got_d = eng.encode(blocks, dev(cb), want_dist=True, mode=mode)
        assert np.array_equal(got_l.cpu().numpy(), want_l)
        assert np.array_equal(got_d.cpu().numpy(), want_d)


DIMS = [1, 2, 3, 4, 5, 8, 11, 12, 16, 17, 24, 27, 32, 48, 63, 64, 75, 96, 100, 128, 147, 192, 200]


@pytest.mark.parametrize("d", DIMS)
def test_encode_random_dims(eng, oracle, d):
    """Every template bucket of the fast kernel, odd dimensions, and d beyond the fast kernels."""
    rng = np.random.default_rng(d)
    N, K = 3001, 37
    blocks = rng.integers(0, 256, size=(N, d), dtype=np.uint8)
    cb = rng.random((K, d)) * 255.0
    want_l, want_d = oracle.encode_l1(blocks, cb, want_dist=True)
    for mode in (0, 1):
        got_l, got_d = eng.encode(dev(blocks), dev(cb), want_dist=True, mode=mode)
        assert np.array_equal(got_l.cpu().numpy(), want_l), f"labels differ, d={d} mode={mode}"
        assert np.array_equal(got_d.cpu().numpy(), want_d), f"distances differ, d={d} mode={mode}"


@pytest.mark.parametrize("K", [1, 2, 3, 5, 64, 65, 255, 256, 257, 1024, 4096, 4097, 6000])
def test_encode_codebook_sizes(eng, oracle, K):
    rng = np.random.default_rng(K)
    N, d = 2500, 16
    blocks = rng.integers(0, 256, size=(N, d), dtype=np.uint8)
    cb = rng.random((K, d)) * 255.0
    want = oracle.encode_l1(blocks, cb)
    got = eng.encode(dev(blocks), dev(cb)).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("K,d", [(300, 48), (70, 48), (130, 192), (1030, 16), (3000, 12), (33, 64), (4096, 3)])
def test_encode_ragged_codebook_tiles(eng, oracle, K, d):
    """Codebook sizes that are not a multiple of the codeword batch / LDS tile of their kernel variant."""
    rng = np.random.default_rng(K * 1000 + d)
    N = 4099
    cb = rng.random((K, d)) * 255.0
    pick = rng.integers(0, K, N)
    blocks = np.clip(cb[pick] + rng.normal(0, 12, (N, d)), 0, 255).astype(np.uint8)
    want_l, want_d = oracle.encode_l1(blocks, cb, want_dist=True)
    got_l, got_d = eng.encode(dev(blocks), dev(cb), want_dist=True, mode=1)
    assert np.array_equal(got_l.cpu().numpy(), want_l)
    assert np.array_equal(got_d.cpu().numpy(), want_d)
    assert len(np.unique(want_l)) > min(K, 30) // 2          # the labels really spread over the codebook


def test_encode_ties_and_duplicates(eng, oracle):
    """Exact ties: duplicated codewords, integer codewords equidistant from integer pixels, and a
    flat image; the lowest index must win, as numpy.argmin does."""
    rng = np.random.default_rng(3)
    d, K, N = 16, 64, 4096
    cb = np.round(rng.random((K, d)) * 255.0)
    cb[7] = cb[3]                       # duplicate rows
    cb[40] = cb[3]
    cb[10] = 0.0
    cb[11] = 0.0                        # the zero twin of an all-zero centroid's split
    cb[20] = cb[21][::-1].copy()        # a permutation: equal distance to a constant vector
    blocks = rng.integers(0, 256, size=(N, d), dtype=np.uint8)
    blocks[:512] = 0
    blocks[512:1024] = 128
    blocks[1024:1100] = cb[3].astype(np.uint8)
    # midpoints between two integer codewords: exact fp64 ties by construction
    mid = ((cb[30] + cb[31]) / 2).astype(np.uint8)
    blocks[1100:1200] = mid
    want_l, want_d = oracle.encode_l1(blocks, cb, want_dist=True)
    for mode in (0, 1):
        got_l, got_d = eng.encode(dev(blocks), dev(cb), want_dist=True, mode=mode)
        assert np.array_equal(got_l.cpu().numpy(), want_l)
        assert np.array_equal(got_d.cpu().numpy(), want_d)


def test_encode_near_ties_need_fp64(eng, oracle):
    """Codeword pairs whose distances to a vector differ by a few ulps of fp64: neither a 16-bit
    filter nor fp32 can rank them; the fp64 re-check must reproduce the sequential-sum ranking."""
    rng = np.random.default_rng(11)
    d, N = 16, 2048
    blocks = rng.integers(0, 256, size=(N, d), dtype=np.uint8)
    base = rng.random((32, d)) * 255.0
    cbs = [base]
    for scale in (1e-13, 3e-13, 1e-12, 1e-9, 1e-6, 1e-3):
        pert = base.copy()
        # move mass between two coordinates so the L1 distance to most vectors changes by ~scale only
        pert[:, 0] += scale
        pert[:, 1] -= scale
        cbs.append(pert)
    cb = np.vstack(cbs[::-1])            # the perturbed copies come FIRST, the originals last
    want_l, want_d = oracle.encode_l1(blocks, cb, want_dist=True)
    for mode in (0, 1):
        got_l, got_d = eng.encode(dev(blocks), dev(cb), want_dist=True, mode=mode)
        assert np.array_equal(got_l.cpu().numpy(), want_l)
        assert np.array_equal(got_d.cpu().numpy(), want_d)


def test_encode_codebook_out_of_pixel_range(eng, oracle):
    """Entries above 255 (255 * 1.01 kept by an empty cluster), far above 65535/2^s, and negative."""
    rng = np.random.default_rng(5)
    d, N = 12, 3000
    blocks = rng.integers(0, 256, size=(N, d), dtype=np.uint8)
    for top in (257.55, 300.0, 600.0, 70000.0, -5.0):
        cb = rng.random((16, d)) * 255.0
        cb[2] = top
        cb[5, 3] = top
        want = oracle.encode_l1(blocks, cb)
        got = eng.encode(dev(blocks), dev(cb)).cpu().numpy()
        assert np.array_equal(got, want), f"top={top}"


@pytest.mark.parametrize("K,d", [(40, 48), (600, 16), (1100, 8)])
def test_encode_codebook_out_of_pixel_range_other_shapes(eng, oracle, K, d):
    """The same through the multi-workgroup codebook preparation, plain and pruned search."""
    rng = np.random.default_rng(K + d)
    N = 2500
    blocks = rng.integers(0, 256, size=(N, d), dtype=np.uint8)
    for top in (257.55, 600.0, 70000.0, -5.0):
        cb = rng.random((K, d)) * 255.0
        cb[2] = top
        cb[5, 3] = top
        cb[K - 1] = cb[7]                                   # and a duplicate row
        want_l, want_d = oracle.encode_l1(blocks, cb, want_dist=True)
        b = dev(blocks)
        for srt in (None, eng.sort_blocks(b)):
            got_l, got_d = eng.encode(b, dev(cb), want_dist=True, sorted=srt)
            assert np.array_equal(got_l.cpu().numpy(), want_l), f"top={top} sorted={srt is not None}"
            assert np.array_equal(got_d.cpu().numpy(), want_d), f"top={top} sorted={srt is not None}"


def test_encode_empty_and_tiny(eng, oracle):
    cb = np.arange(32, dtype=np.float64).reshape(2, 16)
    got = eng.encode(dev(np.zeros((0, 16), dtype=np.uint8)), dev(cb))
    assert got.numel() == 0
    one = np.full((1, 16), 9, dtype=np.uint8)
    assert np.array_equal(eng.encode(dev(one), dev(cb)).cpu().numpy(), oracle.encode_l1(one, cb))


# ---------------------------------------------------------------------------------------------------
# LBG pieces
# ---------------------------------------------------------------------------------------------------
def _check_iteration(eng, oracle, blocks, cb, mode):
    want = oracle.lbg_iter(blocks, cb)
    K, d = cb.shape
    acc, dist, labels = eng.lbg_accum(dev(blocks), dev(cb), want_labels=True, mode=mode)
    acc_h = acc.cpu().numpy().astype(np.uint64)
    assert np.array_equal(labels.cpu().numpy(), want["labels"])
    assert np.array_equal(acc_h[: K * d].reshape(K, d), want["sums"])
    assert np.array_equal(acc_h[K * d:], want["counts"])
    assert dist.item() == oracle.numpy_sum(want["min_dist"])        # numpy's summation order, bit for bit
    assert dist.item() / len(blocks) == want["distortion"]
    new_cb = eng.lbg_finalize(acc, dev(cb)).cpu().numpy()
    assert np.array_equal(new_cb, want["cb_out"])
    return new_cb


@pytest.mark.parametrize("mode", [0, 1])
def test_lbg_iteration_golden(eng, oracle, train_case, mode):
    g = train_case
    if "it_cb_in" not in g:
        pytest.skip("no iteration KAT")
    new_cb = _check_iteration(eng, oracle, g["blocks"], g["it_cb_in"], mode)
    assert np.array_equal(new_cb, g["it_cb_out"])


@pytest.mark.parametrize("K,d", [(2, 16), (8, 4), (256, 16), (1024, 16), (2048, 16), (64, 48), (512, 192), (16, 200)])
def test_lbg_iteration_shapes(eng, oracle, K, d):
    """LDS-table, large-LDS-table and global-atomic accumulation paths; d beyond the fast kernels."""
    rng = np.random.default_rng(K * 1000 + d)
    N = 20000
    blocks = rng.integers(0, 256, size=(N, d), dtype=np.uint8)
    blocks[:3000] = blocks[0]            # a heavily populated cluster (LDS atomic conflicts)
    cb = rng.random((K, d)) * 255.0
    _check_iteration(eng, oracle, blocks, cb, 1)


def test_accum_labels(eng, oracle):
    rng = np.random.default_rng(9)
    N, d, K = 5000, 12, 16
    blocks = rng.integers(0, 256, size=(N, d), dtype=np.uint8)
    labels = rng.integers(0, K, size=N).astype(np.int32)
    acc = eng.accum_labels(dev(blocks), dev(labels), K).cpu().numpy().astype(np.uint64)
    sums, counts = oracle.accumulate(blocks, labels, K)
    assert np.array_equal(acc[: K * d].reshape(K, d), sums)
    assert np.array_equal(acc[K * d:], counts)


def test_split_bitexact(eng, oracle):
    rng = np.random.default_rng(2)
    cb = rng.random((300, 48)) * 255.0
    got = eng.lbg_split(dev(cb), 0.01).cpu().numpy()
    assert np.array_equal(got, oracle.split(cb, 0.01))
    assert np.array_equal(got, np.vstack((cb * (1 + 0.01), cb * (1 - 0.01))))


def test_finalize_division_bitexact(eng, oracle):
    """One IEEE fp64 divide per element: compare against the host's correctly rounded division on
    many (sum, count) pairs, including sums beyond 2^32 and empty clusters."""
    rng = np.random.default_rng(4)
    K, d = 4096, 16
    counts = rng.integers(0, 1 << 28, size=K).astype(np.uint64)
    counts[::17] = 0
    sums = (rng.integers(0, 256, size=(K, d)).astype(np.uint64) * counts[:, None]
            + rng.integers(0, 1 << 20, size=(K, d)).astype(np.uint64))
    old = rng.random((K, d)) * 300.0
    acc = np.concatenate([sums.reshape(-1), counts]).astype(np.int64)
    got = eng.lbg_finalize(dev(acc), dev(old)).cpu().numpy()
    assert np.array_equal(got, oracle.finalize(sums, counts, old))


def test_numpy_order_sum(eng, oracle):
    """vq_numpy_sum_f64 against numpy itself and the oracle: lengths around the 8-, 128- and
    8192-element boundaries of numpy's pairwise reduction, and the reference's own np.mean KATs."""
    rng = np.random.default_rng(8)
    lens = [0, 1, 7, 8, 9, 127, 128, 129, 255, 1000, 4096, 8191, 8192, 8193, 8205, 12769, 16384, 16385, 24575,
            100003, 262144, 1048576 + 77]
    for n in lens:
        a = rng.random(n) * 300.0
        got = eng.numpy_sum(torch.from_numpy(a)).item()
        assert got == float(np.add.reduce(a)) if n else got == 0.0, f"n={n}"
        assert got == oracle.numpy_sum(a), f"n={n}"
    g = load_golden("pairwise_mean")
    pos = 0
    for n, m in zip(g["lens"], g["means"]):
        a = g["data"][pos:pos + int(n)]
        pos += int(n)
        assert eng.numpy_sum(torch.from_numpy(a.copy())).item() / float(n) == m


# ---------------------------------------------------------------------------------------------------
# window-sorted vectors + pruned search: identical results to the full search
# ---------------------------------------------------------------------------------------------------
def _sorted_parts(buf, N, d):
    """(rows, perm) views of the buffer vq_blocks_sort fills: N*d row bytes, padded to 256, then N uint16."""
    off = (N * d + 255) // 256 * 256
    h = buf.cpu().numpy()
    return h[: N * d].reshape(N, d), h[off: off + 2 * N].view(np.uint16)


@pytest.mark.parametrize("N,d", [(1, 16), (100, 16), (8192, 16), (8193, 16), (30000, 16), (20001, 4), (9000, 27), (17000, 32),
                                 (8500, 48), (3, 48), (70000, 48), (5000, 192), (40000, 64), (600000, 64)])
def test_sort_blocks_is_a_windowed_permutation(eng, N, d):
    rng = np.random.default_rng(N + d)
    base = rng.integers(0, 200, size=(N, 1))
    blocks = np.clip(base + rng.integers(0, 56, size=(N, d)), 0, 255).astype(np.uint8)
    buf = eng.sort_blocks(dev(blocks))
    rows, perm = _sorted_parts(buf, N, d)
    shift = 0
    while (255 * d) >> shift >= 2048:
        shift += 1
    for w0 in range(0, N, 8192):
        n = min(8192, N - w0)
        p = perm[w0: w0 + n].astype(np.int64)
        assert np.array_equal(np.sort(p), np.arange(n))                        # a permutation inside the window
        assert np.array_equal(rows[w0: w0 + n], blocks[w0 + p])               # rows moved with their origin
        key = rows[w0: w0 + n].astype(np.int64).sum(1) >> shift
        assert np.all(np.diff(key) >= 0)                                       # ascending element-sum buckets


def _pruned_case(eng, oracle, blocks, cb, want_dist=True):
    want_l, want_d = oracle.encode_l1(blocks, cb, want_dist=True)
    b = dev(blocks)
    srt = eng.sort_blocks(b)
    got_l, got_d = eng.encode(b, dev(cb), want_dist=True, mode=1, sorted=srt)
    assert np.array_equal(got_l.cpu().numpy(), want_l)
    assert np.array_equal(got_d.cpu().numpy(), want_d)


@pytest.mark.parametrize("K,d,N", [(32, 16, 5000), (64, 4, 20000), (255, 16, 9000), (256, 16, 40000), (1000, 16, 8192),
                                   (1024, 16, 12000), (2048, 4, 9000), (100, 27, 7000), (128, 32, 10000), (40, 12, 3000),
                                   (300, 8, 16385), (2048, 16, 9000), (4096, 16, 5000), (512, 48, 8000), (300, 192, 4000),
                                   (4096, 192, 3000), (1000, 96, 2500), (70, 130, 9000)])
def test_encode_pruned_structured(eng, oracle, K, d, N):
    """Vectors and codewords with a wide spread of element sums (where pruning really skips rows)."""
    assert eng.prune_supported(K, d)
    rng = np.random.default_rng(K * 7 + d)
    cb = np.clip(rng.random((K, 1)) * 255.0 + rng.normal(0, 20, (K, d)), 0, 300)
    pick = rng.integers(0, K, N)
    blocks = np.clip(cb[pick] + rng.normal(0, 10, (N, d)), 0, 255).astype(np.uint8)
    _pruned_case(eng, oracle, blocks, cb)


@pytest.mark.parametrize("K,d", [(256, 16), (64, 8)])
def test_encode_pruned_adversarial(eng, oracle, K, d):
    """No spread at all (iid vectors, codewords with equal sums), duplicate and all-zero codewords,
    codewords far outside the pixel range: the pruned search must fall back to the whole codebook."""
    rng = np.random.default_rng(K + d)
    N = 10000
    blocks = rng.integers(0, 256, size=(N, d), dtype=np.uint8)
    cb = rng.random((K, d)) * 255.0
    cb *= (127.5 * d / cb.sum(1))[:, None]                       # equal element sums
    _pruned_case(eng, oracle, blocks, cb)
    cb2 = rng.random((K, d)) * 255.0
    cb2[5] = cb2[3]; cb2[K - 1] = cb2[3]; cb2[7] = 0.0; cb2[8] = 0.0; cb2[9] = 257.55; cb2[10] = 1000.0
    blocks[:500] = 0
    blocks[500:900] = 255
    _pruned_case(eng, oracle, blocks, cb2)


def test_encode_pruned_golden(eng, oracle, train_case):
    g = train_case
    K, d = g["codebook"].shape
    if not eng.prune_supported(K, d):
        pytest.skip("codebook shape is not served by the pruned search")
    b = dev(g["blocks"])
    got = eng.encode(b, dev(g["codebook"]), mode=1, sorted=eng.sort_blocks(b)).cpu().numpy()
    assert np.array_equal(got.reshape(g["labels"].shape), g["labels"])


@pytest.mark.parametrize("K,d", [(256, 16), (64, 4), (1024, 16), (2048, 16), (128, 15), (64, 11), (32, 27), (256, 3),
                                 (256, 48), (600, 48), (512, 192), (4096, 8)])
def test_lbg_iteration_pruned(eng, oracle, K, d):
    rng = np.random.default_rng(K * 3 + d)
    N = 20000
    cb = np.clip(rng.random((K, 1)) * 255.0 + rng.normal(0, 25, (K, d)), 0, 255)
    blocks = np.clip(cb[rng.integers(0, K, N)] + rng.normal(0, 12, (N, d)), 0, 255).astype(np.uint8)
    blocks[:3000] = blocks[0]
    want = oracle.lbg_iter(blocks, cb)
    b = dev(blocks)
    acc, dist, labels = eng.lbg_accum(b, dev(cb), want_labels=True, mode=1, sorted=eng.sort_blocks(b))
    acc_h = acc.cpu().numpy().astype(np.uint64)
    assert np.array_equal(labels.cpu().numpy(), want["labels"])
    assert np.array_equal(acc_h[: K * d].reshape(K, d), want["sums"])
    assert np.array_equal(acc_h[K * d:], want["counts"])
    assert dist.item() == oracle.numpy_sum(want["min_dist"])


@pytest.mark.parametrize("K,d,N", [(64, 16, 54912), (100, 24, 54912), (64, 16, 61440), (32, 12, 122880), (64, 16, 486000),
                                   (256, 16, 491520)])
def test_lbg_iteration_pruned_tile_counts(eng, oracle, K, d, N):
    """Launches whose workgroup tiles do not divide the sort window (the fused pass takes 1536 vectors per workgroup,
    five and a third per window) at the tile counts where the kernel re-orders the tiles of an XCD's share: every
    tile must still be visited exactly once."""
    rng = np.random.default_rng(K + d + N)
    cb = np.clip(rng.random((K, 1)) * 255.0 + rng.normal(0, 25, (K, d)), 0, 255)
    blocks = np.clip(cb[rng.integers(0, K, N)] + rng.normal(0, 12, (N, d)), 0, 255).astype(np.uint8)
    want = oracle.lbg_iter(blocks, cb)
    b = dev(blocks)
    acc, dist, labels = eng.lbg_accum(b, dev(cb), want_labels=True, mode=1, sorted=eng.sort_blocks(b))
    acc_h = acc.cpu().numpy().astype(np.uint64)
    assert np.array_equal(labels.cpu().numpy(), want["labels"])
    assert np.array_equal(acc_h[: K * d].reshape(K, d), want["sums"])
    assert np.array_equal(acc_h[K * d:], want["counts"])
    assert dist.item() == oracle.numpy_sum(want["min_dist"])


# ---------------------------------------------------------------------------------------------------
# full training, through the reference-shaped Codebook class
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_codebook_class_matches_reference(tmp_path, train_case, mode, capsys):
    from vector_quantization_amd import Codebook
    g = train_case
    if mode == "exact" and g["name"] not in ("luffy_2x2_k16", "man_2x2_k5", "synth_iid_128_4x4_k32"):
        pytest.skip("exact mode is checked on the small cases")
    cb = Codebook.from_array(g["img"], int(g["bh"]), int(g["bw"]), base_name=g["name"], device="cuda:0", mode=mode,
                             output_dir=str(tmp_path))
    assert np.array_equal(cb.blocks, g["blocks"])
    final = cb.generate_codebook(int(g["k"]))
    out = capsys.readouterr().out
    assert cb.training.iters_per_stage == list(g["iters_per_stage"])
    assert cb.training.converged_iter == list(g["converged_iter"])
    assert np.array_equal(np.array(cb.training.distortions), g["distortions"])   # bit-exact distortions
    assert np.array_equal(cb.codebook, g["codebook"])                      # bit-exact codebook
    assert np.array_equal(np.array(final).reshape(g["codebook"].shape), g["codebook"])
    ref_conv = [l for l in str(g["log"]).splitlines() if l.startswith("Converged")]
    got_conv = [l for l in out.splitlines() if l.startswith("Converged")]
    assert got_conv == ref_conv                                             # same iteration, same 3-dp distortion
    labels = cb.compress()
    assert labels.dtype == np.int64 and np.array_equal(labels, g["labels"])
    assert np.array_equal(np.fromfile(cb.labels_bin, dtype=np.uint8), g["labels_bin"])
    import hashlib
    assert hashlib.sha256(open(cb.codebook_json, "rb").read()).hexdigest() == str(g["codebook_json_sha256"])
    assert hashlib.sha256(open(cb.labels_json, "rb").read()).hexdigest() == str(g["labels_json_sha256"])
    assert hashlib.sha256(open(cb.codebook_txt, "rb").read()).hexdigest() == str(g["codebook_txt_sha256"])
    arr = Codebook.decompress(cb.labels_json, cb.codebook_json, cb.reconstructed_path, device="cuda:0")
    assert arr.dtype == np.uint8 and np.array_equal(arr, g["decoded"])


def test_gray_codebook_class_matches_old_reference(tmp_path, capsys):
    """GrayCodebook on the GPU against old/CodeBook_VQ.py run unmodified (grayscale, zero padding)."""
    from vector_quantization_amd import GrayCodebook
    from vqtest_util import load_golden
    g = load_golden("gray_man_4x4_k16")
    cb = GrayCodebook("unused", 4, 4, device="cuda:0", image=g["gray"])
    assert np.array_equal(cb.blocks, g["blocks"])
    jp = tmp_path / "cb.json"
    cb.generate_codebook(16, json_path=str(jp))
    assert capsys.readouterr().out == str(g["log"])
    assert np.array_equal(cb.codebook, g["codebook"])
    assert open(jp).read() == str(g["json_text"])


def test_decompress_crop(tmp_path):
    from vector_quantization_amd import Codebook
    from vqtest_util import load_golden
    g = load_golden("synth_gray_250x246_4x4_k64")
    cb = Codebook.from_array(g["img"], 4, 4, base_name="s", device="cuda:0", output_dir=str(tmp_path))
    cb.generate_codebook(int(g["k"]))
    cb.compress()
    crop = Codebook.decompress(cb.labels_json, cb.codebook_json, cb.reconstructed_path, device="cuda:0",
                               crop=(cb.orig_h, cb.orig_w))
    assert np.array_equal(crop, g["decoded"][:250, :246])


# ---------------------------------------------------------------------------------------------------
# decode, bit stream
# ---------------------------------------------------------------------------------------------------
def test_decode_golden(eng, train_case):
    g = train_case
    C = g["img"].shape[2]
    img = eng.decode(dev(g["labels"], torch.int32), dev(g["codebook"]), int(g["bh"]), int(g["bw"]), C)
    assert np.array_equal(img.cpu().numpy(), g["decoded"])


@pytest.mark.parametrize("K,bh,bw,C,nr,nc", [(16, 2, 2, 1, 5, 150), (16, 2, 2, 1, 3, 4500), (7, 2, 2, 3, 4, 100), (5, 3, 3, 1, 3, 70),
                                             (300, 1, 1, 3, 2, 100), (9000, 3, 3, 1, 2, 80), (16, 2, 2, 1, 1025, 1026),
                                             (256, 4, 4, 3, 9, 1023), (64, 8, 8, 1, 5, 65), (40, 5, 3, 3, 7, 333)])
@pytest.mark.parametrize("offset", [0, 1, 2])
def test_decode_shapes_and_alignments(eng, oracle, K, bh, bw, C, nr, nc, offset):
    """Every decode kernel (quads, word-wide block rows, strips through LDS, byte-wise) on images that start at any byte
    alignment, with more blocks per row than one strip holds and codebooks that do and do not fit the LDS."""
    rng = np.random.default_rng(K * 31 + nr + nc)
    cb = rng.random((K, bh * bw * C)) * 300.0 - 20.0
    labels = rng.integers(0, K, size=(nr, nc)).astype(np.int32)
    want = oracle.decode_u8(labels, cb, bh, bw, C)
    shape = (nr * bh, nc * bw, C)
    numel = shape[0] * shape[1] * shape[2]
    buf = torch.full((numel + 64,), 0xAB, dtype=torch.uint8, device="cuda:0")
    out = buf[16 + offset: 16 + offset + numel].view(shape)
    got = eng.decode(dev(labels), dev(cb), bh, bw, C, out=out)
    assert np.array_equal(got.cpu().numpy(), want)
    guard = buf.cpu().numpy()
    assert (guard[:16 + offset] == 0xAB).all() and (guard[16 + offset + numel:] == 0xAB).all()     # nothing written outside


def test_decode_known_answer(eng):
    g = load_golden("decode_kat_2x2")
    img = eng.decode(dev(g["labels"], torch.int32), dev(g["codebook"].reshape(4, 4)), 2, 2, 1)
    assert np.array_equal(img.cpu().numpy()[:, :, 0], g["img"])


def test_decode_cast_table(eng):
    g = load_golden("cast_f64_u8")
    vals = g["vals"]
    labels = np.arange(len(vals), dtype=np.int32).reshape(1, -1)
    img = eng.decode(dev(labels), dev(vals.reshape(-1, 1)), 1, 1, 1)
    assert np.array_equal(img.cpu().numpy().reshape(-1), g["cast"])


@pytest.mark.parametrize("K", [1, 2, 3, 8, 17, 256, 1000, 4096])
def test_pack_unpack(eng, oracle, K):
    rng = np.random.default_rng(K)
    for N in (0, 1, 7, 8, 9, 1001, 65537):
        labels = rng.integers(0, K, size=N).astype(np.int32)
        packed = eng.pack_labels(dev(labels), K)
        assert np.array_equal(packed.cpu().numpy(), oracle.pack_labels(labels, K))
        back = eng.unpack_labels(packed, N, K)
        assert np.array_equal(back.cpu().numpy(), labels)


# ---------------------------------------------------------------------------------------------------
# full-size properties (no oracle at this size: size-independent checks)
# ---------------------------------------------------------------------------------------------------
def _structured(h, w, seed):
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    y = torch.arange(h, device="cuda:0", dtype=torch.float32)[:, None]
    x = torch.arange(w, device="cuda:0", dtype=torch.float32)[None, :]
    v = (128 + 64 * torch.sin(2 * np.pi * x / 257) * torch.cos(2 * np.pi * y / 181)
         + 32 * torch.sin(2 * np.pi * (x + y) / 1021) + 16 * torch.randn((h, w), device="cuda:0", generator=g))
    return v.clamp(0, 255).to(torch.uint8)


def test_full_size_encode_decode_properties(eng, oracle):
    """8192 x 8192 grayscale, 4x4 blocks, K=256 (the headline configuration)."""
    from vector_quantization_amd import train_lbg
    img = _structured(8192, 8192, 0)
    blocks = eng.blocks_from_image(img, 4, 4)
    assert blocks.shape == (2048 * 2048, 16)
    crop = eng.blocks_from_image(img[:1024, :1024].contiguous(), 4, 4)
    cb = train_lbg(eng, crop, 256, log=None).codebook
    lab_fast = eng.encode(blocks, cb, mode=1)
    lab_exact = eng.encode(blocks, cb, mode=0)
    assert torch.equal(lab_fast, lab_exact)                       # filter + re-check == all-fp64 search
    srt = eng.sort_blocks(blocks)
    lab_pruned, mind_pruned = eng.encode(blocks, cb, mode=1, sorted=srt, want_dist=True)
    assert torch.equal(lab_pruned, lab_exact)                     # pruned search == all-fp64 search
    acc_p, dist_p = eng.lbg_accum(blocks, cb, sorted=srt)
    # a sample against the CPU oracle
    idx = torch.randperm(blocks.shape[0], device="cuda:0")[:20000]
    want = oracle.encode_l1(blocks[idx].cpu().numpy(), cb.cpu().numpy())
    assert np.array_equal(lab_fast[idx].cpu().numpy(), want)
    # decode, then encode the decoded image again: labels are a fixed point wherever the codeword is
    # integral after truncation; in general the re-encoded image decodes to the same pixels
    dec = eng.decode(lab_fast.reshape(2048, 2048), cb, 4, 4, 1)
    assert dec.shape == (8192, 8192, 1)
    blocks2 = eng.blocks_from_image(dec, 4, 4)
    cb8 = torch.from_numpy(oracle.decode_u8(np.arange(256, dtype=np.int32).reshape(1, -1), cb.cpu().numpy(), 4, 4, 1)
                           ).to("cuda:0")                           # the truncated codebook as a 4 x 1024 strip
    rows = blocks2.reshape(2048, 2048, 16)
    strip = eng.blocks_from_image(cb8, 4, 4)                        # (256, 16) truncated codewords
    assert torch.equal(rows.reshape(-1, 16), strip[lab_fast.long()])
    # bit stream round trip at full size
    packed = eng.pack_labels(lab_fast, 256)
    assert packed.numel() == blocks.shape[0]
    assert torch.equal(eng.unpack_labels(packed, blocks.shape[0], 256), lab_fast)
    # accumulation is a partition of the data: counts add up, sums add up to the column sums
    acc, dist = eng.lbg_accum(blocks, cb)
    assert int(acc[256 * 16:].sum().item()) == blocks.shape[0]
    assert torch.equal(acc[: 256 * 16].reshape(256, 16).sum(0), eng.colsum(blocks))
    # run-to-run determinism, including the fp64 distance sum; the distance sum is numpy's
    acc2, dist2, lab2 = eng.lbg_accum(blocks, cb, want_labels=True)
    assert torch.equal(acc, acc2) and dist.item() == dist2.item() and torch.equal(lab2, lab_fast)
    _, mind = eng.encode(blocks, cb, want_dist=True)
    assert dist.item() == float(np.add.reduce(mind.cpu().numpy()))
    assert torch.equal(mind, mind_pruned) and torch.equal(acc, acc_p) and dist.item() == dist_p.item()


# ---------------------------------------------------------------------------------------------------
# misuse of the window-sorted copy is reported, not followed
# ---------------------------------------------------------------------------------------------------
def test_sorted_buffer_mismatch_is_refused(eng):
    rng = np.random.default_rng(77)
    K, d = 64, 16
    cb = dev(rng.random((K, d)) * 255.0)
    a = dev(rng.integers(0, 256, size=(20000, d), dtype=np.uint8))
    b = dev(rng.integers(0, 256, size=(9000, d), dtype=np.uint8))
    srt_a = eng.sort_blocks(a)
    good = eng.encode(a, cb, mode=1, sorted=srt_a)
    assert torch.equal(good, eng.encode(a, cb, mode=1))
    # a buffer sorted for another vector count: refused on the host
    with pytest.raises(ValueError, match="sorted buffer"):
        eng.encode(b, cb, mode=1, sorted=srt_a)
    # a buffer vq_blocks_sort never filled
    fresh = torch.zeros_like(srt_a)
    with pytest.raises(ValueError, match="not filled by vq_blocks_sort"):
        eng.encode(a, cb, mode=1, sorted=fresh)
    # contents replaced after the sort (the header no longer says N, d): every label comes back -1 and the
    # stale origin map is not followed
    srt_a.zero_()
    bad = eng.encode(a, cb, mode=1, sorted=srt_a)
    assert bool((bad == -1).all())
    # sorting again makes the buffer valid again
    eng.sort_blocks(a, out=srt_a)
    assert torch.equal(eng.encode(a, cb, mode=1, sorted=srt_a), good)


def test_decompress_rejects_out_of_range_labels(tmp_path):
    """codebook[labels[r][c]] in the reference: negative labels count from the end, anything else outside
    [-K, K) raises IndexError (VQ_Full.py:162)."""
    import json
    from vector_quantization_amd import Codebook
    cb = (np.arange(4 * 2 * 2 * 3, dtype=np.float64).reshape(4, 2, 2, 3) * 3.0).tolist()
    cbp, lp, op = tmp_path / "cb.json", tmp_path / "l.json", tmp_path / "o.png"
    json.dump(cb, open(cbp, "w"))
    json.dump([[0, -1, 3], [-4, 2, 1]], open(lp, "w"))
    arr = Codebook.decompress(str(lp), str(cbp), str(op), device="cuda:0")
    json.dump([[0, 3, 3], [0, 2, 1]], open(lp, "w"))
    assert np.array_equal(arr, Codebook.decompress(str(lp), str(cbp), str(op), device="cuda:0"))
    for bad in (4, -5, 1000000):
        json.dump([[0, bad, 3], [0, 2, 1]], open(lp, "w"))
        with pytest.raises(IndexError, match=f"index {bad} is out of bounds for axis 0 with size 4"):
            Codebook.decompress(str(lp), str(cbp), str(op), device="cuda:0")


# ---------------------------------------------------------------------------------------------------
# RGB 8x8 blocks (d = 192), K = 4096: the reference's own run on team.jpg (N = 19 200)
# ---------------------------------------------------------------------------------------------------
def test_rgb_8x8_k4096_training_and_encode_match_reference(tmp_path, capsys):
    import hashlib
    from vector_quantization_amd import Codebook
    g = load_golden("team_8x8_k4096")
    cb = Codebook.from_array(g["img"], 8, 8, base_name="team", device="cuda:0", output_dir=str(tmp_path))
    assert cb.blocks.shape == (19200, 192)
    cb.generate_codebook(4096)
    out = capsys.readouterr().out
    assert cb.training.iters_per_stage == list(g["iters_per_stage"])
    assert cb.training.converged_iter == list(g["converged_iter"])
    assert np.array_equal(np.array(cb.training.distortions), g["distortions"])
    assert np.array_equal(cb.codebook, g["codebook"])
    ref_conv = [l for l in str(g["log"]).splitlines() if l.startswith("Converged")]
    assert [l for l in out.splitlines() if l.startswith("Converged")] == ref_conv
    labels = cb.compress()
    assert np.array_equal(labels, g["labels"])
    assert np.array_equal(np.fromfile(cb.labels_bin, dtype=np.uint8), g["labels_bin"])
    assert hashlib.sha256(open(cb.codebook_json, "rb").read()).hexdigest() == str(g["codebook_json_sha256"])
    assert hashlib.sha256(open(cb.labels_json, "rb").read()).hexdigest() == str(g["labels_json_sha256"])
    arr = Codebook.decompress(cb.labels_json, cb.codebook_json, cb.reconstructed_path, device="cuda:0")
    assert hashlib.sha256(arr.tobytes()).hexdigest() == str(g["decoded_sha256"])
    # the last iteration's per-vector distances (the values VQ_Full.py:89 gathers), exact-mode cross-check
    eng2 = cb._engine
    lab_x, mind_x = eng2.encode(cb._blocks_dev, dev(g["codebook"]), want_dist=True, mode=0)
    assert np.array_equal(lab_x.cpu().numpy().reshape(labels.shape), g["labels"])


# ---------------------------------------------------------------------------------------------------
# image-fused entry points: image -> sorted rows in one pass, labels straight from the image
# ---------------------------------------------------------------------------------------------------
def test_sort_image_golden(eng, train_case):
    g = train_case
    N, d = g["blocks"].shape
    if d > 192:
        pytest.skip("beyond the fast kernels")
    buf, nd = eng.sort_image(dev(g["img"]), int(g["bh"]), int(g["bw"]))
    assert nd == (N, d)
    rows, perm = _sorted_parts(buf, N, d)
    for w0 in range(0, N, 8192):
        n = min(8192, N - w0)
        p = perm[w0: w0 + n].astype(np.int64)
        assert np.array_equal(np.sort(p), np.arange(n))
        assert np.array_equal(rows[w0: w0 + n], g["blocks"][w0 + p])


@pytest.mark.parametrize("shape,bh,bw", [((7, 5, 3), 2, 2), ((33, 17, 1), 4, 4), ((64, 64, 3), 8, 8), ((9, 9, 1), 3, 3),
                                         ((130, 67, 3), 3, 5), ((16, 20, 2), 4, 2), ((515, 770, 1), 4, 4), ((300, 301, 1), 2, 2),
                                         ((398, 402, 3), 4, 4), ((37, 44, 3), 4, 4), ((3, 2, 3), 4, 4), ((1030, 516, 1), 4, 8),
                                         ((250, 260, 3), 8, 8),
                                         # rows that are not whole words: extraction + in-place sort (d <= 16, 32, 48) ...
                                         ((200, 203, 2), 4, 4), ((333, 1001, 1), 2, 2), ((129, 255, 3), 2, 2), ((600, 601, 3), 4, 4),
                                         # ... and the fused kernels on unaligned words (other row lengths)
                                         ((100, 101, 3), 8, 8), ((64, 66, 1), 8, 8), ((90, 91, 3), 4, 2)])
@pytest.mark.parametrize("pad_mode", [0, 1])
def test_sort_image_ragged(eng, oracle, shape, bh, bw, pad_mode):
    rng = np.random.default_rng(hash((shape, bh, bw)) & 0xffff)
    img = rng.integers(0, 256, size=shape, dtype=np.uint8)
    blocks = oracle.blocks_u8(img, bh, bw, pad_mode)
    N, d = blocks.shape
    buf, nd = eng.sort_image(dev(img), bh, bw, pad_mode)
    assert nd == (N, d)
    rows, perm = _sorted_parts(buf, N, d)
    for w0 in range(0, N, 8192):
        n = min(8192, N - w0)
        p = perm[w0: w0 + n].astype(np.int64)
        assert np.array_equal(np.sort(p), np.arange(n))
        assert np.array_equal(rows[w0: w0 + n], blocks[w0 + p])


@pytest.mark.parametrize("shape,bh,bw", [((256, 256, 3), 4, 4), ((128, 260, 1), 4, 4), ((64, 64, 3), 8, 8)])
@pytest.mark.parametrize("offset", [1, 2, 3])
def test_image_at_an_odd_address(eng, oracle, shape, bh, bw, offset):
    """An image that does not start on a word boundary (a view into a larger byte buffer): block extraction, the
    image sort and labels straight from the image equal the oracle's."""
    rng = np.random.default_rng(offset * 100 + shape[1])
    img = rng.integers(0, 256, size=shape, dtype=np.uint8)
    n = img.size
    buf = torch.zeros(n + 8, dtype=torch.uint8, device="cuda:0")
    view = buf[offset: offset + n].view(*shape)
    view.copy_(dev(img))
    assert view.data_ptr() % 4 == offset
    blocks = oracle.blocks_u8(img, bh, bw, 0)
    N, d = blocks.shape
    assert np.array_equal(eng.blocks_from_image(view, bh, bw).cpu().numpy(), blocks)
    srt, nd = eng.sort_image(view, bh, bw)
    rows, perm = _sorted_parts(srt, N, d)
    for w0 in range(0, N, 8192):
        m = min(8192, N - w0)
        p = perm[w0: w0 + m].astype(np.int64)
        assert np.array_equal(np.sort(p), np.arange(m))
        assert np.array_equal(rows[w0: w0 + m], blocks[w0 + p])
    cb = rng.random((64, d)) * 255.0
    assert np.array_equal(eng.encode_image(view, bh, bw, dev(cb)).cpu().numpy(), oracle.encode_l1(blocks, cb))


def test_encode_image_golden(eng, train_case):
    """Labels straight from the image equal the reference's labels for the image's block vectors; the
    per-vector distances equal the reference's where the fixture holds them."""
    g = train_case
    if g["blocks"].shape[1] > 192:
        pytest.skip("beyond the fast kernels")
    labels = eng.encode_image(dev(g["img"]), int(g["bh"]), int(g["bw"]), dev(g["codebook"]))
    assert np.array_equal(labels.cpu().numpy().astype(np.int64).reshape(g["labels"].shape), g["labels"])
    if "it_cb_in" in g:
        labels, mind = eng.encode_image(dev(g["img"]), int(g["bh"]), int(g["bw"]), dev(g["it_cb_in"]), want_dist=True)
        assert np.array_equal(labels.cpu().numpy().astype(np.int64), g["it_labels"])
        assert np.array_equal(mind.cpu().numpy(), g["it_mind"])


def test_lbg_iteration_from_sorted_image_only(eng, train_case):
    """One LBG pass with the block matrix never built (sorted rows from the image, blocks = None)."""
    g = train_case
    if "it_cb_in" not in g:
        pytest.skip("no iteration KAT")
    K, d = g["it_cb_in"].shape
    if not eng.prune_supported(K, d):
        pytest.skip("shape not served by the pruned search")
    srt, nd = eng.sort_image(dev(g["img"]), int(g["bh"]), int(g["bw"]))
    acc, dist, labels = eng.lbg_accum(None, dev(g["it_cb_in"]), want_labels=True, sorted=srt, shape=nd)
    assert np.array_equal(labels.cpu().numpy().astype(np.int64), g["it_labels"])
    assert dist.item() / nd[0] == float(g["it_distortion"])
    assert np.array_equal(eng.lbg_finalize(acc, dev(g["it_cb_in"])).cpu().numpy(), g["it_cb_out"])


def test_full_size_image_fused_encode(eng):
    """8192 x 8192: labels from the image in one call equal labels from extract + sort + search."""
    from vector_quantization_amd import train_lbg
    img = _structured(8192, 8192, 3)
    blocks = eng.blocks_from_image(img, 4, 4)
    cb = train_lbg(eng, eng.blocks_from_image(img[:1024, :1024].contiguous(), 4, 4), 256, log=None).codebook
    want, want_d = eng.encode(blocks, cb, mode=1, sorted=eng.sort_blocks(blocks), want_dist=True)
    got, got_d = eng.encode_image(img, 4, 4, cb, want_dist=True)
    assert torch.equal(got, want) and torch.equal(got_d, want_d)
    assert torch.equal(got, eng.encode(blocks, cb, mode=0))


# ---------------------------------------------------------------------------------------------------
# 16384 x 16384 grayscale, 4x4 blocks, LBG to K = 1024: the full training workload at its real size
# ---------------------------------------------------------------------------------------------------
def test_full_size_k1024_training_properties(eng, oracle):
    """N = 16 777 216 vectors is beyond what the CPU oracle trains in seconds, so the run is checked through
    what does not depend on size: the device-side stage loop (stop rule of VQ_Full.py:89-98 on the device,
    fused search + accumulation while the table fits) against a host-driven replay of VQ_Full.py:70-98 built
    from the separate entry points with the unpruned search, the partition of the data by the last
    assignment, numpy's distance sum, the all-fp64 search on a quarter of the vectors and the CPU oracle
    on a sample."""
    from vector_quantization_amd import train_lbg, lbg_iteration
    from vector_quantization_amd.lbg import initial_centroid
    img = _structured(16384, 16384, 5)
    blocks = eng.blocks_from_image(img, 4, 4)
    del img
    N = blocks.shape[0]
    assert blocks.shape == (4096 * 4096, 16)
    res = train_lbg(eng, blocks, 1024, log=None)
    assert res.codebook.shape == (1024, 16) and len(res.iters_per_stage) == 10
    assert len(res.distortions) == sum(res.iters_per_stage)
    # host-driven replay: split, assign without the sorted copy, centroid update, stop rule in Python
    cb, n_total = initial_centroid(eng, blocks)
    assert n_total == N
    iters, dists = [], []
    while cb.shape[0] < 1024:
        cb = eng.lbg_split(cb, 0.01)
        prev, done = float("inf"), 0
        for _ in range(100):
            cb, distortion = lbg_iteration(eng, blocks, cb, n_total)
            dists.append(distortion)
            done += 1
            if prev != float("inf") and prev != 0.0 and abs(prev - distortion) / prev < 0.001:
                break
            prev = distortion
        iters.append(done)
    assert iters == list(res.iters_per_stage)
    assert dists == list(res.distortions)
    assert torch.equal(cb, res.codebook)
    # the final codebook: pruned search == unpruned search everywhere, == all-fp64 search on a quarter
    srt = eng.sort_blocks(blocks)
    lab_p, mind_p = eng.encode(blocks, cb, mode=1, sorted=srt, want_dist=True)
    lab_f, mind_f = eng.encode(blocks, cb, mode=1, want_dist=True)
    assert torch.equal(lab_p, lab_f) and torch.equal(mind_p, mind_f)
    q = N // 4
    lab_x, mind_x = eng.encode(blocks[2 * q: 3 * q], cb, mode=0, want_dist=True)
    assert torch.equal(lab_x, lab_p[2 * q: 3 * q]) and torch.equal(mind_x, mind_p[2 * q: 3 * q])
    idx = torch.randperm(N, device="cuda:0")[:20000]
    assert np.array_equal(lab_p[idx].cpu().numpy(), oracle.encode_l1(blocks[idx].cpu().numpy(), cb.cpu().numpy()))
    # one more pass: a partition of the data, numpy's distance sum, the same with and without the sorted copy
    acc, dist, lab_a = eng.lbg_accum(blocks, cb, want_labels=True, sorted=srt)
    acc2, dist2 = eng.lbg_accum(blocks, cb)
    assert torch.equal(acc, acc2) and dist.item() == dist2.item() and torch.equal(lab_a, lab_p)
    assert int(acc[1024 * 16:].sum().item()) == N
    assert torch.equal(acc[: 1024 * 16].reshape(1024, 16).sum(0), eng.colsum(blocks))
    assert torch.equal(acc[1024 * 16:], torch.bincount(lab_p.long(), minlength=1024))
    assert dist.item() == float(np.add.reduce(mind_p.cpu().numpy()))


# ---------------------------------------------------------------------------------------------------
# a batch of 512 images of 2048 x 2048 (one GPU's share of 4096 images over 8 GPUs): N = 2^27 vectors
# ---------------------------------------------------------------------------------------------------
def test_batch_share_of_one_gpu_encode_decode(eng, oracle):
    """The images of a batch are independent and 4 divides their height, so the batch is one tall image:
    2 GiB of pixels, 2^31 bytes of block vectors -- every byte offset beyond 32 bits is exercised.  Labels of
    the whole batch in one call equal the labels of single images encoded on their own (first, middle, last),
    the all-fp64 search on the last image, and the CPU oracle on a sample drawn from the whole batch."""
    from vector_quantization_amd import train_lbg
    n_img, side = 512, 2048
    per = (side // 4) ** 2
    batch = torch.empty((n_img, side, side), dtype=torch.uint8, device="cuda:0")
    for i in range(0, n_img, 64):
        batch[i: i + 64] = _structured(64 * side, side, 100 + i).view(64, side, side)
    cb = train_lbg(eng, eng.blocks_from_image(batch[7], 4, 4), 256, log=None).codebook
    tall = batch.view(n_img * side, side)
    labels = eng.encode_image(tall, 4, 4, cb)
    assert labels.numel() == n_img * per
    labels = labels.view(n_img, per)
    for i in (0, 255, n_img - 1):
        assert torch.equal(eng.encode_image(batch[i], 4, 4, cb).reshape(-1), labels[i])
    last = eng.blocks_from_image(batch[n_img - 1], 4, 4)
    assert torch.equal(eng.encode(last, cb, mode=0).reshape(-1), labels[n_img - 1])
    # block extraction over the whole batch, then a sample against the CPU oracle
    blocks = eng.blocks_from_image(tall, 4, 4)
    assert blocks.shape == (n_img * per, 16)
    assert torch.equal(blocks[(n_img - 1) * per:], last)
    idx = torch.randint(0, n_img * per, (20000,), device="cuda:0")
    idx[:4] = torch.tensor([0, 1, n_img * per - 2, n_img * per - 1], device="cuda:0")
    want = oracle.encode_l1(blocks[idx].cpu().numpy(), cb.cpu().numpy())
    assert np.array_equal(labels.reshape(-1)[idx].cpu().numpy(), want)
    # search on the block matrix (with and without its sorted copy) == search from the image
    flat = labels.reshape(-1)
    assert torch.equal(eng.encode(blocks, cb, mode=1, sorted=eng.sort_blocks(blocks)).reshape(-1), flat)
    assert torch.equal(eng.encode(blocks, cb, mode=1).reshape(-1), flat)
    del blocks
    # decode of the whole batch: every image equals its own decode; bit stream round trip of 2^27 labels
    dec = eng.decode(flat.view(n_img * (side // 4), side // 4), cb, 4, 4, 1).view(n_img, side, side)
    for i in (0, 300, n_img - 1):
        assert torch.equal(dec[i], eng.decode(labels[i].view(side // 4, side // 4), cb, 4, 4, 1).view(side, side))
    packed = eng.pack_labels(flat, 256)
    assert torch.equal(eng.unpack_labels(packed, flat.numel(), 256), flat)


def test_batch_share_of_one_gpu_lbg_pass(eng, oracle):
    """One GPU's share of the 4096-image batch (512 images of 2048^2, N = 2^27 vectors) through one LBG pass: the
    per-column sums pass 2^32 (255 * N = 3.4e10), so the 64-bit accumulators are exercised for real.  The fused pass
    (K = 256) and the search + separate accumulation pass (K = 1024) are both checked: counts add up to N, the column
    totals of the cluster sums equal vq_colsum_u8, the pruned route equals the unpruned one, and sums / counts of a
    2^20-vector slice run as its own call equal the CPU oracle's."""
    from vector_quantization_amd import train_lbg
    n_img, side = 512, 2048
    batch = torch.empty((n_img, side, side), dtype=torch.uint8, device="cuda:0")
    for i in range(0, n_img, 64):
        batch[i: i + 64] = _structured(64 * side, side, 100 + i).view(64, side, side)
    tall = batch.view(n_img * side, side)
    blocks = eng.blocks_from_image(tall, 4, 4)
    del batch, tall
    N, d = blocks.shape
    assert N == 1 << 27
    col = eng.colsum(blocks)
    assert int(col.max().item()) > 1 << 32
    srt = eng.sort_blocks(blocks)
    for K in (256, 1024):
        cb = train_lbg(eng, blocks[: 1 << 18].contiguous(), K, log=None).codebook
        acc, dist = eng.lbg_accum(None, cb, sorted=srt, shape=(N, d))             # pruned (fused at K = 256)
        sums, counts = acc[: K * d].view(K, d), acc[K * d:]
        assert int(counts.sum().item()) == N
        assert torch.equal(sums.sum(dim=0), col)
        acc_u, dist_u = eng.lbg_accum(blocks, cb)                                  # unpruned search + accumulation pass
        assert torch.equal(acc_u, acc)
        assert float(dist_u.item()) == float(dist.item())
        sl = blocks[5 << 20: 6 << 20].contiguous()
        acc_s, dist_s = eng.lbg_accum(None, cb, sorted=eng.sort_blocks(sl), shape=(1 << 20, d))
        ref = oracle.lbg_iter(sl.cpu().numpy(), cb.cpu().numpy())
        assert np.array_equal(acc_s[: K * d].cpu().numpy().astype(np.uint64).reshape(K, d), ref["sums"])
        assert np.array_equal(acc_s[K * d:].cpu().numpy().astype(np.uint64), ref["counts"])
        assert float(dist_s.item()) / (1 << 20) == ref["distortion"]
