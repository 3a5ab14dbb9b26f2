"""The CPU oracle against outputs of the reference implementation (tests/golden/*.npz).

These tests pin the oracle: every function of oracle/vq_oracle.c must reproduce, bit for bit,
what VQ_Full.py produced on the same inputs in the build container.
"""
import numpy as np
import pytest

from vqtest_util import load_golden, stage_end_codebooks


def test_blocks_match_reference(oracle, train_case):
    g = train_case
    blocks = oracle.blocks_u8(g["img"], int(g["bh"]), int(g["bw"]))
    assert blocks.shape == g["blocks"].shape
    assert np.array_equal(blocks, g["blocks"])


def test_zero_padding_mode(oracle):
    g = load_golden("luffy_4x4_k32")
    img = g["img"]
    H, W, C = img.shape
    ref = np.pad(img, ((0, (4 - H % 4) % 4), (0, (4 - W % 4) % 4), (0, 0)), mode="constant", constant_values=0)
    nr, nc = ref.shape[0] // 4, ref.shape[1] // 4
    ref_blocks = ref.reshape(nr, 4, nc, 4, C).swapaxes(1, 2).reshape(-1, 4 * 4 * C)
    assert np.array_equal(oracle.blocks_u8(img, 4, 4, pad_mode=1), ref_blocks)


def test_encode_matches_reference(oracle, train_case):
    g = train_case
    labels = oracle.encode_l1(g["blocks"], g["codebook"])
    assert np.array_equal(labels.astype(np.int64).reshape(g["labels"].shape), g["labels"])


def test_single_iteration_matches_reference(oracle, train_case):
    g = train_case
    if "it_cb_in" not in g:
        pytest.skip("no iteration KAT in this fixture")
    r = oracle.lbg_iter(g["blocks"], g["it_cb_in"])
    assert np.array_equal(r["labels"].astype(np.int64), g["it_labels"])
    assert np.array_equal(r["min_dist"], g["it_mind"])          # sequential fp64 sums, bit for bit
    assert np.array_equal(r["cb_out"], g["it_cb_out"])
    assert r["distortion"] == float(g["it_distortion"])          # numpy pairwise mean, bit for bit


def test_full_training_matches_reference(oracle, train_case):
    g = train_case
    r = oracle.lbg_train(g["blocks"], int(g["k"]), float(g["eps"]), float(g["thr"]), int(g["max_iter"]))
    assert np.array_equal(r["iters_per_stage"], g["iters_per_stage"])
    assert np.array_equal(r["converged_iter"], g["converged_iter"])
    assert np.array_equal(r["distortions"], g["distortions"])
    assert r["codebook"].shape == g["codebook"].shape
    assert np.array_equal(r["codebook"], g["codebook"])


def test_split_and_stage_chain(oracle, train_case):
    g = train_case
    ends = stage_end_codebooks(g)
    for s in range(len(ends) - 1):
        nxt = oracle.split(ends[s], 0.01)
        assert np.array_equal(nxt, np.vstack((ends[s] * (1 + 0.01), ends[s] * (1 - 0.01))))


def test_initial_centroid(oracle, train_case):
    g = train_case
    sums = oracle.colsum_u8(g["blocks"])
    N = g["blocks"].shape[0]
    c0 = oracle.finalize(sums[None, :], np.array([N], dtype=np.uint64), np.zeros((1, len(sums))))
    assert np.array_equal(c0[0], np.mean(g["blocks"], axis=0))


def test_decode_matches_reference(oracle, train_case):
    g = train_case
    C = g["img"].shape[2]
    img = oracle.decode_u8(g["labels"], g["codebook"], int(g["bh"]), int(g["bw"]), C)
    assert img.shape == g["decoded"].shape
    assert np.array_equal(img, g["decoded"])


def test_decode_known_answer(oracle):
    g = load_golden("decode_kat_2x2")
    img = oracle.decode_u8(g["labels"], g["codebook"].reshape(4, 4), 2, 2, 1)
    assert np.array_equal(img[:, :, 0], g["img"])


def test_cast_table(oracle):
    g = load_golden("cast_f64_u8")
    vals = g["vals"]
    cb = vals.reshape(-1, 1)
    labels = np.arange(len(vals), dtype=np.int32).reshape(1, -1)
    img = oracle.decode_u8(labels, cb, 1, 1, 1)
    assert np.array_equal(img.reshape(-1), g["cast"])


def test_pack_labels_matches_reference(oracle, train_case):
    g = train_case
    K = len(g["codebook"])
    packed = oracle.pack_labels(g["labels"].reshape(-1), K)
    assert np.array_equal(packed, g["labels_bin"])


def test_pairwise_mean(oracle):
    g = load_golden("pairwise_mean")
    pos = 0
    for n, m in zip(g["lens"], g["means"]):
        a = g["data"][pos:pos + int(n)]
        pos += int(n)
        assert oracle.numpy_sum(a) / float(n) == m


def test_scipy_restatement_agrees_with_c(oracle):
    g = load_golden("synth_gray_250x246_4x4_k64")
    lab_c, d_c = oracle.encode_l1(g["blocks"], g["codebook"], want_dist=True)
    lab_s, d_s = oracle.scipy_encode_chunked(g["blocks"], g["codebook"], chunk=1000, want_dist=True)
    assert np.array_equal(lab_c, lab_s)
    assert np.array_equal(d_c, d_s)
    new_cb, D = oracle.scipy_lbg_iter_chunked(g["blocks"], g["it_cb_in"], chunk=777)
    assert np.array_equal(new_cb, g["it_cb_out"])
    assert D == float(g["it_distortion"])


def test_k_greater_than_n_raises(oracle):
    blocks = np.zeros((3, 4), dtype=np.uint8)
    with pytest.raises(ValueError):
        oracle.lbg_train(blocks, 5)


# ---------------------------------------------------------------------------------------------------
# the oracle against the third-party routines the reference calls (numpy / scipy are importable here)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 127, 128, 129, 1000, 8191, 8192, 8193, 16384, 20000, 65537, 300001])
def test_numpy_sum_matches_numpy_itself(oracle, n):
    """np.mean of VQ_Full.py:90 is np.add.reduce / n: the oracle's summation order against numpy's own, bit for bit."""
    rng = np.random.default_rng(n)
    v = rng.random(n) * 1000.0 + rng.random(n)
    assert oracle.numpy_sum(v) == float(np.add.reduce(v))
    if n:
        assert oracle.numpy_sum(v) / n == float(np.mean(v))


@pytest.mark.parametrize("d,K", [(1, 3), (4, 16), (12, 5), (16, 256), (27, 40), (48, 33), (192, 7)])
def test_encode_matches_scipy_cdist(oracle, d, K):
    """cdist(..., 'cityblock') + np.argmin (VQ_Full.py:77-78, :124-125) evaluated by scipy itself."""
    from scipy.spatial.distance import cdist
    rng = np.random.default_rng(d * 100 + K)
    blocks = rng.integers(0, 256, size=(700, d), dtype=np.uint8)
    cb = rng.random((K, d)) * 255.0
    cb[K // 2] = cb[0]                                     # a duplicated codeword: the lower index must win
    D = cdist(blocks, cb, metric="cityblock")
    labels, dist = oracle.encode_l1(blocks, cb, want_dist=True)
    assert np.array_equal(labels, np.argmin(D, axis=1))
    assert np.array_equal(dist, D[np.arange(len(blocks)), labels])


def test_split_and_mean_match_numpy(oracle):
    rng = np.random.default_rng(3)
    cb = rng.random((37, 12)) * 255.0
    assert np.array_equal(oracle.split(cb, 0.01), np.vstack([cb * (1 + 0.01), cb * (1 - 0.01)]))   # VQ_Full.py:71-73
    blocks = rng.integers(0, 256, size=(5000, 12), dtype=np.uint8)
    labels = rng.integers(0, 37, size=5000).astype(np.int32)
    labels[labels == 5] = 6                                # an empty cluster keeps its old codeword
    sums, counts = oracle.accumulate(blocks, labels, 37)
    new = oracle.finalize(sums, counts, cb)
    for k in range(37):
        m = labels == k
        want = np.mean(blocks[m], axis=0) if m.any() else cb[k]                                      # VQ_Full.py:81-86
        assert np.array_equal(new[k], want)


def test_rgb_8x8_k4096_encode_decode_pack(oracle):
    """d = 192, K = 4096 (team.jpg, N = 19 200): blocks, labels, decoded pixels and the packed bit stream of
    the reference's own run.  The full training of this case takes the oracle minutes: VQ_SLOW_TESTS=1."""
    import hashlib
    g = load_golden("team_8x8_k4096")
    blocks = oracle.blocks_u8(g["img"], 8, 8)
    assert blocks.shape == (19200, 192)
    labels, mind = oracle.encode_l1(blocks, g["codebook"], want_dist=True)
    assert np.array_equal(labels.astype(np.int64).reshape(g["labels"].shape), g["labels"])
    img = oracle.decode_u8(g["labels"], g["codebook"], 8, 8, 3)
    assert hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest() == str(g["decoded_sha256"])
    assert np.array_equal(oracle.pack_labels(g["labels"].reshape(-1), 4096), g["labels_bin"])


@pytest.mark.skipif(not __import__("os").environ.get("VQ_SLOW_TESTS"), reason="minutes of CPU: set VQ_SLOW_TESTS=1")
def test_rgb_8x8_k4096_full_training(oracle):
    g = load_golden("team_8x8_k4096")
    blocks = oracle.blocks_u8(g["img"], 8, 8)
    r = oracle.lbg_train(blocks, 4096, float(g["eps"]), float(g["thr"]), int(g["max_iter"]))
    assert np.array_equal(r["iters_per_stage"], g["iters_per_stage"])
    assert np.array_equal(r["distortions"], g["distortions"])
    assert np.array_equal(r["codebook"], g["codebook"])
