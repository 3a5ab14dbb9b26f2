"""The kernels of vq_hip.hip executed on CPU threads (tests/emu) through the package's own engine,
checked against the oracle on small inputs.  This does not replace the GPU parity tests -- the
emulation models program logic (indexing, barriers, cross-lane exchange, atomics), not the
hardware -- but it runs here, on every change, and under a debugger."""
import os

import numpy as np
import pytest
import torch

from emu_engine import EmuEngine, emulated_kernels, _EMU_DEVICE as GUARDED


@pytest.fixture(scope="module")
def eng():
    with emulated_kernels():
        yield EmuEngine()


def T(a):
    """Host array -> tensor that ends at a protected page (see emu_engine): kernels that read past the
    end of an input are caught, not only those that write past the end of an output."""
    src = torch.from_numpy(np.ascontiguousarray(a))
    t = torch.empty(tuple(src.shape), dtype=src.dtype, device=GUARDED)
    t.copy_(src)
    return t


def _structured(rng, K, d, N, noise=10.0):
    cb = np.clip(rng.random((K, 1)) * 255.0 + rng.normal(0, 20, (K, d)), 0, 300)
    blocks = np.clip(cb[rng.integers(0, K, N)] + rng.normal(0, noise, (N, d)), 0, 255).astype(np.uint8)
    return blocks, cb


@pytest.mark.parametrize("shape,bh,bw,pad", [((33, 17, 1), 4, 4, 0), ((7, 5, 3), 2, 2, 1), ((64, 64, 1), 4, 4, 0),
                                             ((52, 1040, 1), 4, 4, 0), ((1028, 528, 1), 4, 4, 1),
                                             # row strips of up to 256 blocks (segments of whole words, 16-byte vectors):
                                             # two strips per block row, rows that are not whole words, padding on
                                             # both edges; 64-byte and 192-byte vectors
                                             ((13, 1030, 3), 4, 4, 0), ((13, 1030, 3), 4, 4, 1), ((17, 520, 1), 8, 8, 0),
                                             ((10, 515, 3), 8, 8, 1), ((9, 301, 4), 4, 1, 0),
                                             # strips for segments that are not whole words: 2x2 gray and RGB, 3x3, 1x1 RGB,
                                             # odd widths, more blocks per row than one strip holds
                                             ((9, 300, 1), 2, 2, 0), ((11, 301, 1), 2, 2, 1), ((10, 201, 3), 2, 2, 0),
                                             ((10, 211, 1), 3, 3, 1), ((5, 100, 3), 1, 1, 0), ((4, 9001, 1), 2, 2, 0)])
def test_emu_blocks(eng, oracle, shape, bh, bw, pad):
    img = np.random.default_rng(1).integers(0, 256, size=shape, dtype=np.uint8)
    got = eng.blocks_from_image(T(img), bh, bw, pad).numpy()
    assert np.array_equal(got, oracle.blocks_u8(img, bh, bw, pad_mode=pad))


def _check_sorted(buf, blocks):
    N, d = blocks.shape
    off = (N * d + 255) // 256 * 256
    h = buf.numpy()
    rows, perm = h[: N * d].reshape(N, d), h[off: off + 2 * N].view(np.uint16)
    shift = 0
    while (255 * d) >> shift >= 2048:
        shift += 1
    for w0 in range(0, N, 8192):
        n = min(8192, N - w0)
        p = perm[w0: w0 + n].astype(np.int64)
        assert np.array_equal(np.sort(p), np.arange(n))
        assert np.array_equal(rows[w0: w0 + n], blocks[w0 + p])
        assert np.all(np.diff(rows[w0: w0 + n].astype(np.int64).sum(1) >> shift) >= 0)


@pytest.mark.parametrize("N,d", [(700, 16), (8192 + 300, 16), (3000, 4), (2500, 27), (8192 + 2300, 48), (4100, 32), (5, 48),
                                 (2049, 48), (900, 64)])
def test_emu_window_sort(eng, N, d):
    rng = np.random.default_rng(N + d)
    blocks = np.clip(rng.integers(0, 200, size=(N, 1)) + rng.integers(0, 56, size=(N, d)), 0, 255).astype(np.uint8)
    _check_sorted(eng.sort_blocks(T(blocks)), blocks)


@pytest.mark.parametrize("N,d", [(700, 16), (8192 + 300, 16), (3000, 4), (2600, 12), (1900, 8), (6 * 8192 - 100, 8)])
def test_emu_window_sort_lds_variant(eng, N, d):
    """Rows of 1 .. 4 aligned words take the LDS-staged sort kernel: it produces a valid sorted copy."""
    rng = np.random.default_rng(N * 3 + d)
    blocks = np.clip(rng.integers(0, 200, size=(N, 1)) + rng.integers(0, 56, size=(N, d)), 0, 255).astype(np.uint8)
    _check_sorted(eng.sort_blocks(T(blocks)), blocks)


@pytest.mark.parametrize("K,d,N", [(40, 16, 1500), (5, 12, 700), (70, 48, 600), (300, 8, 900), (20, 192, 500),
                                   (4096, 16, 1100), (33, 64, 700), (1, 16, 300), (4100, 8, 500), (4200, 40, 300)])
def test_emu_encode_full_search(eng, oracle, K, d, N):
    rng = np.random.default_rng(K + d)
    blocks, cb = _structured(rng, K, d, N)
    cb[K // 2] = cb[0]                                               # duplicate row
    want_l, want_d = oracle.encode_l1(blocks, cb, want_dist=True)
    got_l, got_d = eng.encode(T(blocks), T(cb), want_dist=True, mode=1)
    assert np.array_equal(got_l.numpy(), want_l)
    assert np.array_equal(got_d.numpy(), want_d)


@pytest.mark.parametrize("K,d,N", [(64, 16, 3000), (32, 4, 2000), (100, 11, 1500), (256, 16, 2048), (2048, 4, 2500),
                                   (500, 32, 1300), (700, 16, 1200), (400, 6, 800)])
def test_emu_encode_pruned(eng, oracle, K, d, N):
    rng = np.random.default_rng(K * 3 + d)
    blocks, cb = _structured(rng, K, d, N)
    cb[K // 3] = cb[1]                                               # duplicate row
    want_l, want_d = oracle.encode_l1(blocks, cb, want_dist=True)
    b = T(blocks)
    got_l, got_d = eng.encode(b, T(cb), want_dist=True, mode=1, sorted=eng.sort_blocks(b))
    assert np.array_equal(got_l.numpy(), want_l)
    assert np.array_equal(got_d.numpy(), want_d)


def test_emu_encode_exact_mode(eng, oracle):
    rng = np.random.default_rng(8)
    blocks, cb = _structured(rng, 20, 16, 600)
    want_l, want_d = oracle.encode_l1(blocks, cb, want_dist=True)
    got_l, got_d = eng.encode(T(blocks), T(cb), want_dist=True, mode=0)
    assert np.array_equal(got_l.numpy(), want_l) and np.array_equal(got_d.numpy(), want_d)


@pytest.mark.parametrize("K,d,N,use_sorted", [(64, 16, 3000, True), (32, 11, 1500, True), (48, 27, 1200, True),
                                              (16, 16, 1500, False), (40, 48, 700, False), (4096, 16, 1200, False),
                                              (2, 16, 900, True)])
def test_emu_lbg_iteration(eng, oracle, K, d, N, use_sorted):
    """Fused accumulation of the pruned search (with padded staging rows) and the separate pass."""
    rng = np.random.default_rng(K * 5 + d)
    blocks, cb = _structured(rng, K, d, N, noise=14.0)
    blocks[:400] = blocks[0]
    want = oracle.lbg_iter(blocks, cb)
    b = T(blocks)
    srt = eng.sort_blocks(b) if use_sorted else None
    acc, dist, labels = eng.lbg_accum(b, T(cb), want_labels=True, mode=1, sorted=srt)
    a = acc.numpy().astype(np.uint64)
    assert np.array_equal(labels.numpy(), want["labels"])
    assert np.array_equal(a[: K * d].reshape(K, d), want["sums"])
    assert np.array_equal(a[K * d:], want["counts"])
    assert dist.item() == oracle.numpy_sum(want["min_dist"])
    assert np.array_equal(eng.lbg_finalize(acc, T(cb)).numpy(), want["cb_out"])


_MANY_CUS_SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, sys.argv[1] + "/..")
from emu_engine import EmuEngine, emulated_kernels
from oracle import vq_oracle
import torch
vq_oracle.build()
K, d, N = 32, 16, 54912
rng = np.random.default_rng(3)
cb = np.clip(rng.random((K, 1)) * 255.0 + rng.normal(0, 20, (K, d)), 0, 300)
blocks = np.clip(cb[rng.integers(0, K, N)] + rng.normal(0, 14.0, (N, d)), 0, 255).astype(np.uint8)
want = vq_oracle.lbg_iter(blocks, cb)
with emulated_kernels():
    eng = EmuEngine()
    b = torch.from_numpy(blocks)
    acc, dist, labels = eng.lbg_accum(b, torch.from_numpy(cb), want_labels=True, mode=1, sorted=eng.sort_blocks(b))
a = acc.numpy().astype(np.uint64)
assert np.array_equal(labels.numpy(), want["labels"]), "labels"
assert np.array_equal(a[: K * d].reshape(K, d), want["sums"]) and np.array_equal(a[K * d:], want["counts"]), "sums"
print("ok")
"""


def test_emu_lbg_iteration_one_tile_per_workgroup_odd_tiles_per_window():
    """The fused pass as a large chip launches it for a medium input: one 1536-vector tile per workgroup (five and a
    third per sort window), 40 workgroups, the tiles of every XCD's share taken middle-out.  The emulated device
    reports 16 compute units in this process (with the default 4 the same input takes the resident-workgroup form)."""
    import subprocess
    import sys
    env = dict(os.environ, VQ_EMU_CUS="16")
    r = subprocess.run([sys.executable, "-c", _MANY_CUS_SCRIPT, os.path.dirname(os.path.abspath(__file__))], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-4000:]


def test_emu_decode_pack_split_sum(eng, oracle):
    rng = np.random.default_rng(2)
    K, bh, bw, C = 20, 2, 4, 3
    cb = rng.random((K, bh * bw * C)) * 300.0 - 20.0
    labels = rng.integers(0, K, size=(9, 7)).astype(np.int32)
    assert np.array_equal(eng.decode(T(labels), T(cb), bh, bw, C).numpy(), oracle.decode_u8(labels, cb, bh, bw, C))
    flat = labels.reshape(-1)
    packed = eng.pack_labels(T(flat), K)
    assert np.array_equal(packed.numpy(), oracle.pack_labels(flat, K))
    assert np.array_equal(eng.unpack_labels(packed, flat.size, K).numpy(), flat)
    assert np.array_equal(eng.lbg_split(T(cb), 0.01).numpy(), oracle.split(cb, 0.01))
    v = rng.random(20000) * 100.0
    assert eng.numpy_sum(T(v)).item() == oracle.numpy_sum(v)
    blocks = rng.integers(0, 256, size=(3000, 12), dtype=np.uint8)
    assert np.array_equal(eng.colsum(T(blocks)).numpy().astype(np.uint64), oracle.colsum_u8(blocks))


def test_emu_codebook_class_reproduces_reference_golden(tmp_path):
    """Codebook.generate_codebook + compress + decompress of the package, every kernel emulated, against
    the outputs of VQ_Full.py (golden synth_iid_128_4x4_k32; the K = 32 stage runs the pruned search
    with the fused accumulation)."""
    from vector_quantization_amd import Codebook
    from vqtest_util import load_golden
    g = load_golden("synth_iid_128_4x4_k32")
    with emulated_kernels():
        engine = EmuEngine()
        cb = Codebook.from_array(g["img"], int(g["bh"]), int(g["bw"]), base_name="emu", engine=engine,
                                 output_dir=str(tmp_path))
        assert np.array_equal(cb.blocks, g["blocks"])
        cb.generate_codebook(int(g["k"]))
        assert cb.training.iters_per_stage == list(g["iters_per_stage"])
        assert np.array_equal(np.array(cb.training.distortions), g["distortions"])
        assert np.array_equal(cb.codebook, g["codebook"])
        labels = cb.compress()
        assert np.array_equal(labels, g["labels"])
        assert np.array_equal(np.fromfile(cb.labels_bin, dtype=np.uint8), g["labels_bin"])
        arr = Codebook.decompress(cb.labels_json, cb.codebook_json, cb.reconstructed_path, engine=engine)
        assert np.array_equal(arr, g["decoded"])


@pytest.mark.parametrize("K,bh,bw,C,nr,nc", [(256, 4, 4, 1, 16, 64), (17, 2, 2, 3, 5, 8), (33, 8, 8, 3, 3, 4), (9, 3, 3, 1, 4, 5),
                                             (4100, 4, 4, 1, 9, 300), (31, 4, 4, 1, 40, 1800), (12, 4, 2, 2, 7, 44),
                                             (50, 4, 4, 1, 6, 30),
                                             # strips of a block row through LDS (segments that are not whole words, at
                                             # least 64 blocks per block row): one strip, several strips, codebook too
                                             # large for LDS, three-byte and nine-byte vectors
                                             (16, 2, 2, 1, 5, 150), (16, 2, 2, 1, 2, 4500), (7, 2, 2, 3, 4, 100),
                                             (5, 3, 3, 1, 3, 70), (300, 1, 1, 3, 2, 100), (9000, 3, 3, 1, 2, 80)])
def test_emu_decode_both_kernels(eng, oracle, K, bh, bw, C, nr, nc):
    """16-byte codewords of four 4-byte pixel rows take k_decode_quads (codebook in LDS or not, several trips),
    other word-wide block rows k_decode_blockrows_u8, the rest the byte-wise k_decode_u8."""
    rng = np.random.default_rng(K + nr)
    cb = rng.random((K, bh * bw * C)) * 300.0 - 20.0
    labels = rng.integers(0, K, size=(nr, nc)).astype(np.int32)
    want = oracle.decode_u8(labels, cb, bh, bw, C)
    assert np.array_equal(eng.decode(T(labels), T(cb), bh, bw, C).numpy(), want)


def test_emu_prepared_codebook_and_label_accumulation(eng, oracle):
    """vq_codebook_prepare + vq_encode_l1_prepared on several batches, and vq_accum_labels."""
    rng = np.random.default_rng(77)
    K, d = 48, 16
    _, cb = _structured(rng, K, d, 10)
    prepared = eng.prepare_codebook(T(cb), 2000)
    for n in (2000, 700, 1):
        blocks, _ = _structured(rng, K, d, n)
        blocks = np.clip(blocks.astype(np.int64) + rng.integers(-9, 10, size=blocks.shape), 0, 255).astype(np.uint8)
        b = T(blocks)
        want_l, want_d = oracle.encode_l1(blocks, cb, want_dist=True)
        got_l, got_d = eng.encode_prepared(b, prepared, want_dist=True)
        assert np.array_equal(got_l.numpy(), want_l) and np.array_equal(got_d.numpy(), want_d)
        got_l2 = eng.encode_prepared(b, prepared, sorted=eng.sort_blocks(b))
        assert np.array_equal(got_l2.numpy(), want_l)
        acc = eng.accum_labels(b, got_l, K).numpy().astype(np.uint64)
        sums, counts = oracle.accumulate(blocks, want_l, K)
        assert np.array_equal(acc[: K * d].reshape(K, d), sums) and np.array_equal(acc[K * d:], counts)


@pytest.mark.parametrize("K,d,N", [(100, 48, 1500), (64, 64, 1200), (40, 33, 900)])
def test_emu_pruned_search_for_wide_rows(eng, oracle, K, d, N):
    """Rows of 34 .. 64 bytes (RGB 4x4 blocks) through the pruned search (two vectors per lane): encode and
    one LBG pass (separate accumulation) against the oracle."""
    rng = np.random.default_rng(K + d)
    blocks, cb = _structured(rng, K, d, N)
    want = oracle.lbg_iter(blocks, cb)
    b = T(blocks)
    srt = eng.sort_blocks(b)
    assert eng.prune_supported(K, d)
    got_l, got_d = eng.encode(b, T(cb), want_dist=True, mode=1, sorted=srt)
    acc, dist, labels = eng.lbg_accum(b, T(cb), want_labels=True, mode=1, sorted=srt)
    assert np.array_equal(got_l.numpy(), want["labels"]) and np.array_equal(got_d.numpy(), want["min_dist"])
    a = acc.numpy().astype(np.uint64)
    assert np.array_equal(labels.numpy(), want["labels"])
    assert np.array_equal(a[: K * d].reshape(K, d), want["sums"]) and np.array_equal(a[K * d:], want["counts"])
    assert dist.item() == oracle.numpy_sum(want["min_dist"])


@pytest.mark.parametrize("K,d,N,use_sorted", [(64, 16, 800, True), (200, 12, 500, False), (33, 48, 300, False), (2, 16, 300, False)])
def test_emu_pruned_fixup(eng, oracle, K, d, N, use_sorted):
    """The fix-up pass with the element-sum bound on data full of exact ties and near-ties, so that most
    vectors are queued: labels, distances and cluster sums as the oracle's."""
    rng = np.random.default_rng(K * 11 + d)
    base = np.clip(rng.random((max(2, K // 3), 1)) * 255 + rng.normal(0, 15, (max(2, K // 3), d)), 0, 255)
    cb = base[rng.integers(0, len(base), K)] + rng.choice([0.0, 0.001, -0.002, 0.5], size=(K, 1))     # twins and near-twins
    blocks = np.clip(np.rint(cb[rng.integers(0, K, N)]) + rng.integers(-1, 2, size=(N, d)), 0, 255).astype(np.uint8)
    want = oracle.lbg_iter(blocks, cb)
    b = T(blocks)
    srt = eng.sort_blocks(b) if use_sorted else None
    got_l, got_d = eng.encode(b, T(cb), want_dist=True, mode=1, sorted=srt)
    acc, dist, labels = eng.lbg_accum(b, T(cb), want_labels=True, mode=1, sorted=srt)
    assert np.array_equal(got_l.numpy(), want["labels"]) and np.array_equal(got_d.numpy(), want["min_dist"])
    a = acc.numpy().astype(np.uint64)
    assert np.array_equal(labels.numpy(), want["labels"])
    assert np.array_equal(a[: K * d].reshape(K, d), want["sums"]) and np.array_equal(a[K * d:], want["counts"])
    assert dist.item() == oracle.numpy_sum(want["min_dist"])


@pytest.mark.parametrize("shape,bh,bw,pad", [((64, 96, 1), 4, 4, 0), ((37, 53, 1), 4, 4, 0), ((37, 53, 1), 4, 4, 1), ((30, 22, 3), 2, 2, 0),
                                             ((19, 27, 1), 2, 2, 1), ((33, 20, 3), 4, 4, 0), ((21, 50, 1), 2, 4, 0), ((40, 36, 2), 3, 3, 0),
                                             ((400, 400, 1), 4, 4, 0), ((198, 202, 3), 4, 4, 0), ((198, 202, 3), 4, 4, 1), ((260, 128, 1), 8, 4, 0)])
def test_emu_sort_image_equals_extract_then_sort(eng, oracle, shape, bh, bw, pad):
    """Image -> sorted rows in one pass: a valid window-sorted copy of exactly the block matrix that
    block extraction gives (word and byte paths, both pad modes, ragged sizes, more than one window)."""
    rng = np.random.default_rng(shape[0] * 7 + shape[1] + bh)
    img = np.clip(rng.integers(0, 200, size=shape[:2] + (1,)) + rng.integers(0, 56, size=shape), 0, 255).astype(np.uint8)
    blocks = oracle.blocks_u8(img, bh, bw, pad)
    buf, (N, d) = eng.sort_image(T(img), bh, bw, pad)
    assert (N, d) == blocks.shape
    _check_sorted(buf, blocks)


@pytest.mark.parametrize("shape,bh,bw,K", [((120, 136, 1), 4, 4, 64), ((50, 61, 3), 2, 2, 40), ((64, 64, 3), 4, 4, 48), ((70, 90, 1), 4, 4, 8),
                                           ((40, 40, 3), 8, 8, 33)])
def test_emu_encode_image(eng, oracle, shape, bh, bw, K):
    """vq_encode_l1_img (pruned search on rows sorted straight from the image; plain search for shapes the
    pruned one does not serve) and one LBG pass without the block matrix, on data full of near-ties."""
    rng = np.random.default_rng(K * 5 + bh)
    d = bh * bw * shape[2]
    base = np.clip(rng.random((max(2, K // 3), 1)) * 255 + rng.normal(0, 15, (max(2, K // 3), d)), 0, 255)
    cb = base[rng.integers(0, len(base), K)] + rng.choice([0.0, 0.001, -0.002, 0.5], size=(K, 1))
    img = np.clip(rng.integers(0, 200, size=shape[:2] + (1,)) + rng.integers(0, 56, size=shape), 0, 255).astype(np.uint8)
    blocks = oracle.blocks_u8(img, bh, bw, 0)
    blocks[: len(blocks) // 2] = np.clip(np.rint(cb[rng.integers(0, K, len(blocks) // 2)]), 0, 255).astype(np.uint8)
    # put the edited vectors back into the image so that image and blocks agree (sizes are multiples of the block)
    if shape[0] % bh == 0 and shape[1] % bw == 0:
        nr, nc = shape[0] // bh, shape[1] // bw
        img = np.ascontiguousarray(blocks.reshape(nr, nc, bh, bw, shape[2]).swapaxes(1, 2).reshape(shape))
    else:
        blocks = oracle.blocks_u8(img, bh, bw, 0)
    want = oracle.lbg_iter(blocks, cb)
    got_l, got_d = eng.encode_image(T(img), bh, bw, T(cb), want_dist=True)
    assert np.array_equal(got_l.numpy(), want["labels"]) and np.array_equal(got_d.numpy(), want["min_dist"])
    if eng.prune_supported(K, d):
        srt, nd = eng.sort_image(T(img), bh, bw)
        acc, dist, labels = eng.lbg_accum(None, T(cb), want_labels=True, mode=1, sorted=srt, shape=nd)
        a = acc.numpy().astype(np.uint64)
        assert np.array_equal(labels.numpy(), want["labels"])
        assert np.array_equal(a[: K * d].reshape(K, d), want["sums"]) and np.array_equal(a[K * d:], want["counts"])
        assert dist.item() == oracle.numpy_sum(want["min_dist"])


@pytest.mark.parametrize("K,d,N", [(40, 192, 700), (600, 48, 900), (2100, 16, 1300), (4096, 4, 1500), (33, 96, 500), (130, 66, 520)])
def test_emu_pruned_search_lds_tiled(eng, oracle, K, d, N):
    """Shapes whose codewords stream through LDS tiles (long rows, codebooks beyond the scalar cache), pruned
    at workgroup granularity: encode and one LBG pass on data with spread-out element sums and near-ties."""
    rng = np.random.default_rng(K * 13 + d)
    blocks, cb = _structured(rng, K, d, N)
    cb[K // 2] = cb[K // 3]                              # duplicate codeword
    blocks[: N // 10] = np.clip(np.rint(cb[rng.integers(0, K, N // 10)]), 0, 255).astype(np.uint8)   # near-ties
    want = oracle.lbg_iter(blocks, cb)
    b = T(blocks)
    srt = eng.sort_blocks(b)
    assert eng.prune_supported(K, d)
    got_l, got_d = eng.encode(b, T(cb), want_dist=True, mode=1, sorted=srt)
    assert np.array_equal(got_l.numpy(), want["labels"]) and np.array_equal(got_d.numpy(), want["min_dist"])
    acc, dist, labels = eng.lbg_accum(None, T(cb), want_labels=True, mode=1, sorted=srt, shape=blocks.shape)
    a = acc.numpy().astype(np.uint64)
    assert np.array_equal(labels.numpy(), want["labels"])
    assert np.array_equal(a[: K * d].reshape(K, d), want["sums"]) and np.array_equal(a[K * d:], want["counts"])
    assert dist.item() == oracle.numpy_sum(want["min_dist"])


def test_emu_decode_keeps_stray_labels_inside_the_codebook(eng, oracle):
    """A label outside [0, K) is the caller's error; the kernels must still not read outside the codebook
    (guard pages end the run if they do) and decode it as codeword K - 1."""
    rng = np.random.default_rng(3)
    for K, bh, bw, C, nr, nc in [(20, 4, 4, 1, 6, 8), (7, 2, 2, 3, 5, 8), (9, 3, 3, 1, 4, 5)]:
        cb = rng.random((K, bh * bw * C)) * 255.0
        labels = rng.integers(0, K, size=(nr, nc)).astype(np.int32)
        bad = labels.copy()
        bad[1, 2], bad[3, 0] = K + 1000, -5
        want = labels.copy()
        want[1, 2] = want[3, 0] = K - 1
        assert np.array_equal(eng.decode(T(bad), T(cb), bh, bw, C).numpy(), oracle.decode_u8(want, cb, bh, bw, C))
