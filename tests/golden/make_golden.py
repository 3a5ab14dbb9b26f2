#!/usr/bin/env python3
"""Generate the golden parity fixtures by RUNNING the reference implementation.

Run in the build container only (it needs /root/reference, which never travels
to the GPU box):

    python tests/golden/make_golden.py

It imports the reference's VQ_Full module unmodified, redirects its output
directory to a temp dir, and records inputs and outputs of Codebook.__init__,
generate_codebook, compress and decompress as .npz files next to this script.
Per-iteration data (distortion, codebook per stage, iteration counts) is
captured by wrapping the module-level `cdist` name that generate_codebook calls
once per LBG iteration; the wrapper calls the real scipy function and records
what went in and what came out, so every recorded number is produced by the
reference's own arithmetic.

Nothing from the reference's source text is stored; the fixtures hold arrays
(pixels, codebooks, labels, distances) only.
"""
import hashlib
import io
import json
import os
import sys
import tempfile
import contextlib

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

sys.path.insert(0, REF)
import VQ_Full  # noqa: E402
from scipy.spatial.distance import cdist as real_cdist  # noqa: E402
from PIL import Image  # noqa: E402


class Recorder:
    """Wraps cdist; one call == one LBG iteration (or the compress() call)."""

    def __init__(self):
        self.calls = []  # (K, codebook copy, labels, distortion)

    def __call__(self, xa, xb, metric="euclidean"):
        dist = real_cdist(xa, xb, metric=metric)
        labels = np.argmin(dist, axis=1)
        mind = dist[np.arange(len(dist)), labels]
        self.calls.append(dict(K=len(xb), cb=np.array(xb, copy=True), labels=labels,
                               distortion=np.mean(mind), mind=mind))
        return dist


class FakeImageModule:
    """Stands in for PIL.Image inside decompress when C != 3 (saving is not under test)."""

    class _Img:
        def save(self, path):
            pass

    @staticmethod
    def fromarray(arr, mode=None):
        return FakeImageModule._Img()

    open = staticmethod(Image.open)


def make_ref_codebook_from_array(img, bh, bw, base_name, outdir):
    """Build a reference Codebook object for an in-memory (H,W,C) uint8 array.

    The reference constructor only accepts a path and always converts to RGB, so for
    synthetic and single-channel inputs the attributes it would set are assigned here and the
    reference's own image_to_blocks / generate_codebook / compress run on them unchanged.
    """
    cb = object.__new__(VQ_Full.Codebook)
    cb.path = base_name + ".png"
    cb.block_h, cb.block_w = bh, bw
    cb.img_arr = img
    cb.orig_h, cb.orig_w, cb.channels = img.shape
    pad_h = (bh - (cb.orig_h % bh)) % bh
    pad_w = (bw - (cb.orig_w % bw)) % bw
    cb.img_padded = np.pad(img, ((0, pad_h), (0, pad_w), (0, 0)), mode="edge")
    cb.padded_h, cb.padded_w, _ = cb.img_padded.shape
    cb.blocks = cb.image_to_blocks()
    cb.codebook = None
    cb.n_rows = cb.padded_h // bh
    cb.n_cols = cb.padded_w // bw
    cb.base_name = base_name
    cb.codebook_json = os.path.join(outdir, f"{base_name}_codebook.json")
    cb.codebook_txt = os.path.join(outdir, f"{base_name}_codebook.txt")
    cb.labels_json = os.path.join(outdir, f"{base_name}_labels.json")
    cb.labels_bin = os.path.join(outdir, f"{base_name}_labels.bin")
    cb.reconstructed_path = os.path.join(outdir, f"{base_name}_reconstructed.png")
    return cb


def run_case(name, bh, bw, k, path=None, img=None, kat_iter=None, keep_text=False, lean=False):
    """lean: for large codebooks -- stage-end codebooks are stored as sha256 digests (the final one in
    full), the decoded image as a digest, and the block matrix is left out (it follows from img)."""
    outdir = tempfile.mkdtemp(prefix="vqgold_")
    VQ_Full.script_dir = outdir
    rec = Recorder()
    VQ_Full.cdist = rec
    stdout = io.StringIO()
    with contextlib.redirect_stdout(stdout):
        if path is not None:
            cb = VQ_Full.Codebook(path, bh, bw)
        else:
            cb = make_ref_codebook_from_array(img, bh, bw, name, outdir)
        cb.generate_codebook(k)
        n_train_calls = len(rec.calls)
        labels = cb.compress()
        if cb.channels != 3:
            VQ_Full.Image = FakeImageModule
        decoded = VQ_Full.Codebook.decompress(cb.labels_json, cb.codebook_json, cb.reconstructed_path)
        VQ_Full.Image = Image
    VQ_Full.cdist = real_cdist
    log = stdout.getvalue()

    train = rec.calls[:n_train_calls]
    # stage boundaries: K changes
    iters_per_stage, stage_K = [], []
    for c in train:
        if not stage_K or stage_K[-1] != c["K"]:
            stage_K.append(c["K"])
            iters_per_stage.append(0)
        iters_per_stage[-1] += 1
    conv = [int(l.split("iter")[1].split(",")[0]) for l in log.splitlines() if l.startswith("Converged")]
    # codebook at the end of each stage = (next stage's first input)[:K] / 1.01 is NOT exact, so
    # recover it exactly: the final one is cb.codebook; intermediate ones are obtained by replaying
    # the last iteration's centroid update with the reference's own expressions.
    stage_end = []
    pos = 0
    for s, n_it in enumerate(iters_per_stage):
        last = train[pos + n_it - 1]
        newcb = np.zeros_like(last["cb"])
        for idx in range(len(last["cb"])):
            members = cb.blocks[last["labels"] == idx]
            newcb[idx] = np.mean(members, axis=0) if len(members) > 0 else last["cb"][idx]
        stage_end.append(newcb)
        pos += n_it
    assert np.array_equal(stage_end[-1], cb.codebook) if stage_end else True
    # cross-check: next stage's first input must be vstack(prev*1.01, prev*0.99)
    pos = 0
    for s, n_it in enumerate(iters_per_stage[:-1]):
        nxt = train[pos + n_it]["cb"]
        assert np.array_equal(nxt, np.vstack((stage_end[s] * (1 + 0.01), stage_end[s] * (1 - 0.01))))
        pos += n_it

    out = dict(
        img=cb.img_arr, bh=bh, bw=bw, k=k, eps=0.01, thr=0.001, max_iter=100,
        n_rows=cb.n_rows, n_cols=cb.n_cols,
        iters_per_stage=np.array(iters_per_stage, dtype=np.int64),
        stage_K=np.array(stage_K, dtype=np.int64),
        converged_iter=np.array(conv, dtype=np.int64),
        distortions=np.array([c["distortion"] for c in train], dtype=np.float64),
        codebook=cb.codebook, labels=labels,
        labels_bin=np.frombuffer(open(cb.labels_bin, "rb").read(), dtype=np.uint8),
        codebook_json_sha256=hashlib.sha256(open(cb.codebook_json, "rb").read()).hexdigest(),
        labels_json_sha256=hashlib.sha256(open(cb.labels_json, "rb").read()).hexdigest(),
        codebook_txt_sha256=hashlib.sha256(open(cb.codebook_txt, "rb").read()).hexdigest(),
        log=log,
    )
    if lean:
        out["stage_end_sha256"] = np.array([hashlib.sha256(np.ascontiguousarray(s).tobytes()).hexdigest()
                                            for s in stage_end])
        out["decoded_sha256"] = hashlib.sha256(np.ascontiguousarray(decoded).tobytes()).hexdigest()
        out["mind_final"] = train[-1]["mind"] if train else np.zeros(0)
    else:
        out["blocks"] = cb.blocks
        out["stage_end_codebooks"] = (np.concatenate([s.ravel() for s in stage_end]) if stage_end
                                      else np.zeros(0))
        out["decoded"] = decoded
    if keep_text:
        out["codebook_json_text"] = open(cb.codebook_json).read()
        out["codebook_txt_text"] = open(cb.codebook_txt).read()
        out["labels_json_text"] = open(cb.labels_json).read()
    if kat_iter is not None and train:
        c = train[min(kat_iter, len(train) - 1)]
        nxt = np.zeros_like(c["cb"])
        for idx in range(len(c["cb"])):
            members = cb.blocks[c["labels"] == idx]
            nxt[idx] = np.mean(members, axis=0) if len(members) > 0 else c["cb"][idx]
        out.update(it_cb_in=c["cb"], it_labels=c["labels"].astype(np.int64), it_cb_out=nxt,
                   it_distortion=c["distortion"], it_mind=c["mind"])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: N={len(cb.blocks)} d={cb.blocks.shape[1]} K={len(cb.codebook)} "
          f"iters={iters_per_stage} conv={conv} final_D={out['distortions'][-1] if train else None}")
    print("   sha labels/decoded/codebook:",
          hashlib.sha256(labels.astype(np.int32).tobytes()).hexdigest()[:16],
          hashlib.sha256(decoded.tobytes()).hexdigest()[:16],
          hashlib.sha256(cb.codebook.tobytes()).hexdigest()[:16])


def structured_gray(h, w, seed):
    """Synthetic 'structured' grayscale image: smooth waves plus Gaussian noise, clipped to u8."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    v = (128 + 64 * np.sin(2 * np.pi * x / 257) * np.cos(2 * np.pi * y / 181)
         + 32 * np.sin(2 * np.pi * (x + y) / 1021) + rng.normal(0, 16, size=(h, w)))
    return np.clip(v, 0, 255).astype(np.uint8)


def main():
    # (i) the reference's own CPU-runnable case: luffy.png, 2x2, k=16 (RGB, d=12)
    run_case("luffy_2x2_k16", 2, 2, 16, path=os.path.join(REF, "luffy.png"), kat_iter=20, keep_text=True)
    # (ii) man.jpeg 4x4 k=256 (16 empty clusters) and 2x2 k=5 (K rounds up to 8)
    run_case("man_4x4_k256", 4, 4, 256, path=os.path.join(REF, "man.jpeg"), kat_iter=30)
    run_case("man_2x2_k5", 2, 2, 5, path=os.path.join(REF, "man.jpeg"), kat_iter=5)
    # non-multiple size: 225x225 with 4x4 and 8x8 (edge padding on both axes)
    run_case("luffy_4x4_k32", 4, 4, 32, path=os.path.join(REF, "luffy.png"), kat_iter=10)
    run_case("luffy_8x8_k8", 8, 8, 8, path=os.path.join(REF, "luffy.png"), kat_iter=3)
    # (iii) synthetic gray (C=1) 256x256, 4x4, K=256 with an all-zero quadrant (duplicate codewords,
    # permanent ties) and a saturated quadrant (255*1.01 codewords kept by empty clusters -> decode wrap)
    g = structured_gray(256, 256, seed=7)
    g[:128, :128] = 0
    g[128:, 128:] = 255
    run_case("synth_gray_zero_sat_4x4_k256", 4, 4, 256, img=g[:, :, None], kat_iter=40)
    # smaller structured gray, d=16, K=64, odd size (pads by edge)
    g2 = structured_gray(250, 246, seed=11)
    run_case("synth_gray_250x246_4x4_k64", 4, 4, 64, img=g2[:, :, None], kat_iter=12)
    # k = 1: no split, no iterations
    run_case("man_2x2_k1", 2, 2, 1, path=os.path.join(REF, "man.jpeg"))
    # uniform iid noise (worst case for convergence)
    rng = np.random.default_rng(1)
    g3 = rng.integers(0, 256, size=(128, 128, 1), dtype=np.uint8)
    run_case("synth_iid_128_4x4_k32", 4, 4, 32, img=g3, kat_iter=7)

    # (v) decode known-answer vector: the literal tables in old/decompression_VC.py (a 4-entry
    # codebook of 2x2 grayscale blocks and a 3x3 grid of 2-bit codes) and the 6x6 image they decode
    # to, derived by hand from those literals.
    kat_cb = np.array([[[2, 3], [4, 5]], [[8, 9], [7, 7]], [[4, 10], [11, 11]], [[16, 15], [19, 18]]],
                      dtype=np.float64)
    kat_labels = np.array([[0, 1, 2], [2, 3, 1], [0, 3, 0]], dtype=np.int64)
    kat_img = np.array([[2, 3, 8, 9, 4, 10], [4, 5, 7, 7, 11, 11], [4, 10, 16, 15, 8, 9],
                        [11, 11, 19, 18, 7, 7], [2, 3, 16, 15, 2, 3], [4, 5, 19, 18, 4, 5]], dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "decode_kat_2x2.npz"), codebook=kat_cb, labels=kat_labels, img=kat_img)

    # fp64 -> uint8 cast table as NumPy performs it inside decompress
    vals = np.array([0.0, 0.999, 1.0, 127.5, 254.999999, 255.0, 255.999, 256.0, 257.55, 260.1, 300.7, 511.2,
                     512.0, 1000.25, 65535.9, 65536.0, -0.0, -0.5, -1.0, -1.5, -255.5, -256.0, -300.2],
                    dtype=np.float64)
    with np.errstate(invalid="ignore"):
        cast = np.array([vals], dtype=np.uint8)[0]
    np.savez_compressed(os.path.join(HERE, "cast_f64_u8.npz"), vals=vals, cast=cast)

    # mean KAT: np.mean over float64 vectors of awkward lengths, several beyond numpy's 8192-element
    # reduction buffer (where the summation order changes)
    rng = np.random.default_rng(5)
    lens = [1, 7, 8, 9, 127, 128, 129, 255, 256, 257, 1000, 1369, 4096, 5476, 8192, 8193, 8194, 8205, 8240, 8247,
            8289, 12749, 12766, 12769, 16384, 16385, 20011, 30001]
    arrs = [rng.random(n) * 300.0 for n in lens]
    np.savez_compressed(os.path.join(HERE, "pairwise_mean.npz"),
                        lens=np.array(lens), data=np.concatenate(arrs),
                        means=np.array([np.mean(a) for a in arrs]))


def gray_case():
    """old/CodeBook_VQ.py (grayscale, zero padding) run unmodified on man.jpeg, 4x4 blocks, k=16."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_gray", os.path.join(REF, "old", "CodeBook_VQ.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = tempfile.mkdtemp(prefix="vqgold_")
    jp = os.path.join(out, "cb.json")
    stdout = io.StringIO()
    with contextlib.redirect_stdout(stdout):
        cb = mod.Codebook(os.path.join(REF, "man.jpeg"), 4, 4)
        cb.generate_codebook(16, json_path=jp)
    gray = np.array(Image.open(os.path.join(REF, "man.jpeg")).convert("L"))
    np.savez_compressed(os.path.join(HERE, "gray_man_4x4_k16.npz"), gray=gray, blocks=cb.blocks, codebook=cb.codebook,
                        json_text=open(jp).read(), log=stdout.getvalue())
    print("gray_man_4x4_k16:", cb.blocks.shape, cb.codebook.shape)


def main_large():
    """Cases at the codebook sizes of the two large configurations, at vector counts the reference can
    still run here (it materialises the N x K fp64 distance matrix)."""
    # full LBG training to K = 1024 at d = 16 (grayscale 4x4): ten split stages; N = 65 536
    g = structured_gray(1024, 1024, seed=2)
    run_case("synth_gray_1024_4x4_k1024", 4, 4, 1024, img=g[:, :, None], kat_iter=60)
    # RGB 8x8 blocks (d = 192), K = 4096: team.jpg, N = 19 200
    run_case("team_8x8_k4096", 8, 8, 4096, path=os.path.join(REF, "team.jpg"), lean=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "large":
        main_large()
    else:
        main()
        gray_case()
