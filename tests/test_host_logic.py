"""Host logic of the package on a CPU-only machine: the reference-shaped Codebook class (file
formats, printed lines, exceptions, attributes) and the LBG control loop, driven through the
OracleEngine test double.  The numeric kernels themselves are checked in test_gpu_parity.py."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle_engine import OracleEngine
from vqtest_util import load_golden


@pytest.fixture()
def engine():
    return OracleEngine()


def test_codebook_class_end_to_end(tmp_path, train_case, engine, capsys):
    from vector_quantization_amd import Codebook
    g = train_case
    cb = Codebook.from_array(g["img"], int(g["bh"]), int(g["bw"]), base_name=g["name"], engine=engine,
                             output_dir=str(tmp_path))
    H, W, C = g["img"].shape
    assert (cb.orig_h, cb.orig_w, cb.channels) == (H, W, C)
    assert (cb.n_rows, cb.n_cols) == (int(g["n_rows"]), int(g["n_cols"]))
    assert cb.padded_h == cb.n_rows * cb.block_h and cb.padded_w == cb.n_cols * cb.block_w
    assert np.array_equal(cb.blocks, g["blocks"])
    ph, pw = cb.padded_h - H, cb.padded_w - W
    assert np.array_equal(cb.img_padded, np.pad(g["img"], ((0, ph), (0, pw), (0, 0)), mode="edge"))
    assert cb.codebook is None
    with pytest.raises(ValueError, match="No codebook yet."):
        cb.compress()

    final = cb.generate_codebook(int(g["k"]))
    out = capsys.readouterr().out
    assert f"=== Starting LBG for k={int(g['k'])} ===" in out
    assert [l for l in out.splitlines() if l.startswith("Converged")] == \
           [l for l in str(g["log"]).splitlines() if l.startswith("Converged")]
    assert np.array_equal(cb.codebook, g["codebook"])
    assert isinstance(final, list) and isinstance(final[0][0][0][0], float)
    assert np.array(final).shape == (len(g["codebook"]), int(g["bh"]), int(g["bw"]), C)
    assert cb.training.iters_per_stage == list(g["iters_per_stage"])
    assert np.array_equal(np.array(cb.training.distortions), g["distortions"])      # numpy-order mean, bit for bit

    labels = cb.compress()
    assert labels.dtype == np.int64 and labels.shape == g["labels"].shape
    assert np.array_equal(labels, g["labels"])
    # files byte-identical to the reference's
    sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()
    assert sha(cb.codebook_json) == str(g["codebook_json_sha256"])
    assert sha(cb.codebook_txt) == str(g["codebook_txt_sha256"])
    assert sha(cb.labels_json) == str(g["labels_json_sha256"])
    assert np.array_equal(np.fromfile(cb.labels_bin, dtype=np.uint8), g["labels_bin"])
    assert os.path.basename(cb.codebook_json) == f"{g['name']}_codebook.json"

    arr = Codebook.decompress(cb.labels_json, cb.codebook_json, cb.reconstructed_path, engine=engine)
    assert arr.dtype == np.uint8 and np.array_equal(arr, g["decoded"])
    assert os.path.isfile(cb.reconstructed_path)
    assert "✓ Decompression done." in capsys.readouterr().out


def test_text_outputs_match_reference_text(tmp_path, engine):
    from vector_quantization_amd import Codebook
    g = load_golden("luffy_2x2_k16")
    cb = Codebook.from_array(g["img"], 2, 2, base_name="luffy", engine=engine, output_dir=str(tmp_path))
    cb.generate_codebook(16)
    cb.compress()
    assert open(cb.codebook_json).read() == str(g["codebook_json_text"])
    assert open(cb.codebook_txt).read() == str(g["codebook_txt_text"])
    assert open(cb.labels_json).read() == str(g["labels_json_text"])


def test_path_constructor_converts_to_rgb(tmp_path, engine):
    from PIL import Image
    from vector_quantization_amd import Codebook
    import vector_quantization_amd.codebook as cbmod
    gray = (np.arange(35 * 21).reshape(35, 21) % 256).astype(np.uint8)
    p = tmp_path / "g.png"
    Image.fromarray(gray).save(p)
    old = cbmod.script_dir
    cbmod.script_dir = str(tmp_path)          # the reference's output-directory convention
    try:
        cb = Codebook(str(p), 4, 4, engine=engine)
    finally:
        cbmod.script_dir = old
    assert cb.channels == 3 and cb.img_arr.shape == (35, 21, 3)
    assert cb.blocks.shape == (9 * 6, 48)
    assert cb.labels_bin == os.path.join(str(tmp_path), "g_labels.bin")
    assert cb.base_name == "g"


def test_k_exceeds_blocks(engine, tmp_path):
    from vector_quantization_amd import Codebook
    img = np.zeros((4, 4, 1), dtype=np.uint8)
    cb = Codebook.from_array(img, 2, 2, engine=engine, output_dir=str(tmp_path))
    with pytest.raises(ValueError, match=r"Invalid quantization level k=5: cannot exceed the total number of image "
                                         r"blocks \(4\)\."):
        cb.generate_codebook(5)


def test_bad_block_size(engine, tmp_path):
    from vector_quantization_amd import Codebook
    with pytest.raises(ValueError):
        Codebook.from_array(np.zeros((4, 4), dtype=np.uint8), 0, 2, engine=engine, output_dir=str(tmp_path))


def test_validate_image_path(tmp_path):
    from vector_quantization_amd import validate_image_path
    with pytest.raises(FileNotFoundError, match="File does not exist"):
        validate_image_path(str(tmp_path / "nope.png"))
    bad = tmp_path / "a.gif"
    bad.write_bytes(b"x")
    with pytest.raises(ValueError, match="Invalid file extension '.gif'"):
        validate_image_path(str(bad))
    ok = tmp_path / "a.PNG"
    ok.write_bytes(b"x")
    assert validate_image_path(str(ok)) == str(ok)
    assert validate_image_path(str(bad), allowed_exts=[".gif"]) == str(bad)


def test_zero_distortion_never_converges(engine):
    """prev == 0 makes the relative change nan, so the stage runs to max_iterations (as the
    reference does)."""
    import torch
    from vector_quantization_amd import train_lbg
    blocks = torch.zeros((64, 4), dtype=torch.uint8)
    res = train_lbg(engine, blocks, 2, max_iterations=7, log=None)
    assert res.iters_per_stage == [7] and res.converged_iter == [-1]
    assert all(d == 0.0 for d in res.distortions)


def test_forced_iteration_counts(engine):
    import torch
    from vector_quantization_amd import train_lbg
    g = load_golden("man_2x2_k5")
    blocks = torch.from_numpy(g["blocks"])
    res = train_lbg(engine, blocks, 5, log=None, iters_per_stage=list(g["iters_per_stage"]))
    assert np.array_equal(res.codebook.numpy(), g["codebook"])
    res2 = train_lbg(engine, blocks, 5, log=None, iters_per_stage=[1, 1, 1])
    assert res2.iters_per_stage == [1, 1, 1] and res2.codebook.shape == (8, 12)


def test_final_codebook_size():
    from vector_quantization_amd import final_codebook_size
    assert [final_codebook_size(k) for k in (1, 2, 3, 5, 8, 9, 256, 257)] == [1, 2, 4, 8, 8, 16, 256, 512]


def test_gray_codebook_matches_old_reference_class(tmp_path, engine, capsys):
    """GrayCodebook against old/CodeBook_VQ.py run unmodified (golden gray_man_4x4_k16): zero padding,
    printed lines, codebook bit for bit, JSON text byte for byte."""
    from vector_quantization_amd import GrayCodebook
    g = load_golden("gray_man_4x4_k16")
    cb = GrayCodebook("unused", 4, 4, engine=engine, image=g["gray"])
    assert cb.blocks.dtype == np.uint8 and np.array_equal(cb.blocks, g["blocks"])
    jp = tmp_path / "cb.json"
    out_list = cb.generate_codebook(16, json_path=str(jp))
    assert capsys.readouterr().out == str(g["log"])
    assert np.array_equal(cb.codebook, g["codebook"])
    assert open(jp).read() == str(g["json_text"])
    assert np.array(out_list).shape == (16, 4, 4)
    with pytest.raises(ValueError, match="k is larger than number of blocks!"):
        cb.generate_codebook(len(cb.blocks) + 1, json_path=str(jp))


def test_gray_codebook_opens_file_as_luminance(tmp_path, engine):
    from PIL import Image
    from vector_quantization_amd import GrayCodebook
    rgb = np.random.default_rng(3).integers(0, 256, (10, 7, 3), dtype=np.uint8)
    p = tmp_path / "c.png"
    Image.fromarray(rgb).save(p)
    cb = GrayCodebook(str(p), 4, 4, engine=engine)
    lum = np.array(Image.open(p).convert("L"))
    padded = np.pad(lum, ((0, 2), (0, 1)), mode="constant", constant_values=0)
    want = padded.reshape(3, 4, 2, 4).swapaxes(1, 2).reshape(-1, 16)
    assert np.array_equal(cb.blocks, want)


def test_decompress_crop_option(tmp_path, engine):
    from vector_quantization_amd import Codebook
    g = load_golden("synth_gray_250x246_4x4_k64")
    cb = Codebook.from_array(g["img"], 4, 4, base_name="s", engine=engine, output_dir=str(tmp_path))
    cb.generate_codebook(int(g["k"]))
    cb.compress()
    full = Codebook.decompress(cb.labels_json, cb.codebook_json, cb.reconstructed_path, engine=engine)
    assert np.array_equal(full, g["decoded"])                       # default: padded size, as the reference
    crop = Codebook.decompress(cb.labels_json, cb.codebook_json, cb.reconstructed_path, engine=engine,
                               crop=(cb.orig_h, cb.orig_w))
    assert crop.shape[:2] == (250, 246) and np.array_equal(crop, full[:250, :246])
    from PIL import Image
    assert Image.open(cb.reconstructed_path).size == (246, 250)
