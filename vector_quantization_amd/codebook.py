"""Drop-in for the `Codebook` class of VQ_Full.py, running its numeric work on an MI355X.

Same constructor, attributes, method names, return types, printed lines, exceptions and output
files as VQ_Full.py:11-188; the NumPy/SciPy arithmetic inside the methods is replaced by HIP
kernels reached through the engine (engine.py -> libvqhip.so).  Host code here only moves
arrays, formats files and decides when training stops.
"""
import json
import os

import numpy as np
import torch

from . import lbg as _lbg

# Output directory for the files a Codebook writes.  The reference uses the directory of its own
# script (VQ_Full.py:9) and reads it when a Codebook is constructed; here it defaults to the
# current working directory and can be reassigned the same way (or per object with output_dir=).
script_dir = os.getcwd()


def _default_engine(device=None, mode="fast"):
    from .engine import HipEngine
    return HipEngine(device=device, mode=mode)


class Codebook:
    """VQ_Full.Codebook on the GPU.

    Codebook(path, block_h, block_w) follows VQ_Full.py:12-48.  Keyword-only extras (not in the
    reference): device, mode ("fast" | "exact", identical results), engine (an object with the
    HipEngine interface), output_dir, group (torch.distributed process group when several ranks
    each hold a shard of the block vectors).
    """

    def __init__(self, path, block_h, block_w, *, device=None, mode="fast", engine=None, output_dir=None,
                 group=None):
        from PIL import Image
        img = Image.open(path).convert("RGB")              # VQ_Full.py:17
        self._setup(np.array(img), path, block_h, block_w, device, mode, engine, output_dir, group, pad_mode=0)

    @classmethod
    def from_array(cls, img, block_h, block_w, *, base_name="array", device=None, mode="fast", engine=None,
                   output_dir=None, group=None, pad_mode=0):
        """Build from an in-memory uint8 image, (H, W) or (H, W, C); not in the reference, which
        only takes a path and always converts to RGB.  pad_mode 0 = edge (VQ_Full.py:28-33),
        1 = constant zero (old/CodeBook_VQ.py:26)."""
        self = object.__new__(cls)
        arr = img.detach().cpu().numpy() if isinstance(img, torch.Tensor) else np.asarray(img)
        if arr.dtype != np.uint8:
            raise ValueError("image must be uint8")
        if arr.ndim == 2:
            arr = arr[:, :, None]
        if arr.ndim != 3:
            raise ValueError("image must be (H, W) or (H, W, C)")
        self._setup(arr, base_name, block_h, block_w, device, mode, engine, output_dir, group, pad_mode)
        return self

    def _setup(self, img_arr, path, block_h, block_w, device, mode, engine, output_dir, group, pad_mode):
        self.path = path
        self.block_h = block_h
        self.block_w = block_w
        if block_h <= 0 or block_w <= 0:
            raise ValueError("Block height and width must be positive integers.")
        self.img_arr = img_arr
        self.orig_h, self.orig_w, self.channels = self.img_arr.shape
        self._pad_mode = pad_mode
        self._engine = engine if engine is not None else _default_engine(device, mode)
        self._group = group

        pad_h = (self.block_h - (self.orig_h % self.block_h)) % self.block_h       # VQ_Full.py:23-24
        pad_w = (self.block_w - (self.orig_w % self.block_w)) % self.block_w
        self.padded_h, self.padded_w = self.orig_h + pad_h, self.orig_w + pad_w
        self.n_rows = self.padded_h // self.block_h
        self.n_cols = self.padded_w // self.block_w

        # the image goes to the device once; the (N, d) block matrix (padding and extraction are one
        # kernel) and its window-sorted second layout are built from it on first use
        self._img_dev = torch.from_numpy(np.ascontiguousarray(self.img_arr)).to(self._engine.device)
        self._n_vectors = self.n_rows * self.n_cols
        self._dim = block_h * block_w * self.channels
        self._blocks_dev_cache = None
        self._blocks_host = None
        self._blocks_sorted = None
        self._img_padded = None
        self._codebook_dev = None
        self._codebook = None

        out_dir = output_dir if output_dir is not None else script_dir
        self.base_name = os.path.splitext(os.path.basename(str(self.path)))[0]
        self.codebook_json = os.path.join(out_dir, f"{self.base_name}_codebook.json")
        self.codebook_txt = os.path.join(out_dir, f"{self.base_name}_codebook.txt")
        self.labels_json = os.path.join(out_dir, f"{self.base_name}_labels.json")
        self.labels_bin = os.path.join(out_dir, f"{self.base_name}_labels.bin")
        self.reconstructed_path = os.path.join(out_dir, f"{self.base_name}_reconstructed.png")

    # -- attributes the reference materialises eagerly; fetched from the device on first use --------
    @property
    def _blocks_dev(self):
        if self._blocks_dev_cache is None:
            self._blocks_dev_cache = self._engine.blocks_from_image(self._img_dev, self.block_h, self.block_w,
                                                                    self._pad_mode)
        return self._blocks_dev_cache

    @property
    def blocks(self):
        """(N, d) uint8 block matrix (VQ_Full.py:37, :51-57)."""
        if self._blocks_host is None:
            self._blocks_host = self._blocks_dev.cpu().numpy()
        return self._blocks_host

    @property
    def img_padded(self):
        """The padded image of VQ_Full.py:28-36, rebuilt from the block matrix."""
        if self._img_padded is None:
            b = self.blocks.reshape(self.n_rows, self.n_cols, self.block_h, self.block_w, self.channels)
            self._img_padded = np.ascontiguousarray(b.swapaxes(1, 2)).reshape(self.padded_h, self.padded_w,
                                                                                self.channels)
        return self._img_padded

    def image_to_blocks(self):
        return self.blocks

    @property
    def codebook(self):
        """(K, d) float64 codebook on the host, as VQ_Full.py:88 leaves it.  The reference reads this
        attribute afresh in compress (VQ_Full.py:124), so assigning another codebook must take effect:
        the setter drops the device copy, and compress re-uploads whenever the host array is no longer
        the one the device copy was made from (in-place edits included)."""
        return self._codebook

    @codebook.setter
    def codebook(self, value):
        self._codebook = value
        self._codebook_dev = None
        self._codebook_dev_src = None

    def _sorted_blocks(self):
        """Second device layout of the block matrix (windows sorted by element sum) that lets the
        search skip codewords; built on first use, shared by training and compress."""
        if self._blocks_sorted is None:
            if hasattr(self._engine, "sort_image"):           # image -> sorted rows in one pass
                self._blocks_sorted, _ = self._engine.sort_image(self._img_dev, self.block_h, self.block_w,
                                                                 self._pad_mode)
            else:
                self._blocks_sorted = self._engine.sort_blocks(self._blocks_dev)
        return self._blocks_sorted

    # -- training (VQ_Full.py:60-117) -------------------------------------------------------------------
    def generate_codebook(self, k, epsilon=0.01, threshold=0.001, max_iterations=100, *, iters_per_stage=None):
        n_local = self._n_vectors
        if self._group is None and k > n_local:
            raise ValueError(
                f"Invalid quantization level k={k}: cannot exceed the total number of image blocks ({n_local})."
            )
        print(f"\n=== Starting LBG for k={k} ===")
        res = _lbg.train_lbg(self._engine, self._blocks_dev, k, epsilon, threshold, max_iterations,
                             group=self._group, log=print, iters_per_stage=iters_per_stage,
                             sorted=self._sorted_blocks())
        self.codebook = res.codebook.cpu().numpy()
        self._codebook_dev = res.codebook
        self._codebook_dev_src = self._codebook.copy()
        self.training = res

        final = self.codebook.reshape(-1, self.block_h, self.block_w, self.channels).tolist()
        with open(self.codebook_json, "w") as f:
            json.dump(final, f, indent=4)
        print(f"✓ Codebook saved to JSON: {self.codebook_json}")

        with open(self.codebook_txt, "w") as f:
            f.write(f"{'Level':<6}{'Min':>10}{'Max':>10}{'Dequantized':>30}\n")
            f.write("-" * 60 + "\n")
            for idx, vec in enumerate(self.codebook):
                min_val = vec.min()
                max_val = vec.max()
                dequant_val = np.round(vec.mean(), 2)
                f.write(f"{idx:<6}{min_val:>10.2f}{max_val:>10.2f}{dequant_val:>30.2f}\n")
        print(f"✓ Codebook saved as formatted TXT: {self.codebook_txt}")
        return final

    # -- encode (VQ_Full.py:120-150) ----------------------------------------------------------------------
    def compress(self):
        if self.codebook is None:
            raise ValueError("No codebook yet.")
        host = np.ascontiguousarray(self.codebook, dtype=np.float64)
        if host.ndim != 2 or host.shape[1] != self._dim:
            raise ValueError(f"codebook must be (K, {self._dim}), got {host.shape}")
        if (self._codebook_dev is None or self._codebook_dev_src is None
                or self._codebook_dev_src.shape != host.shape
                or not np.array_equal(self._codebook_dev_src.view(np.uint64), host.view(np.uint64))):
            # assigned or edited by the caller since the device copy was made
            self._codebook_dev = torch.from_numpy(host).to(self._engine.device)
            self._codebook_dev_src = host.copy()
        srt = self._sorted_blocks()
        if (srt is not None and self._blocks_dev_cache is None and hasattr(self._engine, "sort_image")
                and self._engine.prune_supported(host.shape[0], self._dim) and getattr(self._engine, "mode", 1) == 1):
            # the search runs on the sorted rows made straight from the image: no block matrix at all
            labels_dev = self._engine.encode(None, self._codebook_dev, sorted=srt, shape=(self._n_vectors, self._dim))
        else:
            labels_dev = self._engine.encode(self._blocks_dev, self._codebook_dev, sorted=srt)
        self._labels_dev = labels_dev
        labels_grid = labels_dev.cpu().numpy().astype(np.int64).reshape(self.n_rows, self.n_cols)

        with open(self.labels_json, "w") as f:
            json.dump(labels_grid.tolist(), f)
        print(f"✓ Labels saved as JSON: {self.labels_json}")

        packed = self._engine.pack_labels(labels_dev, len(self.codebook))
        with open(self.labels_bin, "wb") as f:
            f.write(packed.cpu().numpy().tobytes())
        print(f"✓ Labels saved as binary: {self.labels_bin}")
        return labels_grid

    # -- decode (VQ_Full.py:153-172) ------------------------------------------------------------------------
    @staticmethod
    def decompress(labels_path, codebook_path, output_path, *, device=None, engine=None, crop=None):
        """Called on the class, as the reference's (self-less) method is: shapes come from the two
        JSON files only.  crop=(orig_h, orig_w) cuts the padding off the result (the reference never
        crops, VQ_Full.py:169; old/Fatma_CodeBook_VQ.py:105 shows the intended crop); default off."""
        with open(labels_path) as f:
            labels = np.array(json.load(f))
        with open(codebook_path) as f:
            codebook = np.array(json.load(f))
        n_rows, n_cols = labels.shape
        block_h, block_w, channels = codebook.shape[1], codebook.shape[2], codebook.shape[3]
        # codebook[labels[r][c]] (VQ_Full.py:162): a negative index counts from the end, anything outside
        # [-K, K) raises IndexError -- checked here, because the gather kernel does not range-check
        K = len(codebook)
        if labels.size and not np.issubdtype(labels.dtype, np.integer):
            raise IndexError("only integers, slices (`:`), ellipsis (`...`), numpy.newaxis (`None`) and integer or "
                             "boolean arrays are valid indices")
        bad = (labels < -K) | (labels >= K)
        if bad.any():
            v = int(labels[bad].ravel()[0])
            raise IndexError(f"index {v} is out of bounds for axis 0 with size {K}")
        labels = np.where(labels < 0, labels + K, labels)
        eng = engine if engine is not None else _default_engine(device)
        lab_dev = torch.from_numpy(labels.astype(np.int32)).to(eng.device)
        cb_dev = torch.from_numpy(np.ascontiguousarray(codebook, dtype=np.float64).reshape(len(codebook), -1)).to(
            eng.device)
        arr = eng.decode(lab_dev, cb_dev, block_h, block_w, channels).cpu().numpy()
        if crop is not None:
            arr = np.ascontiguousarray(arr[: int(crop[0]), : int(crop[1])])
        from PIL import Image
        if channels == 3:
            Image.fromarray(arr).save(output_path)
        elif channels == 1:
            Image.fromarray(arr[:, :, 0]).save(output_path)
        else:
            Image.fromarray(arr).save(output_path)
        print(f"✓ Decompression done. Saved as {output_path}")
        return arr


def validate_image_path(path, allowed_exts=None):
    """Follows VQ_Full.py:174-188."""
    if allowed_exts is None:
        allowed_exts = [".png", ".jpg", ".jpeg", ".bmp", ".tiff"]
    if not os.path.isabs(path):
        path = os.path.join(script_dir, path)
    if not os.path.isfile(path):
        raise FileNotFoundError(f"File does not exist: {path}")
    ext = os.path.splitext(path)[1].lower()
    if ext not in allowed_exts:
        raise ValueError(f"Invalid file extension '{ext}'. Allowed: {', '.join(allowed_exts)}")
    return path
