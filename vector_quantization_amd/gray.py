"""Drop-in for the grayscale `Codebook` of old/CodeBook_VQ.py (the reference's single-channel
variant): image opened with convert("L"), bottom/right padding with constant 0
(old/CodeBook_VQ.py:22-26), vectors of block_h*block_w bytes, JSON shaped [K][block_h][block_w].
The LBG arithmetic is the same as VQ_Full.py's and runs on the same HIP kernels.
"""
import json

import numpy as np
import torch

from . import lbg as _lbg
from ._lib import VQ_PAD_ZERO


class GrayCodebook:
    """old/CodeBook_VQ.py:8-91 on the GPU: Codebook(path, block_h, block_w), .blocks,
    .generate_codebook(k, json_path="codebook.json", epsilon, threshold, max_iterations)."""

    def __init__(self, path, block_h, block_w, *, device=None, mode="fast", engine=None, image=None):
        self.path = path
        self.block_h = block_h
        self.block_w = block_w
        if engine is None:
            from .engine import HipEngine
            engine = HipEngine(device=device, mode=mode)
        self._engine = engine
        self._image = image
        self._blocks_dev = None
        self.blocks = self.image_to_blocks()
        self.codebook = None

    def image_to_blocks(self):
        if self._image is not None:
            img_arr = np.asarray(self._image, dtype=np.uint8)
        else:
            from PIL import Image
            img_arr = np.array(Image.open(self.path).convert("L"))      # old/CodeBook_VQ.py:18-19
        dev = torch.from_numpy(np.ascontiguousarray(img_arr)).to(self._engine.device)
        self._blocks_dev = self._engine.blocks_from_image(dev, self.block_h, self.block_w, VQ_PAD_ZERO)
        return self._blocks_dev.cpu().numpy()

    def generate_codebook(self, k, json_path="codebook.json", epsilon=0.01, threshold=0.001, max_iterations=100):
        if k > len(self.blocks):
            raise ValueError("k is larger than number of blocks!")                 # old/CodeBook_VQ.py:40-41
        print(f"Starting Optimized LBG for k={k}...")
        res = _lbg.train_lbg(
            self._engine, self._blocks_dev, k, epsilon, threshold, max_iterations, log=None,
            on_converged=lambda size, i, dist: print(                              # old/CodeBook_VQ.py:79
                f"  Converged for size {size} at iter {i} (Distortion: {dist:.2f})"))
        self.codebook = res.codebook.cpu().numpy()
        self.training = res
        print(f"✓ Codebook generated. Final size: {len(self.codebook)}")
        final_codebook_list = self.codebook.reshape(-1, self.block_h, self.block_w).tolist()
        with open(json_path, "w") as f:
            json.dump(final_codebook_list, f, indent=4)
        return final_codebook_list
