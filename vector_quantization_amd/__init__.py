"""vector_quantization_amd -- the VQ hot path of VQ_Full.py (block extraction, L1 nearest-codeword
search, LBG codebook training, decode) on AMD MI355X (gfx950), behind the reference's own
`Codebook` interface.

Importing the package loads libvqhip.so; a missing library is an ImportError-time failure, there
is no CPU fallback.
"""
from . import _lib  # noqa: F401  (loads and type-checks the C ABI; raises if the library is missing)
from .codebook import Codebook, validate_image_path
from .engine import HipEngine
from .gray import GrayCodebook
from .pipeline import EncodePipeline
from .lbg import train_lbg, lbg_iteration, initial_centroid, final_codebook_size, LBGResult
from ._lib import VQ_MODE_EXACT, VQ_MODE_FAST, VQ_PAD_EDGE, VQ_PAD_ZERO

__all__ = [
    "Codebook", "GrayCodebook", "validate_image_path", "HipEngine", "EncodePipeline", "train_lbg", "lbg_iteration", "initial_centroid",
    "final_codebook_size", "LBGResult", "VQ_MODE_EXACT", "VQ_MODE_FAST", "VQ_PAD_EDGE", "VQ_PAD_ZERO",
]
