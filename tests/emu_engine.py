"""Runs the package's own HipEngine host code against the CPU emulation build of the kernels
(tests/emu): the engine's buffers become CPU tensors, its library handle the emulated library.
Test infrastructure only -- it lets `-m "not gpu"` tests step through the real kernel source."""
import contextlib
import ctypes
from unittest import mock

import torch

from emu.build_emu import build
import vector_quantization_amd.engine as engine_mod
from vector_quantization_amd import _lib

_emu_lib = None


def emu_lib():
    global _emu_lib
    if _emu_lib is None:
        lib = ctypes.CDLL(build())
        for name, (res, args) in _lib.SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        assert lib.vq_abi_version() == _lib.ABI_VERSION
        _emu_lib = lib
    return _emu_lib


class EmuEngine(engine_mod.HipEngine):
    """HipEngine with host tensors; every launch goes to the emulated kernels."""

    def __init__(self, mode="fast"):
        self.device = torch.device("cpu")
        self.mode = _lib.VQ_MODE_FAST if mode == "fast" else _lib.VQ_MODE_EXACT
        self._ws = None
        self._scratch_n = -1
        self._labels_scratch = None
        self._dist_scratch = None

    def _stream(self):
        return ctypes.c_void_p(0)

    def synchronize(self):
        pass


@contextlib.contextmanager
def emulated_kernels():
    """Inside this context HipEngine-derived engines call the emulated library."""
    with mock.patch.object(engine_mod, "lib", emu_lib()), \
            mock.patch.object(torch.cuda, "device", lambda *_a, **_k: contextlib.nullcontext()):
        yield
