"""Runs the package's own HipEngine host code against the CPU emulation build of the kernels
(tests/emu): the engine's buffers become CPU tensors, its library handle the emulated library.
Test infrastructure only -- it lets `-m "not gpu"` tests step through the real kernel source."""
import contextlib
import ctypes
import mmap
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


# ---- buffers that end at a protected page ---------------------------------------------------------
# Every tensor the engine allocates (labels, distances, workspace, sorted copy, images, sums) is placed
# so that it ends right before a PROT_NONE page: a kernel that runs past the end of a buffer -- the kind
# of access that faults a GPU -- stops the emulated run with a segmentation fault at the faulting line.
_EMU_DEVICE = torch.device("cpu")              # identity marks "allocated by an EmuEngine"
_libc = ctypes.CDLL(None, use_errno=True)
_real_empty, _real_zeros = torch.empty, torch.zeros
_ITEMSIZE = {torch.uint8: 1, torch.int32: 4, torch.int64: 8, torch.float64: 8, torch.int16: 2, torch.float32: 4}


def _guarded(shape, dtype):
    if isinstance(shape, int):
        shape = (shape,)
    shape = tuple(int(x) for x in shape)
    n = 1
    for x in shape:
        n *= x
    nbytes = n * _ITEMSIZE[dtype]
    if nbytes == 0:
        return _real_empty(shape, dtype=dtype)
    page = mmap.PAGESIZE
    body = (nbytes + page - 1) // page * page
    m = mmap.mmap(-1, body + page)
    addr = ctypes.addressof(ctypes.c_char.from_buffer(m))
    if _libc.mprotect(ctypes.c_void_p(addr + body), ctypes.c_size_t(page), 0) != 0:
        raise OSError(ctypes.get_errno(), "mprotect")
    start = body - (nbytes + 15) // 16 * 16        # 16-byte aligned start, at most 15 bytes of slack at the end
    return torch.frombuffer(m, dtype=dtype, count=n, offset=start).reshape(shape)


def _empty(*size, **kw):
    if kw.get("device", None) is _EMU_DEVICE and set(kw) <= {"device", "dtype"}:
        shape = size[0] if len(size) == 1 and not isinstance(size[0], int) else size
        return _guarded(shape, kw.get("dtype", torch.float32))
    return _real_empty(*size, **kw)


def _zeros(*size, **kw):
    if kw.get("device", None) is _EMU_DEVICE and set(kw) <= {"device", "dtype"}:
        shape = size[0] if len(size) == 1 and not isinstance(size[0], int) else size
        t = _guarded(shape, kw.get("dtype", torch.float32))
        t.zero_()
        return t
    return _real_zeros(*size, **kw)


class EmuEngine(engine_mod.HipEngine):
    """HipEngine with host tensors; every launch goes to the emulated kernels."""

    def __init__(self, mode="fast"):
        self.device = _EMU_DEVICE
        self.mode = _lib.VQ_MODE_FAST if mode == "fast" else _lib.VQ_MODE_EXACT
        self._ws = None
        self._scratch_n = -1
        self._labels_scratch = None
        self._dist_scratch = None

    def _stream(self):
        return ctypes.c_void_p(0)

    def _mark(self):
        return None

    def _read_bytes(self, buf, off, nbytes, mark):
        return buf[off: off + nbytes].numpy().copy()

    def synchronize(self):
        pass


@contextlib.contextmanager
def emulated_kernels():
    """Inside this context HipEngine-derived engines call the emulated library."""
    with mock.patch.object(engine_mod, "lib", emu_lib()), \
            mock.patch.object(torch.cuda, "device", lambda *_a, **_k: contextlib.nullcontext()), \
            mock.patch.object(torch, "empty", _empty), mock.patch.object(torch, "zeros", _zeros):
        yield
