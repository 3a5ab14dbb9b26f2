"""ctypes binding of libvqhip.so (the C ABI declared in include/vq_hip.h).

The library is the only compute backend of this package: if it cannot be loaded the import
fails loudly -- there is no CPU fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

VQ_OK = 0
VQ_E_INVALID = -1
VQ_E_HIP = -2
VQ_E_WORKSPACE = -3
VQ_E_UNSUPPORTED = -4

VQ_MODE_EXACT = 0
VQ_MODE_FAST = 1

VQ_PAD_EDGE = 0
VQ_PAD_ZERO = 1

ABI_VERSION = 8

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_sz = ctypes.c_size_t
_dbl = ctypes.c_double

# name -> (restype, argtypes); one entry per declaration in include/vq_hip.h
SIGNATURES = {
    "vq_abi_version": (_i, []),
    "vq_last_error": (ctypes.c_char_p, []),
    "vq_workspace_bytes": (_sz, [_i, _i, _i64]),
    "vq_blocks_u8": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "vq_colsum_u8": (_i, [_vp, _i64, _i, _vp, _vp]),
    "vq_encode_l1": (_i, [_vp, _i64, _i, _vp, _i, _vp, _vp, _i, _vp, _sz, _vp]),
    "vq_static_scratch_bytes": (_sz, []),
    "vq_encode_l1_static": (_i, [_vp, _i64, _i, _vp, _i, _vp, _vp, _i, _vp]),
    "vq_codebook_prepare": (_i, [_vp, _i, _i, _vp, _sz, _vp]),
    "vq_sorted_bytes": (_sz, [_i, _i64]),
    "vq_prune_supported": (_i, [_i, _i]),
    "vq_prune_resolves_in_filter": (_i, [_i, _i]),
    "vq_blocks_sort": (_i, [_vp, _i64, _i, _vp, _sz, _vp]),
    "vq_blocks_sort_img": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "vq_encode_l1_img": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "vq_encode_l1_prepared": (_i, [_vp, _vp, _sz, _i64, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "vq_encode_l1_prepared_phases": (_i, [_vp, _vp, _sz, _i64, _i, _vp, _i, _vp, _vp, _i, _vp, _sz, _vp]),
    "vq_workspace_counters_offset": (_sz, [_i, _i, _i64]),
    "vq_profile_next_search": (None, [_vp, _vp]),
    "vq_lbg_accum": (_i, [_vp, _vp, _sz, _i64, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "vq_lbg_pass_fused": (_i, [_i, _i]),
    "vq_lbg_state_bytes": (_sz, [_i]),
    "vq_lbg_stage_begin": (_i, [_vp, _sz, _i, _vp]),
    "vq_lbg_iterate": (_i, [_vp, _vp, _sz, _i64, _i, _vp, _i, _i64, _dbl, _i, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp, _sz, _vp]),
    "vq_lbg_iterate_local": (_i, [_vp, _vp, _sz, _i64, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp, _sz, _vp]),
    "vq_lbg_iterate_finish": (_i, [_vp, _vp, _i, _i64, _vp, _i, _i64, _vp, _vp, _i, _i, _i64, _dbl, _i, _vp, _sz, _vp, _sz,
                                   _vp]),
    "vq_sorted_forget": (_i, [_vp]),
    "vq_numpy_sum_f64": (_i, [_vp, _i64, _vp, _vp, _vp]),
    "vq_accum_labels": (_i, [_vp, _i64, _i, _vp, _i, _vp, _vp, _vp]),
    "vq_lbg_finalize": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "vq_lbg_split": (_i, [_vp, _i, _i, _dbl, _vp, _vp]),
    "vq_decode_u8": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "vq_packed_bytes": (_sz, [_i64, _i]),
    "vq_pack_labels": (_i, [_vp, _i64, _i, _vp, _vp]),
    "vq_unpack_labels": (_i, [_vp, _i64, _i, _vp, _vp]),
}


def library_path():
    """Path of libvqhip.so: $VQ_HIP_LIB if set, else the in-tree build."""
    env = os.environ.get("VQ_HIP_LIB")
    if env:
        return env
    return os.path.join(_HERE, "lib", "libvqhip.so")


class VQLibraryError(RuntimeError):
    pass


def _load():
    path = library_path()
    if not os.path.isfile(path):
        raise VQLibraryError(
            f"libvqhip.so not found at {path}. Build it with vector_quantization_amd/csrc/build.sh "
            f"(or python -c 'import __graft_entry__ as g; g.build()'); this package has no CPU fallback."
        )
    try:
        lib = ctypes.CDLL(path)
    except OSError as exc:  # pragma: no cover - depends on the machine
        raise VQLibraryError(f"cannot load {path}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise VQLibraryError(f"{path} does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    got = lib.vq_abi_version()
    if got != ABI_VERSION:
        raise VQLibraryError(f"{path} has ABI version {got}, this package expects {ABI_VERSION}")
    return lib


lib = _load()


def last_error():
    msg = lib.vq_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, what):
    """Map a C status code to the Python exception the reference would raise for that condition."""
    if rc == VQ_OK:
        return
    msg = f"{what}: {last_error()}"
    if rc == VQ_E_INVALID:
        raise ValueError(msg)
    if rc == VQ_E_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)
