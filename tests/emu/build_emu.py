#!/usr/bin/env python3
"""Builds tests/emu/_build/libvqemu.so: the kernels and the C ABI of
vector_quantization_amd/csrc/vq_hip.hip compiled for the host against tests/emu/hip/hip_runtime.h.
TEST INFRASTRUCTURE ONLY.  The source is used as it is, apart from a few textual substitutions a
host compiler needs (each checked to match, so that a change of the source is noticed here):
dynamic LDS declarations become pointers into the emulated LDS, and the three kinds of inline
gfx950 assembly become their C++ meaning (the LDS wait of a lockstep wave becomes a wave rendezvous)."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, "vector_quantization_amd", "csrc", "vq_hip.hip")
OUT_DIR = os.path.join(HERE, "_build")
# VQ_EMU_SANITIZE=1: the same build with -fsanitize=undefined (host code only), aborting on the first finding
SANITIZE = os.environ.get("VQ_EMU_SANITIZE") == "1"
OUT = os.path.join(OUT_DIR, "libvqemu_ubsan.so" if SANITIZE else "libvqemu.so")

RUNTIME = r'''
namespace emu {
Block *g_block = nullptr;
unsigned char g_lds[LDS_BYTES] __attribute__((aligned(256)));
// compute units the emulated device reports (VQ_EMU_CUS overrides: launches then take the forms they have on a large chip)
int g_cus = std::getenv("VQ_EMU_CUS") ? std::atoi(std::getenv("VQ_EMU_CUS")) : 4;
}
thread_local emu::Idx threadIdx, blockIdx, blockDim, gridDim;
'''


def transform(text):
    n_total = 0

    def sub(pattern, repl, text, expect_min, flags=0):
        nonlocal n_total
        out, n = re.subn(pattern, repl, text, flags=flags)
        if n < expect_min:
            raise SystemExit(f"emu build: pattern {pattern!r} matched {n} times, expected at least {expect_min}")
        n_total += n
        return out

    # extern __shared__ [__align__(16)] T name[];  ->  T *name = (T *)emulated LDS
    text = sub(r"extern __shared__ (?:__align__\(16\) )?([A-Za-z_][A-Za-z0-9_ ]*?) (s_[a-z0-9_A-Z]+)\[\];",
               r"\1 *\2 = reinterpret_cast<\1 *>(emu::g_lds);", text, 7)
    # v_med3_u32: the median of three
    text = sub(r'asm\("v_med3_u32 %0, %1, %2, %3" : "=v"\(r\) : "v"\(a\), "v"\(b\), "v"\(c\)\);',
               "r = std::max(std::min(a, b), std::min(std::max(a, b), c));", text, 1)
    # scalar loads issued ahead of their use: the source carries its own host form next to the gfx950 one
    text = sub(r"#ifndef VQ_HOST_EMULATION\n.*?\n#else\n(.*?)\n#endif  // VQ_HOST_EMULATION\n", r"\1\n", text, 1, flags=re.S)
    # register-class hints and LDS waits have no host meaning
    text = sub(r'asm volatile\("" : "\+v"\(([a-z0-9_]+)\)\);', r"(void)\1;", text, 2)
    # "wait for my LDS operations" orders LDS traffic between the lanes of a wave that runs in lockstep;
    # the emulated lanes are free-running threads, so it becomes a rendezvous of the wave
    text = sub(r'asm volatile\("s_waitcnt lgkmcnt\(0\)" ::: "memory"\);', "emu::wave().bar.wait();", text, 2)
    if "asm" in re.sub(r"//.*", "", text).replace("hipMemsetAsync", ""):
        raise SystemExit("emu build: inline assembly this script does not know")
    text = text.replace('#include "vq_hip.h"', '#include "vq_hip.h"\n' + RUNTIME, 1)
    return text


def _fresh(deps):
    return os.path.isfile(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(p) for p in deps)


def build(force=False):
    """Several test processes may ask for the library at once (spawned ranks): the build runs under a file
    lock and the library appears under its name only when it is complete."""
    import fcntl
    os.makedirs(OUT_DIR, exist_ok=True)
    deps = [SRC, os.path.join(HERE, "hip", "hip_runtime.h"), os.path.abspath(__file__)]
    if not force and _fresh(deps):
        return OUT
    with open(os.path.join(OUT_DIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and _fresh(deps):          # another process built it while this one waited
            return OUT
        cpp = os.path.join(OUT_DIR, "vq_hip_emu_ubsan.cpp" if SANITIZE else "vq_hip_emu.cpp")
        with open(cpp, "w") as f:
            f.write(transform(open(SRC).read()))
        tmp = OUT + f".{os.getpid()}.tmp"
        cmd = ["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-pthread", "-w",
               "-I", HERE, "-I", os.path.join(ROOT, "include"), cpp, "-o", tmp]
        if SANITIZE:
            # (alignment is left out: the kernels read rows as words only where the host has checked the alignment the
            # GPU needs, which is weaker than what the host compiler assumes for uint4)
            cmd[1:1] = ["-fsanitize=undefined", "-fno-sanitize=alignment", "-fno-sanitize-recover=undefined"]
        subprocess.run(cmd, check=True)
        os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
