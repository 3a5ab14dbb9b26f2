"""Host-side sanitizers over the CPU builds (no GPU code is ever built with a sanitizer).

* oracle/sanitize_main.c drives every function of the CPU oracle with exactly sized buffers; built with
  -fsanitize=address,undefined it must run clean and print the checksum the plain build prints.
* the kernels compiled for the host (tests/emu) are built with -fsanitize=undefined (aborting on the first finding)
  and a subset of the emulated kernel tests is run on that build in a child process.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_and_ubsan():
    odir = os.path.join(ROOT, "oracle")
    subprocess.run(["make", "-C", odir, "sanitize_check", "sanitize_plain"], check=True, capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:verify_asan_link_order=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    plain = subprocess.run([os.path.join(odir, "sanitize_plain")], check=True, capture_output=True, text=True)
    san = subprocess.run([os.path.join(odir, "sanitize_check")], capture_output=True, text=True, env=env)
    assert san.returncode == 0, san.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in san.stderr and "runtime error" not in san.stderr, san.stderr[-4000:]
    assert san.stdout.strip() == plain.stdout.strip() and len(plain.stdout.strip()) == 16


def test_emulated_kernels_under_ubsan():
    env = dict(os.environ, VQ_EMU_SANITIZE="1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_emu_kernels.py"),
           "-k", "blocks or decode or pack or sort_image or encode_pruned or lbg_iteration"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-3000:] + out.stderr[-3000:])
    assert "runtime error" not in out.stderr and "runtime error" not in out.stdout
    assert " passed" in out.stdout
