"""A short run of tools/gpu_fuzz.py inside the GPU suite: random image shapes, block shapes, codebook sizes and
padding modes through every search path, the fused LBG pass, decode, label packing, the static-scratch entry
point and short training runs, each against the all-fp64 search of the same library (which the parity tests pin
to the CPU oracle).  Longer runs with other seeds are recorded in profiles/."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [5])
def test_random_shapes_for_twenty_seconds(seed):
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tools", "gpu_fuzz.py"), "20", str(seed)], cwd=ROOT,
                       capture_output=True, text=True, timeout=110)
    tail = r.stdout[-1500:] + r.stderr[-3000:]
    assert r.returncode == 0 and "fuzz ok" in r.stdout, tail


def test_random_small_images_against_the_cpu_oracle_for_fifteen_seconds():
    """tests/gpu_fuzz_oracle.py: block matrix, whole training runs, search, decode and packed labels of random small
    images against the CPU oracle."""
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tests", "gpu_fuzz_oracle.py"), "15", "9"], cwd=ROOT,
                       capture_output=True, text=True, timeout=110)
    tail = r.stdout[-1500:] + r.stderr[-3000:]
    assert r.returncode == 0 and "oracle fuzz ok" in r.stdout, tail
