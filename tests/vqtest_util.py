"""Helpers shared by the test modules (golden fixture loading)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

TRAIN_CASES = [
    "luffy_2x2_k16", "man_4x4_k256", "man_2x2_k5", "luffy_4x4_k32", "luffy_8x8_k8",
    "synth_gray_zero_sat_4x4_k256", "synth_gray_250x246_4x4_k64", "man_2x2_k1", "synth_iid_128_4x4_k32",
    "synth_gray_1024_4x4_k1024",
]



def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}



def stage_end_codebooks(g):
    """Split the flat stage_end_codebooks array of a golden file into per-stage (K, d) arrays."""
    d = g["blocks"].shape[1]
    out, pos = [], 0
    for K in g["stage_K"]:
        n = int(K) * d
        out.append(g["stage_end_codebooks"][pos:pos + n].reshape(int(K), d))
        pos += n
    return out
