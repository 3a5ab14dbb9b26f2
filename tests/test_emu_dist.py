"""The sharded LBG loop with the package's engine on emulated kernels (tests/emu): two gloo ranks, each
holding a shard, against the single-process run and the oracle.  The GPU twin of this test is
test_gpu_dist.py; this one runs without a GPU and covers the same host code (lbg.py's exchange fed
by the real kernels' per-chunk distance sums)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make_blocks():
    rng = np.random.default_rng(5)
    base = np.clip(rng.random((40, 1)) * 255 + rng.normal(0, 20, (40, 16)), 0, 255)
    idx = rng.integers(0, 40, size=8192 + 900)
    return np.clip(base[idx] + rng.normal(0, 8, (len(idx), 16)), 0, 255).astype(np.uint8)


def _worker(rank, world, port, cut, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from emu_engine import EmuEngine, emulated_kernels
        from vector_quantization_amd import train_lbg
        blocks = _make_blocks()
        bounds = [0, cut, len(blocks)]
        with emulated_kernels():
            shard = torch.from_numpy(blocks[bounds[rank]:bounds[rank + 1]].copy())
            res = train_lbg(EmuEngine(), shard, 32, max_iterations=3, group=dist.group.WORLD, log=None)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), codebook=res.codebook.numpy(),
                 iters=np.array(res.iters_per_stage), dist=np.array(res.distortions))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cut", [8192])
def test_two_emulated_ranks_equal_one(tmp_path, cut, oracle):
    blocks = _make_blocks()
    ref = oracle.lbg_train(blocks, 32, max_iter=3)      # three iterations per stage keep the emulation short
    mp.spawn(_worker, args=(2, _free_port(), cut, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["codebook"], r1["codebook"])
    assert np.array_equal(r0["codebook"], ref["codebook"])              # integer sums: identical to one process
    assert list(r0["iters"]) == list(ref["iters_per_stage"])
    if cut % 8192 == 0:
        assert np.array_equal(r0["dist"], ref["distortions"])           # chunk-aligned shards: numpy's order kept
    else:
        assert np.allclose(r0["dist"], ref["distortions"], rtol=1e-12, atol=0)
