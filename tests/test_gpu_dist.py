"""The sharded LBG loop on the real HIP engine: two ranks (gloo rendezvous, both on cuda:0) each
hold a shard of the block vectors; the exchange is the one bench.py and multi-GPU jobs use
(one all-gather of integer sums/counts and per-chunk distance sums per iteration).  The result must be
bit-identical to the single-GPU run, including every distortion when the shards are 8192-aligned."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make_blocks():
    rng = np.random.default_rng(21)
    base = rng.integers(0, 256, size=(40, 16))
    idx = rng.integers(0, 40, size=3 * 8192 + 517)
    noise = rng.integers(-6, 7, size=(len(idx), 16))
    return np.clip(base[idx] + noise, 0, 255).astype(np.uint8)


def _worker(rank, world, port, cut, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vector_quantization_amd import HipEngine, train_lbg
        blocks = _make_blocks()
        bounds = [0, cut, len(blocks)]
        eng = HipEngine("cuda:0")
        shard = torch.from_numpy(blocks[bounds[rank]:bounds[rank + 1]].copy()).to("cuda:0")
        res = train_lbg(eng, shard, 32, group=dist.group.WORLD, log=None)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), codebook=res.codebook.cpu().numpy(),
                 iters=np.array(res.iters_per_stage), dist=np.array(res.distortions))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cut", [8192, 2 * 8192, 10000])
def test_two_ranks_equal_one_gpu(tmp_path, cut, oracle):
    from vector_quantization_amd import HipEngine, train_lbg
    blocks = _make_blocks()
    single = train_lbg(HipEngine("cuda:0"), torch.from_numpy(blocks).to("cuda:0"), 32, log=None)
    ref = oracle.lbg_train(blocks, 32)
    assert np.array_equal(single.codebook.cpu().numpy(), ref["codebook"])
    assert np.array_equal(np.array(single.distortions), ref["distortions"])
    mp.spawn(_worker, args=(2, _free_port(), cut, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["codebook"], r1["codebook"])
    assert np.array_equal(r0["codebook"], single.codebook.cpu().numpy())
    assert list(r0["iters"]) == single.iters_per_stage
    if cut % 8192 == 0:
        assert np.array_equal(r0["dist"], np.array(single.distortions))
    else:
        assert np.allclose(r0["dist"], np.array(single.distortions), rtol=1e-12, atol=0)


def _nccl_worker(rank, world, port, cut, out_dir, allreduce_from):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from vector_quantization_amd import HipEngine, train_lbg
        import vector_quantization_amd.lbg as lbg_mod
        if allreduce_from is not None:
            lbg_mod.ALLREDUCE_FROM_BYTES = allreduce_from
        blocks = _make_blocks()
        bounds = [0, cut, len(blocks)]
        eng = HipEngine(f"cuda:{rank}")
        shard = torch.from_numpy(blocks[bounds[rank]:bounds[rank + 1]].copy()).to(f"cuda:{rank}")
        res = train_lbg(eng, shard, 32, group=dist.group.WORLD, log=None)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), codebook=res.codebook.cpu().numpy(),
                 iters=np.array(res.iters_per_stage), dist=np.array(res.distortions))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("allreduce_from", [None, 0])
def test_two_gpus_rccl_equal_one_gpu(tmp_path, allreduce_from):
    """The same sharded loop with one rank per GPU over RCCL (backend "nccl"): needs two GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from vector_quantization_amd import HipEngine, train_lbg
    blocks = _make_blocks()
    single = train_lbg(HipEngine("cuda:0"), torch.from_numpy(blocks).to("cuda:0"), 32, log=None)
    mp.spawn(_nccl_worker, args=(2, _free_port(), 2 * 8192, str(tmp_path), allreduce_from), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["codebook"], r1["codebook"])
    assert np.array_equal(r0["codebook"], single.codebook.cpu().numpy())
    assert list(r0["iters"]) == single.iters_per_stage
    assert np.array_equal(r0["dist"], np.array(single.distortions))


def _nccl_one_gpu_worker(rank, world, port, out_dir):
    """RCCL with a single rank: the collectives of the exchange run on device tensors through the nccl
    backend (world size 1), so that code path executes on a one-GPU box too."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        from vector_quantization_amd import HipEngine, train_lbg
        import vector_quantization_amd.lbg as lbg_mod
        blocks = torch.from_numpy(_make_blocks()).to("cuda:0")
        eng = HipEngine("cuda:0")
        out = {}
        for name, thr in (("gather", 1 << 40), ("allreduce", 0)):
            lbg_mod.ALLREDUCE_FROM_BYTES = thr
            res = train_lbg(eng, blocks, 32, group=dist.group.WORLD, log=None)
            out[name + "_cb"] = res.codebook.cpu().numpy()
            out[name + "_dist"] = np.array(res.distortions)
        np.savez(os.path.join(out_dir, "nccl1.npz"), **out)
    finally:
        dist.destroy_process_group()


def test_rccl_exchange_world_size_one(tmp_path, oracle):
    blocks = _make_blocks()
    ref = oracle.lbg_train(blocks, 32)
    mp.spawn(_nccl_one_gpu_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "nccl1.npz")
    for name in ("gather", "allreduce"):
        assert np.array_equal(r[name + "_cb"], ref["codebook"])
        assert np.array_equal(r[name + "_dist"], ref["distortions"])


def _batch_images(n_images=16, side=512):
    """A batch of grayscale images (config 4 of the baseline in miniature: many images, one codebook)."""
    rng = np.random.default_rng(404)
    yy, xx = np.mgrid[0:side, 0:side]
    imgs = []
    for i in range(n_images):
        base = 96 + 64 * np.sin(xx / (9.0 + i)) * np.cos(yy / (13.0 + 2 * i)) + rng.normal(0, 6, (side, side))
        imgs.append(np.clip(base + 8 * i, 0, 255).astype(np.uint8))
    return imgs


def _batch_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vector_quantization_amd import HipEngine, train_lbg
        eng = HipEngine("cuda:0")
        imgs = _batch_images()
        mine = imgs[rank * len(imgs) // world:(rank + 1) * len(imgs) // world]       # the batch is sharded by image
        dev = [torch.from_numpy(im).to("cuda:0") for im in mine]
        blocks = torch.cat([eng.blocks_from_image(im, 4, 4) for im in dev])
        res = train_lbg(eng, blocks, 64, group=dist.group.WORLD, log=None)
        labels = [eng.encode_image(im, 4, 4, res.codebook).cpu().numpy() for im in dev]
        np.savez(os.path.join(out_dir, f"batch{rank}.npz"), codebook=res.codebook.cpu().numpy(),
                 dist=np.array(res.distortions), labels=np.stack(labels))
    finally:
        dist.destroy_process_group()


def test_batch_of_images_sharded_over_two_ranks(tmp_path):
    """Images of a batch go to different ranks, the codebook is trained over all of them: codebook, every
    distortion and every image's labels equal the single-process run on the whole batch."""
    from vector_quantization_amd import HipEngine, train_lbg
    eng = HipEngine("cuda:0")
    imgs = _batch_images()
    dev = [torch.from_numpy(im).to("cuda:0") for im in imgs]
    blocks = torch.cat([eng.blocks_from_image(im, 4, 4) for im in dev])
    single = train_lbg(eng, blocks, 64, log=None)
    want = np.stack([eng.encode_image(im, 4, 4, single.codebook).cpu().numpy() for im in dev])
    mp.spawn(_batch_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "batch0.npz"), np.load(tmp_path / "batch1.npz")
    assert np.array_equal(r0["codebook"], single.codebook.cpu().numpy())
    assert np.array_equal(r1["codebook"], r0["codebook"])
    assert np.array_equal(r0["dist"], np.array(single.distortions))          # 16384 vectors per image: aligned shards
    assert np.array_equal(np.concatenate([r0["labels"], r1["labels"]]), want)
    # decoded images of the batch through the plain and the sharded labels are the same pixels
    dec = eng.decode(torch.from_numpy(want[3]).to("cuda:0").view(128, 128), single.codebook, 4, 4, 1).cpu().numpy()
    dec2 = eng.decode(torch.from_numpy(r0["labels"][3]).to("cuda:0").view(128, 128), single.codebook, 4, 4, 1).cpu().numpy()
    assert np.array_equal(dec, dec2)
