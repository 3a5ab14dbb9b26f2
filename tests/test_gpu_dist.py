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
        # every stage must go through the device-resident form, and that form must not come back to the host: no
        # Tensor.cpu / item / tolist and no device-wide synchronisation between a stage's first enqueue and its last
        # record (the records are read through a side stream that waits on events only)
        calls = {"stage": 0, "host": 0}
        inner = eng.lbg_stage_group

        def counting(name, fn):
            def wrapped(*a, **k):
                if calls["stage_open"]:
                    calls["host"] += 1
                return fn(*a, **k)
            return wrapped

        def stage(*a, **k):
            calls["stage"] += 1
            calls["stage_open"] = True
            try:
                return inner(*a, **k)
            finally:
                calls["stage_open"] = False

        calls["stage_open"] = False
        eng.lbg_stage_group = stage
        saved = (torch.Tensor.cpu, torch.Tensor.item, torch.Tensor.tolist, torch.cuda.synchronize)
        torch.Tensor.cpu = counting("cpu", saved[0])
        torch.Tensor.item = counting("item", saved[1])
        torch.Tensor.tolist = counting("tolist", saved[2])
        torch.cuda.synchronize = counting("synchronize", saved[3])
        try:
            for name, thr in (("gather", 1 << 40), ("allreduce", 0)):
                lbg_mod.ALLREDUCE_FROM_BYTES = thr
                res = train_lbg(eng, blocks, 32, group=dist.group.WORLD, log=None)
                out[name + "_cb"] = res.codebook.cpu().numpy()
                out[name + "_dist"] = np.array(res.distortions)
        finally:
            torch.Tensor.cpu, torch.Tensor.item, torch.Tensor.tolist, torch.cuda.synchronize = saved
        out["stages"] = np.array(calls["stage"])
        out["host_calls"] = np.array(calls["host"])
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
    assert int(r["stages"]) == 2 * 5                    # K = 32: five split stages per training, all device-resident
    assert int(r["host_calls"]) == 0


def _p(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


@pytest.mark.parametrize("cuts", [(8192, 2 * 8192), (8192, 8192), (5000, 20000), (0, 3 * 8192)])
@pytest.mark.parametrize("K", [32, 64])
@pytest.mark.parametrize("layout", ["reduced", "gathered"])
def test_iterate_local_and_finish_equal_one_process(cuts, K, layout, oracle):
    """vq_lbg_iterate_local on three shards (one GPU, one after the other, each with its own workspace and state),
    the exchange done by hand, vq_lbg_iterate_finish: codebook and distortion record equal vq_lbg_iterate on all
    vectors -- bit for bit when the shards are 8192-aligned -- and the CPU oracle's iteration.  Covers an empty shard
    and shards that are not aligned.  layout "reduced": sums added up beforehand, chunk sums laid out rank by rank (an
    all-reduce plus an all-gather); "gathered": every rank's [sums | counts | chunk sums] written by the local call
    straight into its row of one buffer, which the finish call reads as it is (one all-gather)."""
    import ctypes
    from vector_quantization_amd import HipEngine
    from vector_quantization_amd._lib import lib, check
    dev = torch.device("cuda:0")
    eng = HipEngine(dev)
    blocks_h = _make_blocks()
    N, d = blocks_h.shape
    rng = np.random.default_rng(5)
    cb_h = np.clip(blocks_h[rng.choice(N, K, replace=False)].astype(np.float64) + rng.normal(0, 0.3, (K, d)), 0, 255)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    bounds = [0, cuts[0], cuts[1], N]
    sizes = [bounds[i + 1] - bounds[i] for i in range(3)]
    aligned = all(n % 8192 == 0 for n in [s for s in sizes if s > 0][:-1])

    def state_buf():
        n = lib.vq_lbg_state_bytes(4)
        st = torch.empty(n, dtype=torch.uint8, device=dev)
        check(lib.vq_lbg_stage_begin(_p(st), n, 4, stream), "stage_begin")
        return st, n

    # the single-process iteration
    blocks = torch.from_numpy(blocks_h).to(dev)
    srt = eng.sort_blocks(blocks)
    cb1 = torch.from_numpy(cb_h).to(dev)
    st1, nst = state_buf()
    acc1 = torch.empty(K * d + K, dtype=torch.int64, device=dev)
    dist1 = torch.empty(N, dtype=torch.float64, device=dev)
    ws1 = torch.empty(lib.vq_workspace_bytes(K, d, N), dtype=torch.uint8, device=dev)
    check(lib.vq_lbg_iterate(None, _p(srt), srt.numel(), N, d, _p(cb1), K, N, 0.001, 0, _p(acc1[:K * d]), _p(acc1[K * d:]),
                             None, _p(dist1), 1, _p(st1), nst, _p(ws1), ws1.numel(), stream), "vq_lbg_iterate")
    # the sharded iteration
    width = max(1, max(-(-n // 8192) for n in sizes)) if aligned else 1
    nacc = K * d + K
    gathered = torch.zeros((3, nacc + width), dtype=torch.int64, device=dev)
    scratch_chunks = torch.zeros(-(-N // 8192) + 1, dtype=torch.float64, device=dev)
    accs, tails = [], []
    for r in range(3):
        n = sizes[r]
        sh = blocks[bounds[r]:bounds[r + 1]].contiguous()
        srt_r = eng.sort_blocks(sh) if n else None
        cb_r = torch.from_numpy(cb_h).to(dev)
        st_r, _ = state_buf()
        acc = torch.empty(K * d + K, dtype=torch.int64, device=dev)
        chunk = torch.zeros(-(-n // 8192) + 1, dtype=torch.float64, device=dev)
        total = torch.zeros(1, dtype=torch.float64, device=dev)
        dist_r = torch.empty(max(n, 1), dtype=torch.float64, device=dev)
        lab_r = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        ws = torch.empty(lib.vq_workspace_bytes(K, d, n), dtype=torch.uint8, device=dev)
        if layout == "gathered":
            # the outputs go straight into this rank's row: [sums | counts | chunk sums or the local total]
            row = gathered[r]
            row_tail = row[nacc:].view(torch.float64)
            check(lib.vq_lbg_iterate_local(_p(sh) if n else None, _p(srt_r), srt_r.numel() if n else 0, n, d, _p(cb_r), K, 0,
                                           _p(row[:K * d]), _p(row[K * d:nacc]),
                                           _p(row_tail) if aligned else _p(scratch_chunks),
                                           None if aligned else _p(row_tail), _p(lab_r), _p(dist_r), 1,
                                           _p(st_r), nst, _p(ws), ws.numel(), stream), "vq_lbg_iterate_local")
            acc.copy_(row[:nacc])
            accs.append(acc)
            continue
        check(lib.vq_lbg_iterate_local(_p(sh) if n else None, _p(srt_r), srt_r.numel() if n else 0, n, d, _p(cb_r), K, 0,
                                       _p(acc[:K * d]), _p(acc[K * d:]), _p(chunk), _p(total), _p(lab_r), _p(dist_r), 1,
                                       _p(st_r), nst, _p(ws), ws.numel(), stream), "vq_lbg_iterate_local")
        accs.append(acc)
        t = torch.zeros(width, dtype=torch.float64, device=dev)
        if aligned:
            t[: -(-n // 8192)] = chunk[: -(-n // 8192)]
        else:
            t[0] = total[0]
        tails.append(t)
    acc_total = torch.stack(accs).sum(dim=0)
    counts = (ctypes.c_int * 3)(*[(-(-n // 8192) if aligned else 1) for n in sizes])
    cb2 = torch.from_numpy(cb_h).to(dev)
    st2, _ = state_buf()
    if layout == "gathered":
        stride = nacc + width
        check(lib.vq_lbg_iterate_finish(_p(gathered[0, :K * d]), _p(gathered[0, K * d:nacc]), 3, stride,
                                        _p(gathered[0, nacc:]), 3, stride, counts, _p(cb2), K, d, N, 0.001, 0, _p(st2), nst,
                                        None, 0, stream), "vq_lbg_iterate_finish")
    else:
        chunks_all = torch.stack(tails).contiguous()
        check(lib.vq_lbg_iterate_finish(_p(acc_total[:K * d]), _p(acc_total[K * d:]), 1, 0, _p(chunks_all), 3, width, counts,
                                        _p(cb2), K, d, N, 0.001, 0, _p(st2), nst, None, 0, stream), "vq_lbg_iterate_finish")
    torch.cuda.synchronize(dev)
    assert torch.equal(acc_total, acc1)
    assert torch.equal(cb2, cb1)
    rec1 = st1[64:80].cpu().numpy().view(np.float64)[0]
    rec2 = st2[64:80].cpu().numpy().view(np.float64)[0]
    if aligned:
        assert rec1 == rec2
    else:
        assert np.isclose(rec1, rec2, rtol=1e-12, atol=0)
    ref = oracle.lbg_iter(blocks_h, cb_h)
    assert np.array_equal(cb1.cpu().numpy(), ref["cb_out"]) and rec1 == ref["distortion"]


@pytest.mark.parametrize("K", [64, 256, 1024])
def test_iterate_local_finish_chain_prepares_the_next_iteration(K):
    """Three iterations as vq_lbg_iterate_local + vq_lbg_iterate_finish (one rank, the finish call given the
    workspace, so that it prepares the next iteration's codebook where the shape allows) against three calls of
    vq_lbg_iterate: codebooks and all records identical."""
    import ctypes
    from vector_quantization_amd import HipEngine
    from vector_quantization_amd._lib import lib, check
    dev = torch.device("cuda:0")
    eng = HipEngine(dev)
    blocks_h = _make_blocks()
    N, d = blocks_h.shape
    rng = np.random.default_rng(6)
    cb_h = np.clip(blocks_h[rng.choice(N, K, replace=False)].astype(np.float64) + rng.normal(0, 0.3, (K, d)), 0, 255)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    blocks = torch.from_numpy(blocks_h).to(dev)
    srt = eng.sort_blocks(blocks)
    nst = lib.vq_lbg_state_bytes(3)
    nacc = K * d + K
    nch = -(-N // 8192)
    out = []
    for split in (False, True):
        cb = torch.from_numpy(cb_h).to(dev)
        st = torch.empty(nst, dtype=torch.uint8, device=dev)
        check(lib.vq_lbg_stage_begin(_p(st), nst, 3, stream), "stage_begin")
        row = torch.zeros(nacc + nch, dtype=torch.int64, device=dev)
        dist_b = torch.empty(N, dtype=torch.float64, device=dev)
        lab = torch.empty(N, dtype=torch.int32, device=dev)
        ws = torch.empty(lib.vq_workspace_bytes(K, d, N), dtype=torch.uint8, device=dev)
        counts = (ctypes.c_int * 1)(nch)
        for it in range(3):
            if split:
                check(lib.vq_lbg_iterate_local(None, _p(srt), srt.numel(), N, d, _p(cb), K, it, _p(row[:K * d]),
                                               _p(row[K * d:nacc]), _p(row[nacc:]), None, _p(lab), _p(dist_b), 1, _p(st), nst,
                                               _p(ws), ws.numel(), stream), "vq_lbg_iterate_local")
                check(lib.vq_lbg_iterate_finish(_p(row[:K * d]), _p(row[K * d:nacc]), 1, 0, _p(row[nacc:]), 1, nch, counts,
                                                _p(cb), K, d, N, -1.0, it, _p(st), nst, _p(ws), ws.numel(), stream),
                      "vq_lbg_iterate_finish")
            else:
                check(lib.vq_lbg_iterate(None, _p(srt), srt.numel(), N, d, _p(cb), K, N, -1.0, it, _p(row[:K * d]),
                                         _p(row[K * d:nacc]), _p(lab), _p(dist_b), 1, _p(st), nst, _p(ws), ws.numel(), stream),
                      "vq_lbg_iterate")
        torch.cuda.synchronize(dev)
        out.append((cb.cpu().numpy(), st[64:64 + 48].cpu().numpy().tobytes()))
    assert np.array_equal(out[0][0], out[1][0])
    assert out[0][1] == out[1][1]


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
