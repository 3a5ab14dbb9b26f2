"""EncodePipeline (consecutive images on different HIP streams) against the same calls made one after the other on
one stream, and against the CPU oracle: labels and decoded pixels bit for bit, whatever lane an image ran on and
however often the lanes' buffers were reused."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _image(h, w, c, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 120 + 70 * np.sin(xx / (7.0 + seed)) * np.cos(yy / (11.0 + 2 * seed))
    img = base[..., None] + rng.normal(0, 9, (h, w, c)) + 10 * np.arange(c)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("lanes", [1, 3, 4])
def test_pipeline_equals_one_stream_and_the_oracle(lanes, oracle):
    from vector_quantization_amd import HipEngine, EncodePipeline, train_lbg
    dev = torch.device("cuda:0")
    eng = HipEngine(dev)
    shapes = [(256, 320, 1), (131, 77, 1), (512, 512, 1), (64, 64, 1), (300, 200, 1), (256, 320, 1), (1024, 256, 1),
              (90, 90, 1), (512, 512, 1)]
    imgs = [_image(h, w, c, i) for i, (h, w, c) in enumerate(shapes)]
    cb = train_lbg(eng, eng.blocks_from_image(torch.from_numpy(imgs[2]).to(dev), 4, 4), 64, log=None).codebook
    cb_h = cb.cpu().numpy()
    pipe = EncodePipeline(dev, lanes=lanes)
    got = []
    for im in imgs:
        r = pipe.submit(torch.from_numpy(im).to(dev), 4, 4, cb)
        pipe.wait(r)                                           # the copies below run on the current stream
        got.append((r.labels.clone(), r.decoded.clone(), r.lane))
    pipe.synchronize()
    assert sorted({g[2] for g in got}) == list(range(min(lanes, len(imgs))))
    for im, (lab, dec, _) in zip(imgs, got):
        t = torch.from_numpy(im).to(dev)
        ref_lab = eng.encode_image(t, 4, 4, cb)
        h, w, c = im.shape
        nr, nc = -(-h // 4), -(-w // 4)
        ref_dec = eng.decode(ref_lab.view(nr, nc), cb, 4, 4, c)
        assert torch.equal(lab, ref_lab)
        assert torch.equal(dec, ref_dec)
    # the oracle on two of them
    for i in (1, 4):
        blocks = oracle.blocks_u8(imgs[i], 4, 4)
        ref = oracle.encode_l1(blocks, cb_h)
        assert np.array_equal(got[i][0].cpu().numpy(), ref)
        h, w, c = imgs[i].shape
        nr, nc = -(-h // 4), -(-w // 4)
        assert np.array_equal(got[i][1].cpu().numpy(), oracle.decode_u8(ref.reshape(nr, nc), cb_h, 4, 4, c))


def test_pipeline_results_without_waiting_are_complete_after_synchronize():
    """Many submits in flight at once (no wait in between, buffers read only after synchronize): every lane's last
    result is that lane's last image."""
    from vector_quantization_amd import HipEngine, EncodePipeline, train_lbg
    dev = torch.device("cuda:0")
    eng = HipEngine(dev)
    imgs = [torch.from_numpy(_image(512, 512, 3, 20 + i)).to(dev) for i in range(8)]
    cb = train_lbg(eng, eng.blocks_from_image(imgs[0], 4, 4), 256, log=None).codebook
    pipe = EncodePipeline(dev, lanes=4)
    res = [pipe.submit(im, 4, 4, cb) for im in imgs]
    pipe.synchronize()
    for i in range(4, 8):                                      # the last image of each lane
        ref = eng.encode_image(imgs[i], 4, 4, cb)
        assert torch.equal(res[i].labels, ref)
        assert torch.equal(res[i].decoded, eng.decode(ref.view(128, 128), cb, 4, 4, 3))


def test_result_read_after_wait_survives_the_lanes_next_image():
    """wait(result) followed by a read on the caller's stream, while that stream is held up behind a long image on
    another lane; the result's lane meanwhile gets its next image (same shape, so the same buffers) with
    inputs_ready=True.  The lane must not overwrite the labels before the pending read has happened."""
    from vector_quantization_amd import HipEngine, EncodePipeline
    dev = torch.device("cuda:0")
    eng = HipEngine(dev)
    rng = np.random.default_rng(77)
    big = torch.from_numpy(_image(2048, 2048, 1, 3)).to(dev).repeat(8, 8, 1).contiguous()    # 16384 x 16384
    x = torch.from_numpy(_image(96, 96, 1, 4)).to(dev)
    y = torch.from_numpy(_image(96, 96, 1, 5)).to(dev)
    cb_big = torch.from_numpy(rng.random((1024, 16)) * 255.0).to(dev)
    cb = torch.from_numpy(rng.random((64, 16)) * 255.0).to(dev)
    want_x = eng.encode_image(x, 4, 4, cb).clone()
    want_y = eng.encode_image(y, 4, 4, cb).clone()
    assert not torch.equal(want_x, want_y)
    torch.cuda.synchronize(dev)
    for _ in range(3):
        pipe = EncodePipeline(dev, lanes=2)
        r_big = pipe.submit(big, 4, 4, cb_big, inputs_ready=True)       # lane 0: more than a millisecond
        r_x = pipe.submit(x, 4, 4, cb, inputs_ready=True)               # lane 1: done long before
        pipe.wait(r_big)
        keep = r_big.labels.clone()                                     # the caller's stream now waits for lane 0
        pipe.wait(r_x)
        got_x = r_x.labels.clone()                                      # ... and reads lane 1's labels after that
        pipe.submit(big, 4, 4, cb_big, inputs_ready=True)               # lane 0 again
        r_y = pipe.submit(y, 4, 4, cb, inputs_ready=True)               # lane 1 again: same buffers as x
        pipe.wait(r_y)
        got_y = r_y.labels.clone()
        pipe.synchronize()
        torch.cuda.synchronize(dev)
        assert torch.equal(got_x, want_x) and torch.equal(got_y, want_y) and keep.numel() == 4096 * 4096
