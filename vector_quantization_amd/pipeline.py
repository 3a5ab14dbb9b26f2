"""A stream of images through the encode (+ decode) path with consecutive images on different HIP streams.

One image's pass is a chain of kernels with different bottlenecks: the window sort and the decode move bytes
(HBM-bound), the codebook preparation is one workgroup (latency-bound), the search is bound by the vector ALU and
ends with a stretch in which its last workgroups leave most of the chip idle.  Issued back to back on one stream they
add up; issued for consecutive images on different streams, the byte movers and the start of the next search run in
the slots the current search leaves free.  Every lane owns its engine (workspace), sorted buffer, labels and output
image, so nothing is shared between images in flight.  This is the batch form of VQ_Full.py:120-126 + :153-169
(compress + decompress of one image after the other with one codebook)."""
import collections

import torch

from ._lib import lib, VQ_PAD_EDGE
from .engine import HipEngine

PipelineResult = collections.namedtuple("PipelineResult", ["labels", "decoded", "lane"])
PipelineResult.__doc__ = """labels: (N,) int32 (the lane's buffer: valid until the lane is used again, `lanes` submits
later); decoded: (H', W', C) uint8 or None; lane: the stream it was enqueued on (EncodePipeline.wait / synchronize)."""


class EncodePipeline:
    """submit(img, bh, bw, cb) enqueues image -> window-sorted vectors -> prepared codebook -> search -> labels
    (-> decoded image) on the next of `lanes` streams and returns at once.  Results live in buffers owned by the lane:
    consume (or copy) them before submitting `lanes` more images; wait(result) makes the caller's current stream wait
    for them, synchronize() waits for everything.  (Measured at 8192 x 8192 gray, 4x4 blocks, K = 256: 0.168 ms per
    image on one stream, 0.150 with 4 lanes, 0.143 with 6 -- a process has 4 hardware queues, but which lanes share one
    depends on the streams created before, and a few more lanes even that out.)"""

    def __init__(self, device="cuda:0", lanes=6, mode="fast"):
        if lanes < 1:
            raise ValueError("lanes must be at least 1")
        self.device = torch.device(device)
        self.engines = [HipEngine(self.device, mode=mode) for _ in range(lanes)]
        self.device = self.engines[0].device
        self.streams = [torch.cuda.Stream(self.device) for _ in range(lanes)]
        self._scratch = [None] * lanes
        self._labels = [None] * lanes
        self._out = [None] * lanes
        self._readers = [None] * lanes        # per lane: the stream that wait() made a consumer of its last result
        self._next = 0

    @property
    def lanes(self):
        return len(self.engines)

    def submit(self, img, bh, bw, cb, pad_mode=VQ_PAD_EDGE, decode=True, inputs_ready=False):
        """inputs_ready: the image and the codebook are complete already (nothing pending on the caller's stream
        writes them), so the lane need not wait for the caller's stream -- one cross-stream dependency less per image."""
        if img.dim() == 2:
            img = img.unsqueeze(-1)
        H, W, C = img.shape
        K, d = cb.shape
        nr, nc = -(-H // bh), -(-W // bw)
        N = nr * nc
        k = self._next
        self._next = (k + 1) % len(self.engines)
        eng, stream = self.engines[k], self.streams[k]
        need = max(int(lib.vq_sorted_bytes(int(d), int(N))), N * d, 16)
        shape = (nr * bh, nc * bw, C)
        # the lane starts after whatever the caller has enqueued so far (the image, the codebook)
        if not inputs_ready:
            stream.wait_stream(torch.cuda.current_stream(self.device))
        # a stream that wait() handed this lane's previous result to may still be reading it: the lane's buffers are
        # overwritten only after what that stream has enqueued so far
        if self._readers[k] is not None:
            stream.wait_stream(self._readers[k])
            self._readers[k] = None
        with torch.cuda.stream(stream):
            # The lane's buffers are allocated with the lane's stream current: when an image of another shape replaces
            # one of them, the allocator hands the old block only to later work of this same stream, never to another
            # lane whose kernels could overlap with this lane's pending ones.
            if self._scratch[k] is None or self._scratch[k].numel() < need:
                self._scratch[k] = torch.empty(need, dtype=torch.uint8, device=self.device)
            if self._labels[k] is None or self._labels[k].numel() != N:
                self._labels[k] = torch.empty(N, dtype=torch.int32, device=self.device)
            if decode and (self._out[k] is None or tuple(self._out[k].shape) != shape):
                self._out[k] = torch.empty(shape, dtype=torch.uint8, device=self.device)
            labels = eng.encode_image(img, bh, bw, cb, pad_mode=pad_mode, scratch=self._scratch[k], labels=self._labels[k])
            decoded = eng.decode(labels.view(nr, nc), cb, bh, bw, C, out=self._out[k]) if decode else None
        return PipelineResult(labels, decoded, k)

    def wait(self, result):
        """The caller's current stream waits for this result (for everything enqueued on its lane so far; no event
        is recorded per image: an event record is a release fence in the middle of the lane's work).  The next submit
        on that lane in turn waits for what the caller's stream has enqueued until then -- reads of the result enqueued
        right after wait() are safe, also with inputs_ready=True; a consumer on any other stream has to order itself."""
        reader = torch.cuda.current_stream(self.device)
        reader.wait_stream(self.streams[result.lane])
        self._readers[result.lane] = reader

    def synchronize(self):
        for s in self.streams:
            s.synchronize()
