"""OracleEngine: the HipEngine interface served by the CPU oracle.

TEST INFRASTRUCTURE ONLY.  It lets the host logic of the package (the reference-shaped Codebook
class, the LBG control loop, the multi-rank exchange) run in CPU-only tests and under gloo; the
package itself never constructs it.
"""
import numpy as np
import torch

from oracle import vq_oracle as o


class OracleEngine:
    def __init__(self):
        self.device = torch.device("cpu")

    def synchronize(self):
        pass

    def blocks_from_image(self, img, bh, bw, pad_mode=0):
        return torch.from_numpy(o.blocks_u8(img.numpy(), bh, bw, pad_mode))

    def colsum(self, blocks):
        return torch.from_numpy(o.colsum_u8(blocks.numpy()).astype(np.int64))

    def sort_blocks(self, blocks, out=None):
        return None          # a layout of the device path only; the search result does not depend on it

    def encode(self, blocks, cb, want_dist=False, mode=None, sorted=None):
        labels, mind = o.encode_l1(blocks.numpy(), cb.numpy(), want_dist=True)
        if want_dist:
            return torch.from_numpy(labels), torch.from_numpy(mind)
        return torch.from_numpy(labels)

    def lbg_accum(self, blocks, cb, want_labels=False, mode=None, want_chunks=False, sorted=None):
        b = blocks.numpy()
        K, d = cb.shape
        N = b.shape[0]
        if N == 0:
            acc = torch.zeros(K * d + K, dtype=torch.int64)
            labels, mind = np.zeros(0, dtype=np.int32), np.zeros(0)
        else:
            labels, mind = o.encode_l1(b, cb.numpy(), want_dist=True)
            sums, counts = o.accumulate(b, labels, K)
            acc = torch.from_numpy(np.concatenate([sums.reshape(-1), counts]).astype(np.int64))
        dist = torch.tensor([o.numpy_sum(mind)], dtype=torch.float64)
        out = [acc, dist]
        if want_labels:
            out.append(torch.from_numpy(labels))
        if want_chunks:
            out.append(torch.tensor([o.pairwise_sum(mind[lo:lo + 8192]) for lo in range(0, N, 8192)],
                                    dtype=torch.float64))
        return tuple(out)

    def lbg_finalize(self, acc, cb_old):
        K, d = cb_old.shape
        a = acc.numpy().astype(np.uint64)
        return torch.from_numpy(o.finalize(a[: K * d].reshape(K, d), a[K * d:], cb_old.numpy()))

    def lbg_split(self, cb, eps):
        return torch.from_numpy(o.split(cb.numpy(), eps))

    def decode(self, labels, cb, bh, bw, C):
        return torch.from_numpy(o.decode_u8(labels.numpy(), cb.numpy().reshape(-1, bh * bw * C), bh, bw, C))

    def pack_labels(self, labels, K):
        return torch.from_numpy(o.pack_labels(labels.numpy(), K))
