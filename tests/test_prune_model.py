"""A numpy model of the pruned search the HIP kernels implement (fixed-point rows sorted by element
sum, threshold -> position table, window-sorted vectors, per-wave range that widens in rounds),
checked against the oracle: the element-sum bound never drops the nearest codeword or a near-tie.
The model follows the kernel's integer arithmetic step by step, so it also pins the table rounding
rules (conservative on both sides).  CPU only; the kernels themselves are checked in
test_gpu_parity.py."""
import numpy as np
import pytest

SORT_WIN, WAVE_VECS, G, NT = 8192, 256, 16, 1024


def prepare(cb, clim=65535):
    K, d = cb.shape
    cmax = int(np.ceil(max(cb.max(), 0.0)))
    top = max(cmax, 255)
    s = 8
    while s > 0 and (top << s) > clim:
        s -= 1
    t = cb * float(1 << s)
    q = np.clip(np.rint(t), 0, clim)
    err = np.abs(t - q).sum(1)
    emax = int(err.max()) + 2
    C = q.astype(np.int64)
    sc = C.sum(1)
    order = np.lexsort((np.arange(K), sc))                # ascending (element sum, index)
    mtab = d * (255 << s) + 1
    tshift = 0
    while (mtab >> tshift) > NT - 1:
        tshift += 1
    nt = (mtab >> tshift) + 1
    scs = sc[order]
    postab = np.array([np.searchsorted(scs, t_ << tshift, side="left") for t_ in range(nt + 2)], dtype=np.int64)
    postab[nt + 1] = K
    return dict(s=s, emax=emax, C=C[order], orig=order, postab=postab, tshift=tshift, nt=nt, K=K)


def pruned_search(blocks, cb):
    """Returns (labels or -1 where the filter is ambiguous, evaluated codewords per wave)."""
    P = prepare(cb)
    N, d = blocks.shape
    K, s, window = P["K"], P["s"], 2 * P["emax"]
    kpad = (K + 7) & ~7
    Cp = np.zeros((kpad, d), dtype=np.int64)
    Cp[:K] = P["C"]
    labels = np.full(N, -2, dtype=np.int64)
    evaluated = []
    X_all = blocks.astype(np.int64) << s
    for w0 in range(0, N, SORT_WIN):
        idx = np.arange(w0, min(N, w0 + SORT_WIN))
        idx = idx[np.argsort(blocks[idx].astype(np.int64).sum(1) >> 1, kind="stable")]     # bucketed element sum
        for g0 in range(0, len(idx), WAVE_VECS):
            g = idx[g0:g0 + WAVE_VECS]
            X = X_all[g]
            sx = X.sum(1)
            sxlo, sxhi = int(sx.min()), int(sx.max())
            m1 = np.full(len(g), np.iinfo(np.int64).max)
            m2 = m1.copy()
            b1 = np.zeros(len(g), dtype=np.int64)

            def run(lo, hi):
                nonlocal m1, m2, b1
                for k in range(lo, hi):
                    S = np.abs(X - Cp[k]).sum(1) if k < K else np.full(len(g), 1 << 40)
                    key = S * 4096 + (P["orig"][k] if k < K else 4095)                  # distance first, then index
                    m2 = np.where(key < m1, m1, np.minimum(m2, key))
                    b1 = np.where(key < m1, k, b1)
                    m1 = np.minimum(m1, key)

            pa = int(P["postab"][((sxlo >> 1) + (sxhi >> 1)) >> P["tshift"]] & ~7) - G // 2
            pa = max(0, min(pa, kpad - G))
            pb = min(pa + G, kpad)
            segs = [(pa, pb)]
            while True:
                for lo, hi in segs:
                    run(lo, hi)
                bm = int((m1 // 4096).max())
                thr = bm + window
                lo_sum = max(sxlo - thr, 0)
                ih = min(((sxhi + thr) >> P["tshift"]) + 1, P["nt"] + 1)
                klo = int(P["postab"][lo_sum >> P["tshift"]] & ~7)
                khi = int((P["postab"][ih] + 7) & ~7)
                na = max(klo, pa - G) if pa > klo else pa
                nb = min(khi, pb + G) if pb < khi else pb
                if na == pa and nb == pb:
                    break
                segs = [(na, pa), (pb, nb)]
                pa, pb = na, nb
            evaluated.append(pb - pa)
            s1, s2 = m1 // 4096, m2 // 4096
            amb = (s2 - s1) <= window
            labels[g] = np.where(amb, -1, m1 % 4096)
    return labels, np.array(evaluated), P


def _structured(rng, K, d, N, spread, noise):
    cb = np.clip(rng.random((K, 1)) * spread + rng.normal(0, 20, (K, d)) + (255 - spread) / 2, 0, 300)
    blocks = np.clip(cb[rng.integers(0, K, N)] + rng.normal(0, noise, (N, d)), 0, 255).astype(np.uint8)
    return blocks, cb


@pytest.mark.parametrize("K,d,N,spread,noise", [(64, 16, 3000, 255, 8), (256, 16, 9000, 255, 12), (100, 4, 2000, 200, 5),
                                                 (64, 16, 2000, 0, 40), (40, 12, 1500, 120, 25)])
def test_pruned_model_matches_oracle(oracle, K, d, N, spread, noise):
    rng = np.random.default_rng(K * 31 + d)
    blocks, cb = _structured(rng, K, d, N, spread, noise)
    want, dist = oracle.encode_l1(blocks, cb, want_dist=True)
    got, evaluated, P = pruned_search(blocks, cb)
    settled = got >= 0
    assert settled.mean() > 0.9                                   # the filter decides almost everything itself
    assert np.array_equal(got[settled], want[settled])            # and never differently from the fp64 search
    # the vectors it leaves to the fp64 re-check really are near-ties in the scaled distance
    D = np.abs(blocks[~settled].astype(np.float64)[:, None, :] - cb[None, :, :]).sum(2)
    two = np.sort(D, axis=1)[:, :2]
    assert np.all((two[:, 1] - two[:, 0]) * (1 << P["s"]) <= 4 * P["emax"])
    if spread >= 200:
        assert evaluated.mean() < 0.8 * K                         # the bound really skips codewords here


def test_pruned_model_degrades_to_full_range(oracle):
    """No spread of element sums at all: every wave must end up searching the whole codebook."""
    rng = np.random.default_rng(5)
    K, d, N = 64, 16, 1500
    blocks = rng.integers(0, 256, size=(N, d), dtype=np.uint8)
    cb = rng.random((K, d)) * 255.0
    cb *= (127.5 * d / cb.sum(1))[:, None]
    want = oracle.encode_l1(blocks, cb)
    got, evaluated, _ = pruned_search(blocks, cb)
    assert np.array_equal(got[got >= 0], want[got >= 0])
    assert evaluated.min() >= K


def test_element_sum_bound_is_exact_in_integers():
    rng = np.random.default_rng(11)
    X = rng.integers(0, 65536, size=(2000, 16))
    C = rng.integers(0, 65536, size=(2000, 16))
    assert np.all(np.abs(X - C).sum(1) >= np.abs(X.sum(1) - C.sum(1)))
