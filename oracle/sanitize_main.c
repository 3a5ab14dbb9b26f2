/* Test infrastructure only: drives every function of the CPU oracle over seeded inputs -- ragged image sizes, both
 * pad modes, dimensions and codebook sizes that are not powers of two, duplicate codewords, out-of-range codewords
 * in decode, odd lengths in the numpy-order sum -- with every buffer allocated at its exact size, so that a build
 * with -fsanitize=address,undefined (oracle/Makefile: sanitize_check) stops at any out-of-bounds access or undefined
 * operation.  Prints one checksum; tests/test_oracle_sanitize.py compares it with the plain build's. */
#include "vq_oracle.c"

#include <stdio.h>

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void)
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 24);
}

static uint64_t mix(uint64_t h, const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 0x100000001B3ull;
    return h;
}

int main(void)
{
    uint64_t h = 0xcbf29ce484222325ull;
    const int shapes[][5] = {{7, 5, 3, 2, 2}, {33, 17, 1, 4, 4}, {9, 9, 1, 3, 3}, {5, 7, 3, 1, 1}, {3, 3, 1, 4, 4},
                             {64, 48, 3, 8, 8}, {130, 67, 1, 3, 5}, {40, 40, 2, 4, 2}};
    for (size_t s = 0; s < sizeof(shapes) / sizeof(shapes[0]); ++s) {
        const int H = shapes[s][0], W = shapes[s][1], C = shapes[s][2], bh = shapes[s][3], bw = shapes[s][4];
        const int nr = (H + bh - 1) / bh, nc = (W + bw - 1) / bw, d = bh * bw * C;
        const int64_t N = (int64_t)nr * nc;
        uint8_t *img = (uint8_t *)malloc((size_t)H * W * C);
        for (int i = 0; i < H * W * C; ++i) img[i] = (i % 11 == 0) ? 0 : (uint8_t)rnd();
        for (int pad = 0; pad < 2; ++pad) {
            uint8_t *blocks = (uint8_t *)malloc((size_t)N * d);
            vqo_blocks_u8(img, H, W, C, bh, bw, pad, blocks);
            h = mix(h, blocks, (size_t)N * d);
            uint64_t *col = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)d);
            vqo_colsum_u8(blocks, N, d, col);
            h = mix(h, col, sizeof(uint64_t) * (size_t)d);
            free(col);
            /* training: k not a power of two, k = 1, k > N */
            const int ks[] = {1, 5, 16};
            for (int t = 0; t < 3; ++t) {
                const int k = ks[t];
                int Kf = 1;
                while (Kf < k) Kf *= 2;
                double *cb = (double *)malloc(sizeof(double) * (size_t)Kf * d);
                int its[64], conv[64], nst = 0, nd = 0;
                double dist[256];
                const int K = vqo_lbg_train(blocks, N, d, k, 0.01, 0.001, 20, cb, its, conv, &nst, dist, 256, &nd);
                if (K < 0) {                                  /* k > N: the reference raises ValueError */
                    free(cb);
                    continue;
                }
                h = mix(h, cb, sizeof(double) * (size_t)K * d);
                h = mix(h, dist, sizeof(double) * (size_t)(nd < 256 ? nd : 256));
                if (K >= 2) memcpy(cb + (size_t)(K - 1) * d, cb, sizeof(double) * (size_t)d);   /* a duplicate codeword */
                int32_t *labels = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
                double *mind = (double *)malloc(sizeof(double) * (size_t)N);
                vqo_encode_l1(blocks, N, d, cb, K, labels, mind);
                h = mix(h, labels, sizeof(int32_t) * (size_t)N);
                h = mix(h, mind, sizeof(double) * (size_t)N);
                const double total = vqo_numpy_sum(mind, N);
                h = mix(h, &total, sizeof(total));
                cb[0] = 257.55;                               /* decode: wrap modulo 256 */
                if (d > 1) cb[1] = -1.5;
                if (d > 2) cb[2] = 1e300;
                uint8_t *out = (uint8_t *)malloc((size_t)nr * bh * nc * bw * C);
                vqo_decode_u8(labels, nr, nc, cb, K, bh, bw, C, out);
                h = mix(h, out, (size_t)nr * bh * nc * bw * C);
                const int bits = vqo_bits_needed(K);
                const size_t nbytes = (size_t)((N * bits + 7) / 8);
                uint8_t *packed = (uint8_t *)malloc(nbytes ? nbytes : 1);
                const size_t wrote = vqo_pack_labels(labels, N, K, packed);
                h = mix(h, packed, wrote);
                free(packed); free(out); free(mind); free(labels); free(cb);
            }
            free(blocks);
        }
        free(img);
    }
    /* numpy-order sum at lengths around the block boundaries of the pairwise routine */
    const int lens[] = {0, 1, 7, 8, 9, 127, 128, 129, 1000, 8191, 8192, 8193, 20000};
    for (size_t i = 0; i < sizeof(lens) / sizeof(lens[0]); ++i) {
        const int n = lens[i];
        double *a = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
        for (int j = 0; j < n; ++j) a[j] = (double)rnd() / 7.0;
        const double s1 = vqo_numpy_sum(a, n), s2 = vqo_pairwise_sum(a, n < 8192 ? n : 8192);
        h = mix(h, &s1, sizeof(s1));
        h = mix(h, &s2, sizeof(s2));
        free(a);
    }
    printf("%016llx\n", (unsigned long long)h);
    return 0;
}
