/*
 * mlp_oracle.c -- CPU ORACLE (test infrastructure) for the per-point 1x1-conv MLP stacks of the
 * Dis-PU generator.  Same rules as dispu_oracle.c: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it.
 *
 * A 1x1 conv (Common/tf_util.py:52-115 conv1d, :120-185 conv2d) is a per-point dense layer
 *      y = act( x . W + b )
 * TensorFlow/cuDNN's summation order is not pinned by anything in the reference tree ("parity
 * unpinned" at that boundary).  The restatement pins it to what a v_mfma_f32_32x32x2_f32
 * k-loop computes: acc = 0; for k ascending: acc = fmaf(x[k], W[k][o], acc); then one rounded
 * "+ b" (tf.nn.bias_add is a separate op, tf_util.py:106,176), then the activation.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

int orc_nt(long items);     /* dispu_oracle.c: min(configured threads, items) */

/* y[r, 0:n] (row stride ldy) = act( x[r, 0:k] (row stride ldx) . w[k, n] + bias ), act: 0 none, 1 relu.
 * bias may be NULL (no add at all, the raw chain). */
ORC_API void orc_linear(long m, int k, int n, const float *x, long ldx, const float *w, const float *bias, int act,
                        float *y, long ldy) {
#pragma omp parallel num_threads(orc_nt(m / 16))
    {
        float *acc = (float *)malloc(sizeof(float) * (size_t)n);
#pragma omp for schedule(static)
        for (long r = 0; r < m; ++r) {
            const float *xr = x + r * ldx;
            for (int o = 0; o < n; ++o) acc[o] = 0.0f;
            for (int kk = 0; kk < k; ++kk) {
                const float xv = xr[kk];
                const float *wr = w + (size_t)kk * n;
                for (int o = 0; o < n; ++o) acc[o] = fmaf(xv, wr[o], acc[o]);
            }
            float *yr = y + r * ldy;
            for (int o = 0; o < n; ++o) {
                float v = acc[o];
                if (bias) v = v + bias[o];
                if (act == 1) v = v > 0.0f ? v : 0.0f;
                yr[o] = v;
            }
        }
        free(acc);
    }
}

/* c[b, i, j] = chain_k a[b, i, k] * bt[b, j, k]   (A . B^T per batch; the attention logits
 * Q.K^T of PointNonLocalCell, Common/ops.py:326) */
ORC_API void orc_matmul_nt(int b, int m, int n, int k, const float *a, const float *bt, float *c) {
#pragma omp parallel for collapse(2) schedule(static) num_threads(orc_nt((long)b * m / 16))
    for (int bb = 0; bb < b; ++bb)
        for (int i = 0; i < m; ++i) {
            const float *ar = a + ((size_t)bb * m + i) * k;
            for (int j = 0; j < n; ++j) {
                const float *br = bt + ((size_t)bb * n + j) * k;
                float acc = 0.0f;
                for (int kk = 0; kk < k; ++kk) acc = fmaf(ar[kk], br[kk], acc);
                c[((size_t)bb * m + i) * n + j] = acc;
            }
        }
}

/* c[b, i, j] = chain_k a[b, i, k] * bm[b, k, j]   (A . B per batch: attention . V, ops.py:339, and the
 * per-point feature x weight product of PointShuffle2, ops.py:1066-1067) */
static long bb_items(long b, int m) { return m >= 16 ? b : b / 16; }
ORC_API void orc_matmul_nn(long b, int m, int n, int k, const float *a, const float *bm, float *c) {
#pragma omp parallel for schedule(static) num_threads(orc_nt(bb_items(b, m)))
    for (long bb = 0; bb < b; ++bb)
        for (int i = 0; i < m; ++i) {
            const float *ar = a + ((size_t)bb * m + i) * k;
            float *cr = c + ((size_t)bb * m + i) * n;
            for (int j = 0; j < n; ++j) cr[j] = 0.0f;
            for (int kk = 0; kk < k; ++kk) {
                const float av = ar[kk];
                const float *br = bm + ((size_t)bb * k + kk) * n;
                for (int j = 0; j < n; ++j) cr[j] = fmaf(av, br[j], cr[j]);
            }
        }
}
