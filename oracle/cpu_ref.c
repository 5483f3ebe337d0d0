/*
 * cpu_ref.c -- plain-C restatement of the DPDist forward path (second, independent oracle).
 *
 * TEST INFRASTRUCTURE: only tests/ and bench.py's cpu_baseline leg may load the library built from this file
 * (oracle/_build/libdpd_cpuref.so, recipe in oracle/Makefile).  The product never links it.
 * Pinned by tests/test_oracle_c.py against tests/golden/*.npz (outputs of the reference's own Python run under
 * oracle/tfstub) and against oracle/restate.py.
 *
 * Each function cites the reference lines (relative to /root/reference) it follows.  float32 arithmetic, op by op,
 * like the TF graph (compile with -ffp-contract=off).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define F 20

/* utils/dpdist_util.py:42 / :987-988 -- axis centres, double then cast to float32 */
static void grid_axis(int m, float* ax) {
    const double step = 2.0 / (double)m;
    for (int i = 0; i < m; ++i) {
        volatile double v = (double)i * step;
        v = v + (-1.0);
        v = v + 1.0 / (double)m;
        ax[i] = (float)v;
    }
}

static float pnorm(float x) { /* :119-121 */
    if (x != x) return x;
    float s = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
    return s * sqrtf(fmaxf(fabsf(x), 1e-12f));
}

/* utils/dpdist_util.py:22-141: pts [C,N,3] -> fv [C,m^3,20] */
void cpuref_mfv3d(const float* pts, int C, int N, int m, float sigma, float* fv) {
    const int G = m * m * m;
    float ax[64];
    grid_axis(m, ax);
    const float w = 1.0f / (float)G;
    const float lognorm = (float)(0.5 * 3.0 * log(2.0 * M_PI) + 3.0 * log((double)sigma));
    const float dpi_den = sqrtf(w) * (float)N, mu_scale = 1.0f / sqrtf(w), sg_scale = 1.0f / sqrtf(2.0f * w);
#pragma omp parallel for schedule(dynamic)
    for (int c = 0; c < C; ++c) {
        const float* p = pts + (size_t)c * N * 3;
        float* den = (float*)malloc(sizeof(float) * N);
        float* out = fv + (size_t)c * G * F;
        for (int n = 0; n < N; ++n) { /* :73-74 denominators */
            float acc = 0.f;
            for (int g = 0; g < G; ++g) {
                const int i = g / (m * m), j = (g / m) % m, t = g % m; /* centre (x,y,z) = (l[j], l[i], l[t]), :47-48 */
                const float zx = (p[n * 3] - ax[j]) / sigma, zy = (p[n * 3 + 1] - ax[i]) / sigma, zz = (p[n * 3 + 2] - ax[t]) / sigma;
                acc += expf(-0.5f * (zx * zx + zy * zy + zz * zz) - lognorm) * w;
            }
            den[n] = acc;
        }
        for (int g = 0; g < G; ++g) {
            const int i = g / (m * m), j = (g / m) % m, t = g % m;
            const float cc[3] = {ax[j], ax[i], ax[t]};
            float pis = 0.f, pimx = -INFINITY, mus[3] = {0, 0, 0}, mumx[3], mumn[3], sgs[3] = {0, 0, 0}, sgmx[3], sgmn[3];
            for (int d = 0; d < 3; ++d) { mumx[d] = sgmx[d] = -INFINITY; mumn[d] = sgmn[d] = INFINITY; }
            for (int n = 0; n < N; ++n) {
                float z[3];
                for (int d = 0; d < 3; ++d) z[d] = (p[n * 3 + d] - cc[d]) / sigma;
                const float Q = (expf(-0.5f * (z[0] * z[0] + z[1] * z[1] + z[2] * z[2]) - lognorm) * w) / den[n];
                const float dpi = (Q - w) / dpi_den; /* :78 */
                pis += dpi;
                if (dpi > pimx || dpi != dpi) pimx = dpi;
                for (int d = 0; d < 3; ++d) {
                    const float a = Q * z[d], b = Q * (z[d] * z[d] - 1.0f); /* :87, :100 */
                    mus[d] += a; sgs[d] += b;
                    if (a > mumx[d] || a != a) mumx[d] = a;
                    if (a < mumn[d] || a != a) mumn[d] = a;
                    if (b > sgmx[d] || b != b) sgmx[d] = b;
                    if (b < sgmn[d] || b != b) sgmn[d] = b;
                }
            }
            float* o = out + (size_t)g * F;
            o[0] = pnorm(pis / (float)N);
            o[1] = pnorm(pimx);
            for (int d = 0; d < 3; ++d) {
                o[2 + d] = pnorm((mus[d] / (float)N) * mu_scale);
                o[5 + d] = pnorm(mumx[d] * mu_scale);
                o[8 + d] = pnorm(mumn[d] * mu_scale);
                o[11 + d] = pnorm((sgs[d] / (float)N) * sg_scale);
                o[14 + d] = pnorm(sgmx[d] * sg_scale);
                o[17 + d] = pnorm(sgmn[d] * sg_scale);
            }
        }
        for (int f = 0; f < F; ++f) { /* :124-126 l2_normalize over the Gaussian axis */
            float ss = 0.f;
            for (int g = 0; g < G; ++g) ss += out[(size_t)g * F + f] * out[(size_t)g * F + f];
            const float sc = (ss != ss) ? ss : 1.0f / sqrtf(fmaxf(ss, 1e-12f));
            for (int g = 0; g < G; ++g) out[(size_t)g * F + f] *= sc;
        }
        free(den);
    }
}

/* first cell with q > c - g && q <= c + g (utils/dpdist_util.py:478-487), -1 if none */
static int cell_of(const float* ax, int m, float half, float q) {
    for (int i = 0; i < m; ++i)
        if (q > ax[i] - half && q <= ax[i] + half) return i;
    return -1;
}

static void dense_relu(const float* x, int R, int K, const float* W, const float* b, int Nout, int relu, float* y) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < R; ++r) {
        float* yr = y + (size_t)r * Nout;
        for (int n = 0; n < Nout; ++n) yr[n] = 0.f;
        for (int k = 0; k < K; ++k) {
            const float xv = x[(size_t)r * K + k];
            if (xv == 0.f) continue;
            const float* wr = W + (size_t)k * Nout;
            for (int n = 0; n < Nout; ++n) yr[n] += xv * wr[n];
        }
        for (int n = 0; n < Nout; ++n) {
            const float v = yr[n] + b[n];
            yr[n] = relu ? fmaxf(v, 0.f) : v;
        }
    }
}

/*
 * models/dpdist_and_aue.py:31-86 forward.  Weights in the TF layout flattened: W1 [k^3*20+3, H] (rows 0-2 multiply
 * the local xyz), W2,W3 [H,H], W4 [H,3].  Outputs predAB, predBA [B,N,3].  noise may be NULL.
 */
void cpuref_forward(const float* pcA, const float* pcB, const float* noise, int B, int N, int m, int k, float sigma,
                    const float* W1, const float* b1, const float* W2, const float* b2, const float* W3, const float* b3,
                    const float* W4, const float* b4, int H, float* predAB, float* predBA) {
    const int G = m * m * m, E = k * k * k * F, D = E + 3, h = (k - 1) / 2, C = 2 * B, Q = C * N;
    float ax[64];
    grid_axis(m, ax);
    const float half = fabsf(ax[0] - ax[1]) / 2.0f; /* :468 */
    float* pts = (float*)malloc(sizeof(float) * (size_t)C * N * 3);
    float* qs = (float*)malloc(sizeof(float) * (size_t)C * N * 3);
    for (size_t i = 0; i < (size_t)B * N * 3; ++i) {
        pts[i] = noise ? pcA[i] + noise[i] : pcA[i];          /* :45 */
        pts[(size_t)B * N * 3 + i] = pcB[i];
        qs[i] = pcB[i];                                       /* AB half queries pcB against fv(A) (:494-496) */
        qs[(size_t)B * N * 3 + i] = pcA[i];                   /* BA half queries the un-noised pcA (:69, :498-500) */
    }
    float* fv = (float*)malloc(sizeof(float) * (size_t)C * G * F);
    cpuref_mfv3d(pts, C, N, m, sigma, fv);
    float* X = (float*)calloc((size_t)Q * D, sizeof(float));
    float* mask = (float*)malloc(sizeof(float) * Q);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < Q; ++r) {
        const int c = r / N;
        const float* q = qs + (size_t)r * 3;
        int ix = cell_of(ax, m, half, q[0]), iy = cell_of(ax, m, half, q[1]), iz = cell_of(ax, m, half, q[2]);
        const int valid = ix >= 0 && iy >= 0 && iz >= 0;
        if (!valid) ix = iy = iz = 0;                          /* argmax of an all-zero row (:490) */
        mask[r] = valid ? 1.f : 0.f;
        float* x = X + (size_t)r * D;
        x[0] = q[0] - ax[ix]; x[1] = q[1] - ax[iy]; x[2] = q[2] - ax[iz];   /* :491, local coords FIRST (:455) */
        for (int d0 = 0; d0 < k; ++d0)
            for (int d1 = 0; d1 < k; ++d1)
                for (int d2 = 0; d2 < k; ++d2) {               /* extract_volume_patches, SAME, (plane,row,col,ch) (:922) */
                    const int g0 = iy + d0 - h, g1 = ix + d1 - h, g2 = iz + d2 - h;
                    float* dst = x + 3 + ((d0 * k + d1) * k + d2) * F;
                    if (g0 >= 0 && g0 < m && g1 >= 0 && g1 < m && g2 >= 0 && g2 < m)
                        memcpy(dst, fv + ((size_t)c * G + (g0 * m + g1) * m + g2) * F, sizeof(float) * F);
                }
    }
    float* h1 = (float*)malloc(sizeof(float) * (size_t)Q * H);
    float* h2 = (float*)malloc(sizeof(float) * (size_t)Q * H);
    float* y = (float*)malloc(sizeof(float) * (size_t)Q * 3);
    dense_relu(X, Q, D, W1, b1, H, 1, h1);                     /* :516-544 */
    dense_relu(h1, Q, H, W2, b2, H, 1, h2);
    dense_relu(h2, Q, H, W3, b3, H, 1, h1);
    dense_relu(h1, Q, H, W4, b4, 3, 0, y);
    for (int r = 0; r < Q; ++r)
        for (int ch = 0; ch < 3; ++ch) {
            const float v = fminf(fmaxf(y[(size_t)r * 3 + ch], 0.f), 6.f) / 3.0f * mask[r];   /* :691, :697-698 */
            if (r < B * N) predAB[(size_t)r * 3 + ch] = v;
            else predBA[(size_t)(r - B * N) * 3 + ch] = v;
        }
    free(pts); free(qs); free(fv); free(X); free(mask); free(h1); free(h2); free(y);
}
