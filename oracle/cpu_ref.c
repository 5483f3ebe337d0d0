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

/* =====================================================================================================================
 * Training step (forward + backward to the 8 decoder variables), the CPU baseline of bench.py.
 *
 * variant 0 "compact":  no [C, m^3, k^3*20] window tensor; every query row gathers its 5^3 window straight from fv.
 * variant 1 "faithful": the dataflow of the TF graph -- local_z_3d materialises emb [C, m^3, k^3*20]
 *                       (utils/dpdist_util.py:911-930, 164 MB per cloud set at B = 32), the cell lookup compares every query
 *                       against all m^3 centres and takes the argmax (:459-492), rows are gathered from emb (:434-457).
 * Both run the same dense layers: an OpenMP cache-blocked, packed SGEMM written here (no BLAS in the image).
 * Backward = TF autodiff of loss_samples = mean |pred_AB[...,0] - labels| (utils/dpdist_util.py:967-974) w.r.t. the variables
 * under 'pc_compare' (train_multi_gpu_pc_compare_dist.py:274-277): only the AB half of the rows carries gradient.
 * ===================================================================================================================== */
/* Dense layers: a cache-blocked, packed SGEMM in the BLIS loop order (no BLAS in the image).  The widest vector the build
 * target has: AVX-512 (16 floats, 12 x 32 register tile: 24 accumulators + 2 operands + 1 broadcast of 32 zmm) or AVX2 (8 floats,
 * 6 x 16 tile: 15 of 16 ymm).  K is cut into KC-deep slices; a slice of B is packed once into NR-wide panels (shared, L2/L3), every
 * task packs its MC x KC block of A into MR-tall panels (private, L2) and walks the register tiles with both operands streaming
 * from contiguous memory.  Round 2's kernel streamed the full K per register tile from the unpacked operands (0.5 TFLOP/s on a
 * 128-core host, under 2 % of its fp32 peak).  Tasks = (row block, column block) pairs, OpenMP dynamic. */
#ifdef __AVX512F__
#define VL 16
#define MR 12
#else
#define VL 8
#define MR 6
#endif
typedef float vNf __attribute__((vector_size(4 * VL)));
#define NR (2 * VL)
#define KC 256
#define MC (MR * 8)
#define NC 256
#define GEMM_FAST __attribute__((optimize("-ffp-contract=fast")))   /* FMA in the dense layers only (the encoder stays op by op) */

/* acc[MR][2] (+)= Ap[k][0:MR] (x) Bp[k][0:NR] over k < kc; stores the first mr rows */
GEMM_FAST static inline void micro_tile(int kc, const float* ap, const float* bp, float* c, long ldc, int mr, int accumulate) {
    vNf acc[MR][2];
    for (int r = 0; r < MR; ++r) { acc[r][0] = (vNf){0}; acc[r][1] = (vNf){0}; }
    for (int k = 0; k < kc; ++k) {
        vNf b0, b1;
        memcpy(&b0, bp + (long)k * NR, sizeof(vNf));
        memcpy(&b1, bp + (long)k * NR + VL, sizeof(vNf));
#pragma GCC unroll 12
        for (int r = 0; r < MR; ++r) {
            const float av = ap[(long)k * MR + r];     /* scalar * vector: GCC splats the scalar (one vbroadcastss) */
            acc[r][0] += av * b0;
            acc[r][1] += av * b1;
        }
    }
    for (int r = 0; r < mr; ++r) {
        float* cr = c + (long)r * ldc;
        if (accumulate) {
            vNf c0, c1;
            memcpy(&c0, cr, sizeof(vNf)); memcpy(&c1, cr + VL, sizeof(vNf));
            acc[r][0] += c0; acc[r][1] += c1;
        }
        memcpy(cr, &acc[r][0], sizeof(vNf));
        memcpy(cr + VL, &acc[r][1], sizeof(vNf));
    }
}

/* C [M,N] = op(A) B, B [K,N] row major, N % NR == 0.  a_rs / a_ks: row / k strides of A (NN: lda, 1;  TN with A stored [K,M]: 1, lda). */
static void sgemm_blocked(int M, int N, int K, const float* A, long a_rs, long a_ks, const float* B, int ldb, float* C, int ldc) {
    const int nblk_m = (M + MC - 1) / MC, nblk_n = (N + NC - 1) / NC;
    float* Bp = (float*)aligned_alloc(64, sizeof(float) * (size_t)KC * (size_t)((N + NR - 1) / NR * NR));
    for (int pc = 0; pc < K; pc += KC) {
        const int kc = (K - pc < KC) ? K - pc : KC;
#pragma omp parallel
        {
#pragma omp for schedule(static)
            for (int jr = 0; jr < N; jr += NR) {          /* pack B[pc:pc+kc, jr:jr+NR] -> panel [kc][NR] */
                float* dst = Bp + (size_t)(jr / NR) * KC * NR;
                for (int k = 0; k < kc; ++k) memcpy(dst + (size_t)k * NR, B + (size_t)(pc + k) * ldb + jr, sizeof(float) * NR);
            }
            float* Ap = (float*)aligned_alloc(64, sizeof(float) * (size_t)MC * KC);
#pragma omp for collapse(2) schedule(dynamic, 1)
            for (int ib = 0; ib < nblk_m; ++ib)
                for (int jb = 0; jb < nblk_n; ++jb) {
                    const int ic = ib * MC, mc = (M - ic < MC) ? M - ic : MC;
                    const int jc = jb * NC, nc = (N - jc < NC) ? N - jc : NC;
                    for (int ir = 0; ir < mc; ir += MR) {  /* pack A[ic+ir : +MR, pc : pc+kc] -> panel [kc][MR], zero rows beyond M */
                        float* dst = Ap + (size_t)(ir / MR) * KC * MR;
                        const int mr = (mc - ir < MR) ? mc - ir : MR;
                        for (int k = 0; k < kc; ++k) {
                            const float* src = A + (long)(ic + ir) * a_rs + (long)(pc + k) * a_ks;
                            for (int r = 0; r < MR; ++r) dst[(size_t)k * MR + r] = (r < mr) ? src[(long)r * a_rs] : 0.f;
                        }
                    }
                    for (int jr = 0; jr < nc; jr += NR)
                        for (int ir = 0; ir < mc; ir += MR)
                            micro_tile(kc, Ap + (size_t)(ir / MR) * KC * MR, Bp + (size_t)((jc + jr) / NR) * KC * NR,
                                       C + (size_t)(ic + ir) * ldc + jc + jr, ldc, (mc - ir < MR) ? mc - ir : MR, pc > 0);
                }
            free(Ap);
        }
    }
    free(Bp);
}

static void sgemm_nn(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc) {
    sgemm_blocked(M, N, K, A, lda, 1, B, ldb, C, ldc);
}

/* C [M,N] = A^T B with A stored [K,M], B [K,N] (the weight gradients: act^T g). */
static void sgemm_tn(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc) {
    sgemm_blocked(M, N, K, A, 1, lda, B, ldb, C, ldc);
}

static void transpose(const float* A, int R, int Cc, float* AT) {
#pragma omp parallel for schedule(static)
    for (int c = 0; c < Cc; ++c)
        for (int r = 0; r < R; ++r) AT[(size_t)c * R + r] = A[(size_t)r * Cc + c];
}

static void bias_relu(float* y, int R, int N, const float* b) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < R; ++r)
        for (int n = 0; n < N; ++n) y[(size_t)r * N + n] = fmaxf(y[(size_t)r * N + n] + b[n], 0.f);
}

static void colsum(const float* g, int R, int N, float* out) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += g[(size_t)r * N + n];
        out[n] = s;
    }
}

/*
 * One training step's forward + backward.  Weights as in cpuref_forward (TF layout, W1 rows 0-2 = local xyz); H % 16 == 0.
 * Outputs: loss[2] = (loss_samples, loss_pred); gradients dW1 [D,H], db1 [H], dW2, db2, dW3, db3 [H,H],[H], dW4 [H,3], db4 [3]
 * (any may be NULL -> that gradient is still computed but not returned); predAB/predBA [B,N,3] may be NULL.
 * do_backward = 0: forward only.
 */
void cpuref_train_step(const float* pcA, const float* pcB, const float* noise, const float* labels, int B, int N, int m, int k,
                       float sigma, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                       const float* b3, const float* W4, const float* b4, int H, int variant, int do_backward, float* loss,
                       float* dW1, float* db1, float* dW2, float* db2, float* dW3, float* db3, float* dW4, float* db4,
                       float* predAB, float* predBA) {
    const int G = m * m * m, E = k * k * k * F, D = E + 3, h = (k - 1) / 2, C = 2 * B, Q = C * N, Qb = B * N;
    float ax[64];
    grid_axis(m, ax);
    const float half = fabsf(ax[0] - ax[1]) / 2.0f;
    float* pts = (float*)malloc(sizeof(float) * (size_t)C * N * 3);
    float* qs = (float*)malloc(sizeof(float) * (size_t)C * N * 3);
    for (size_t i = 0; i < (size_t)B * N * 3; ++i) {
        pts[i] = noise ? pcA[i] + noise[i] : pcA[i];
        pts[(size_t)B * N * 3 + i] = pcB[i];
        qs[i] = pcB[i];
        qs[(size_t)B * N * 3 + i] = pcA[i];
    }
    float* fv = (float*)malloc(sizeof(float) * (size_t)C * G * F);
    cpuref_mfv3d(pts, C, N, m, sigma, fv);
    float* X = (float*)calloc((size_t)Q * D, sizeof(float));
    float* mask = (float*)malloc(sizeof(float) * Q);
    float* emb = NULL;
    if (variant == 1) {   /* local_z_3d: the window of EVERY voxel of every cloud (:911-930) */
        emb = (float*)calloc((size_t)C * G * E, sizeof(float));
#pragma omp parallel for collapse(2) schedule(static)
        for (int c = 0; c < C; ++c)
            for (int v = 0; v < G; ++v) {
                const int iy = v / (m * m), ix = (v / m) % m, iz = v % m;
                float* e = emb + ((size_t)c * G + v) * E;
                for (int d0 = 0; d0 < k; ++d0)
                    for (int d1 = 0; d1 < k; ++d1)
                        for (int d2 = 0; d2 < k; ++d2) {
                            const int g0 = iy + d0 - h, g1 = ix + d1 - h, g2 = iz + d2 - h;
                            if (g0 >= 0 && g0 < m && g1 >= 0 && g1 < m && g2 >= 0 && g2 < m)
                                memcpy(e + ((d0 * k + d1) * k + d2) * F, fv + ((size_t)c * G + (g0 * m + g1) * m + g2) * F, sizeof(float) * F);
                        }
            }
    }
#pragma omp parallel for schedule(static)
    for (int r = 0; r < Q; ++r) {
        const int c = r / N;
        const float* q = qs + (size_t)r * 3;
        int ix, iy, iz, valid;
        if (variant == 1) {   /* mask against ALL centres + argmax (:470-490); centre v = (l[ix], l[iy], l[iz]), v = (iy*m+ix)*m+iz */
            int best = 0, found = 0;
            for (int v = 0; v < G && !found; ++v) {
                const int vy = v / (m * m), vx = (v / m) % m, vz = v % m;
                if (q[0] > ax[vx] - half && q[0] <= ax[vx] + half && q[1] > ax[vy] - half && q[1] <= ax[vy] + half &&
                    q[2] > ax[vz] - half && q[2] <= ax[vz] + half) { best = v; found = 1; }
            }
            valid = found; iy = best / (m * m); ix = (best / m) % m; iz = best % m;
        } else {
            ix = cell_of(ax, m, half, q[0]); iy = cell_of(ax, m, half, q[1]); iz = cell_of(ax, m, half, q[2]);
            valid = ix >= 0 && iy >= 0 && iz >= 0;
            if (!valid) ix = iy = iz = 0;
        }
        mask[r] = valid ? 1.f : 0.f;
        float* x = X + (size_t)r * D;
        x[0] = q[0] - ax[ix]; x[1] = q[1] - ax[iy]; x[2] = q[2] - ax[iz];
        if (variant == 1) {
            memcpy(x + 3, emb + ((size_t)c * G + (iy * m + ix) * m + iz) * E, sizeof(float) * E);   /* gather_nd (:436-453) */
        } else {
            for (int d0 = 0; d0 < k; ++d0)
                for (int d1 = 0; d1 < k; ++d1)
                    for (int d2 = 0; d2 < k; ++d2) {
                        const int g0 = iy + d0 - h, g1 = ix + d1 - h, g2 = iz + d2 - h;
                        if (g0 >= 0 && g0 < m && g1 >= 0 && g1 < m && g2 >= 0 && g2 < m)
                            memcpy(x + 3 + ((d0 * k + d1) * k + d2) * F, fv + ((size_t)c * G + (g0 * m + g1) * m + g2) * F, sizeof(float) * F);
                    }
        }
    }
    float* h1 = (float*)malloc(sizeof(float) * (size_t)Q * H);
    float* h2 = (float*)malloc(sizeof(float) * (size_t)Q * H);
    float* h3 = (float*)malloc(sizeof(float) * (size_t)Q * H);
    float* y = (float*)malloc(sizeof(float) * (size_t)Q * 3);
    sgemm_nn(Q, H, D, X, D, W1, H, h1, H); bias_relu(h1, Q, H, b1);      /* :516-544, tf_util.conv2d == dense */
    sgemm_nn(Q, H, H, h1, H, W2, H, h2, H); bias_relu(h2, Q, H, b2);
    sgemm_nn(Q, H, H, h2, H, W3, H, h3, H); bias_relu(h3, Q, H, b3);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < Q; ++r)
        for (int ch = 0; ch < 3; ++ch) {
            float s = 0.f;
            for (int j = 0; j < H; ++j) s += h3[(size_t)r * H + j] * W4[(size_t)j * 3 + ch];
            y[(size_t)r * 3 + ch] = s + b4[ch];
        }
    double sl = 0.0, sab = 0.0, sba = 0.0;
    for (int r = 0; r < Q; ++r) {
        for (int ch = 0; ch < 3; ++ch) {
            const float v = fminf(fmaxf(y[(size_t)r * 3 + ch], 0.f), 6.f) / 3.0f * mask[r];
            if (r < Qb) { if (predAB) predAB[(size_t)r * 3 + ch] = v; }
            else if (predBA) predBA[(size_t)(r - Qb) * 3 + ch] = v;
            if (ch == 0) {
                if (r < Qb) { sl += fabsf(v - labels[r]); sab += v; } else sba += v;
            }
        }
    }
    if (loss) { loss[0] = (float)(sl / Qb); loss[1] = (float)((sab / Qb + sba / Qb) / 2.0); }
    if (do_backward) {
        float* dy = (float*)calloc((size_t)Qb * 3, sizeof(float));
        float* g3 = (float*)malloc(sizeof(float) * (size_t)Qb * H);
        float* g2 = (float*)malloc(sizeof(float) * (size_t)Qb * H);
        float* g1 = (float*)malloc(sizeof(float) * (size_t)Qb * H);
        float* WT = (float*)malloc(sizeof(float) * (size_t)H * H);
        float* tmpW = (float*)malloc(sizeof(float) * (size_t)D * H);
        float tb[3] = {0, 0, 0};
        for (int r = 0; r < Qb; ++r) {   /* d mean|p - l| / d y: sign * 1/Qb * mask / 3 on (0, 6) (relu6 gradient) */
            const float yv = y[(size_t)r * 3];
            const float p = fminf(fmaxf(yv, 0.f), 6.f) / 3.0f * mask[r];
            const float df = p - labels[r];
            const float sg = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
            dy[(size_t)r * 3] = (yv > 0.f && yv < 6.f) ? sg / (float)Qb * mask[r] / 3.0f : 0.f;
            tb[0] += dy[(size_t)r * 3];
        }
        if (db4) { db4[0] = tb[0]; db4[1] = 0.f; db4[2] = 0.f; }
        if (dW4) {
#pragma omp parallel for schedule(static)
            for (int j = 0; j < H; ++j) {
                float s = 0.f;
                for (int r = 0; r < Qb; ++r) s += h3[(size_t)r * H + j] * dy[(size_t)r * 3];
                dW4[(size_t)j * 3] = s; dW4[(size_t)j * 3 + 1] = 0.f; dW4[(size_t)j * 3 + 2] = 0.f;
            }
        }
#pragma omp parallel for schedule(static)
        for (int r = 0; r < Qb; ++r)
            for (int j = 0; j < H; ++j)
                g3[(size_t)r * H + j] = (h3[(size_t)r * H + j] > 0.f) ? dy[(size_t)r * 3] * W4[(size_t)j * 3] : 0.f;
        float* dWo[3] = {dW3, dW2, dW1};
        float* dbo[3] = {db3, db2, db1};
        const float* Wl[2] = {W3, W2};
        float* gin[3] = {g3, g2, g1};
        const float* act_in[3] = {h2, h1, X};
        const float* act_gate[2] = {h2, h1};
        for (int l = 0; l < 3; ++l) {
            const int Kin = (l == 2) ? D : H;
            if (dbo[l]) colsum(gin[l], Qb, H, dbo[l]);
            float* dst = dWo[l] ? dWo[l] : tmpW;
            sgemm_tn(Kin, H, Qb, act_in[l], Kin, gin[l], H, dst, H);        /* dW = act^T g */
            if (l < 2) {                                                     /* g_prev = (g W^T) * [act > 0] */
                transpose(Wl[l], H, H, WT);
                sgemm_nn(Qb, H, H, gin[l], H, WT, H, gin[l + 1], H);
                const float* gate = act_gate[l];
                float* gp = gin[l + 1];
#pragma omp parallel for schedule(static)
                for (size_t i = 0; i < (size_t)Qb * H; ++i) gp[i] = (gate[i] > 0.f) ? gp[i] : 0.f;
            }
        }
        free(dy); free(g3); free(g2); free(g1); free(WT); free(tmpW);
    }
    free(pts); free(qs); free(fv); free(X); free(mask); free(h1); free(h2); free(h3); free(y);
    if (emb) free(emb);
}
