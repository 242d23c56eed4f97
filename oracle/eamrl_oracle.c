/*
 * eamrl_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's batched autoregressive construction rollout
 * (RL4CO fork at /root/reference; paths below are relative to it) with a DEFINED
 * floating-point evaluation order, so that the hand-written HIP kernels can be held to
 * bit-for-bit equality with it.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path never does.
 *
 * Parity status: PINNED -- this oracle is checked against golden vectors produced by
 * running the reference itself (tests/golden/make_golden.py, tests/test_oracle_golden.py):
 * identical actions / masks / integer state, float outputs within the tolerances written
 * in those tests (the reference's torch CPU kernels have no defined summation order, so
 * float equality with them is tolerance-level by nature, SURVEY.md section 7 hard part 1).
 *
 * What each function follows:
 *   orc_linear ................ torch.nn.Linear as used by every projection below
 *   orc_mha_encoder ........... rl4co/models/nn/attention.py:66-136 (MultiHeadAttention)
 *   orc_batchnorm_eval ........ rl4co/models/nn/ops.py:32-47  (BatchNorm1d, eval mode)
 *   orc_instancenorm .......... rl4co/models/nn/ops.py:48-49  (InstanceNorm1d, affine)
 *   orc_batchnorm_train ....... rl4co/models/nn/ops.py:45-47  (BatchNorm1d, training mode: batch statistics)
 *   orc_init_embed_* .......... rl4co/models/nn/env_embeddings/init.py:55-68,115-138
 *   orc_pointer_attention ..... rl4co/models/nn/attention.py:282-328 (PointerAttention.forward; the pointer= injection point)
 *   orc_exp1_noise ............ the counter-based Exp(1) draws of in-kernel sampling (replaces torch.multinomial's, utils/decoding.py:403-417)
 *   orc_mean_nodes ............ rl4co/models/zoo/am/decoder.py:225-227 (embeddings.mean(1))
 *   orc_decode_step ........... rl4co/models/zoo/am/decoder.py:133-198 (_compute_q/_compute_kvl/forward),
 *                               rl4co/models/nn/env_embeddings/context.py:50-74,105-157,
 *                               rl4co/models/nn/attention.py:282-328 (PointerAttention),
 *                               rl4co/utils/decoding.py:140-190 (process_logits), 391-417, 430-465
 *   orc_tsp_step .............. rl4co/envs/routing/tsp/env.py:62-88
 *   orc_cvrp_step ............. rl4co/envs/routing/cvrp/env.py:68-100,132-144
 *   orc_cvrp_mask ............. rl4co/envs/routing/cvrp/env.py:132-144
 *   orc_tour_length ........... rl4co/utils/ops.py:59-95 + tsp/env.py:152-159 + cvrp/env.py:146-155
 *   orc_check_tsp/cvrp ........ tsp/env.py:161-168, cvrp/env.py:157-185
 *   orc_rollout ............... rl4co/models/common/constructive/base.py:236-250 (decode loop)
 *   orc_sdvrp_mask/step ....... rl4co/envs/routing/sdvrp/env.py:58-92,137-146 (+ dynamic embedding, dynamic.py:59-78)
 *   orc_cvrptw_mask/step ...... rl4co/envs/routing/cvrptw/env.py:103-138; orc_check_cvrptw_time: :203-227
 *   orc_pctsp_mask/step ....... rl4co/envs/routing/pctsp/env.py:64-97,156-163; orc_pctsp_reward: :165-187;
 *                               orc_check_pctsp: :189-205
 *   orc_op_mask/step .......... rl4co/envs/routing/op/env.py:69-102,149-165; orc_op_reward: :167-177; orc_check_op: :179-212
 *   top-k / top-p, beam search  rl4co/utils/decoding.py:110-136,170-176,468-608 (beam search in oracle.py)
 *
 * DEFINED ORDER (DESIGN.md "Canonical arithmetic"):
 *   chain(x,y,K,init): acc=init; for k ascending: acc=fmaf(x[k],y[k],acc).  This is exactly
 *   what gfx950's f32 MFMA computes along its K dimension, and what a lane-sequential loop does.
 *   lane_tree(v,n): pad to a multiple of 64 with +0; inside each 64-block add adjacent pairs level
 *   by level (the result of a xor-butterfly over a 64-lane wavefront); add block sums ascending.
 *   Transcendentals are the polynomial d_expf/d_logf/d_tanhf below (only fmaf/mul/add/div, all
 *   IEEE-exact), never libm, so CPU and GPU agree bit-for-bit.
 *
 * Build: gcc -O2 -ffp-contract=off (see Makefile).  -ffp-contract=off is REQUIRED.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

enum { ORC_ENV_TSP = 0, ORC_ENV_CVRP = 1, ORC_ENV_SDVRP = 2, ORC_ENV_PCTSP = 3, ORC_ENV_OP = 4, ORC_ENV_CVRPTW = 5 };
enum { ORC_GREEDY = 0, ORC_SAMPLE = 1, ORC_EVALUATE = 2 };
#define ORC_NCHUNK 4 /* node chunks for the glimpse accumulation, column chunks for the logit dot */

/* ------------------------------------------------------------------------------------------
 * defined transcendentals
 * ---------------------------------------------------------------------------------------- */
static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* exp(x) for x <= 88; returns 0 below -87 (results stay normal numbers). */
static float d_expf(float x)
{
    if (!(x >= -87.0f)) return 0.0f;      /* also maps -inf (and NaN) to 0 */
    if (x > 88.0f) x = 88.0f;
    float t = x * 1.44269504088896341f;
    float n = rintf(t);                    /* round-half-even == v_rndne_f32 */
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    int ni = (int)n;
    return y * bits2f((uint32_t)(ni + 127) << 23);
}

/* log(x) for normal positive x. */
static float d_logf(float x)
{
    uint32_t u = f2bits(x);
    int e = (int)(u >> 23) - 126;                         /* x = m * 2^e, m in [0.5,1) */
    float m = bits2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = fmaf(p, m, -1.1514610310e-1f);
    p = fmaf(p, m, 1.1676998740e-1f);
    p = fmaf(p, m, -1.2420140846e-1f);
    p = fmaf(p, m, 1.4249322787e-1f);
    p = fmaf(p, m, -1.6668057665e-1f);
    p = fmaf(p, m, 2.0000714765e-1f);
    p = fmaf(p, m, -2.4999993993e-1f);
    p = fmaf(p, m, 3.3333331174e-1f);
    float y = (p * m) * z;
    float fe = (float)e;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    return fmaf(0.693359375f, fe, r);
}

/* 1 / x, x normal and positive: three Newton steps from an integer seed (integer subtraction + fma only; dmath.hpp d_rcpf). */
static float d_rcpf(float x)
{
    float r = bits2f(0x7EF311C7u - f2bits(x));
    for (int i = 0; i < 3; ++i) {
        float t = fmaf(-x, r, 1.0f);
        r = fmaf(r, t, r);
    }
    return r;
}

static float d_tanhf(float x)
{
    float a = fabsf(x);
    float t;
    if (a < 0.625f) {
        float z = a * a;
        float p = -5.70498872745e-3f;
        p = fmaf(p, z, 2.06390887954e-2f);
        p = fmaf(p, z, -5.37397155531e-2f);
        p = fmaf(p, z, 1.33314422036e-1f);
        p = fmaf(p, z, -3.33332819422e-1f);
        t = fmaf(p * z, a, a);
    } else if (a > 9.0f) {
        t = 1.0f;
    } else {
        float e = d_expf(a + a);
        t = fmaf(-2.0f, d_rcpf(e + 1.0f), 1.0f);
    }
    return copysignf(t, x);
}

/* worker threads used by every parallel loop below; returns the value in force */
ORC_API int orc_set_threads(int n)
{
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
}

ORC_API void orc_math_probe(const float* x, float* y_exp, float* y_log, float* y_tanh, long n)
{
    for (long i = 0; i < n; ++i) {
        y_exp[i] = d_expf(x[i]);
        y_log[i] = (x[i] > 0.0f) ? d_logf(x[i]) : 0.0f;
        y_tanh[i] = d_tanhf(x[i]);
    }
}

/* Counter-based Exp(1) noise (csrc/dmath.hpp exp1_noise4): Philox4x32-10, counter (node / 4, step, row), key = seed;
 * word x -> u = (2 (x >> 9) + 1) 2^-24 -> -d_logf(u).  noise [R][T][M]. */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

ORC_API void orc_exp1_noise(uint64_t seed, float* noise, long R, int T, int M)
{
#pragma omp parallel for schedule(static)
    for (long r = 0; r < R; ++r)
        for (int t = 0; t < T; ++t)
            for (int q = 0; 4 * q < M; ++q) {
                uint32_t c[4] = {(uint32_t)q, (uint32_t)t, (uint32_t)r, (uint32_t)((uint64_t)r >> 32)};
                philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
                for (int i = 0; i < 4 && 4 * q + i < M; ++i) {
                    const float u = (float)(2u * (c[i] >> 9) + 1u) * 5.9604644775390625e-8f;
                    noise[((long)r * T + t) * M + 4 * q + i] = 0.0f - d_logf(u);
                }
            }
}

/* adjacent-pair tree inside 64-blocks, blocks ascending */
static float lane_tree(const float* v, int n)
{
    float total = 0.0f;
    for (int b0 = 0; b0 < n; b0 += 64) {
        float t[64];
        for (int i = 0; i < 64; ++i) t[i] = (b0 + i < n) ? v[b0 + i] : 0.0f;
        for (int w = 1; w < 64; w <<= 1)
            for (int i = 0; i < 64; i += 2 * w) t[i] = t[i] + t[i + w];
        total = (b0 == 0) ? t[0] : total + t[0];
    }
    return total;
}

ORC_API float orc_lane_tree(const float* v, int n) { return lane_tree(v, n); }

/* ------------------------------------------------------------------------------------------
 * one-shot encoder pieces
 * ---------------------------------------------------------------------------------------- */
/* y[r][j] = chain_k x[r][k]*W[j][k], init bias[j] (or 0); optional ReLU.  W is [out][in] (torch). */
ORC_API void orc_linear(const float* x, const float* W, const float* bias, float* y,
                        long rows, int in_dim, int out_dim, int relu)
{
    /* 8 output columns at a time: 8 independent chains (each still strictly k-ascending) hide the fma latency */
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        const float* xr = x + r * in_dim;
        int j = 0;
        for (; j + 8 <= out_dim; j += 8) {
            const float* w = W + (long)j * in_dim;
            float a[8];
            for (int c = 0; c < 8; ++c) a[c] = bias ? bias[j + c] : 0.0f;
            for (int k = 0; k < in_dim; ++k) {
                const float xv = xr[k];
                for (int c = 0; c < 8; ++c) a[c] = fmaf(xv, w[(long)c * in_dim + k], a[c]);
            }
            for (int c = 0; c < 8; ++c) {
                float v = a[c];
                if (relu && !(v > 0.0f)) v = 0.0f;
                y[r * out_dim + j + c] = v;
            }
        }
        for (; j < out_dim; ++j) {
            const float* w = W + (long)j * in_dim;
            float acc = bias ? bias[j] : 0.0f;
            for (int k = 0; k < in_dim; ++k) acc = fmaf(xr[k], w[k], acc);
            if (relu && !(acc > 0.0f)) acc = 0.0f;
            y[r * out_dim + j] = acc;
        }
    }
}

/* y[r][j] = chain_k x[r][k]*Wt[k][j]  (right-multiply by a [in][out] matrix; used for Lp = L*Wout) */
ORC_API void orc_matmul_right(const float* x, const float* Wt, float* y, long rows, int in_dim, int out_dim)
{
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        const float* xr = x + r * in_dim;
        for (int j = 0; j < out_dim; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < in_dim; ++k) acc = fmaf(xr[k], Wt[(long)k * out_dim + j], acc);
            y[r * out_dim + j] = acc;
        }
    }
}

/* Encoder self-attention on packed qkv [B][N][3E] ("b s (three h d)"), no mask.
 * s = chain_d(q,k)*scale ; w = exp(s-max) ; Z = (P0 + P1) + (P2 + P3), Pg = sequential sum of w[j], j = g mod 4 ;
 * o = chain_j(w, v) / Z. */
ORC_API void orc_mha_encoder(const float* qkv, float* out, long B, int N, int E, int H)
{
    const int D = E / H;
    const float scale = 1.0f / sqrtf((float)D);
#pragma omp parallel for schedule(static)
    for (long b = 0; b < B; ++b) {
        float* s = (float*)malloc(sizeof(float) * N);
        const float* base = qkv + b * (long)N * 3 * E;
        for (int h = 0; h < H; ++h)
            for (int i = 0; i < N; ++i) {
                const float* q = base + (long)i * 3 * E + h * D;
                float m = -INFINITY;
                for (int j = 0; j < N; ++j) {
                    const float* k = base + (long)j * 3 * E + E + h * D;
                    float acc = 0.0f;
                    for (int d = 0; d < D; ++d) acc = fmaf(q[d], k[d], acc);
                    acc = acc * scale;
                    s[j] = acc;
                    if (acc > m) m = acc;
                }
                /* Z: four interleaved partial sums (keys j = g mod 4, ascending) combined as (P0 + P1) + (P2 + P3) -- the
                 * order a 16x16x4 MFMA score tile gives for free: a lane holds the keys of one residue class */
                float P[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                for (int j = 0; j < N; ++j) { s[j] = d_expf(s[j] - m); P[j & 3] = P[j & 3] + s[j]; }
                const float Z = (P[0] + P[1]) + (P[2] + P[3]);
                for (int d = 0; d < D; ++d) {
                    float acc = 0.0f;
                    for (int j = 0; j < N; ++j)
                        acc = fmaf(s[j], base[(long)j * 3 * E + 2 * E + h * D + d], acc);
                    out[(b * N + i) * (long)E + h * D + d] = acc / Z;
                }
            }
        free(s);
    }
}

/* PointerAttention.forward (rl4co/models/nn/attention.py:282-328) in the order of k_pointer_attention (csrc/pointer.hip):
 * per head, lane l of a 64-wide wavefront owns the nodes l, l+64, ...; its partial sums are combined by the lane tree. */
ORC_API void orc_pointer_attention(const float* q, const float* K, const float* V, const float* Lk, const uint8_t* mask,
                                   int mask_per_query, const float* Wout, const float* bout, float* logits, long B, int L,
                                   int M, int E, int H, int mask_inner)
{
    const int D = E / H;
    const float scale = 1.0f / sqrtf((float)D);
#pragma omp parallel for schedule(static)
    for (long row = 0; row < B * L; ++row) {
        const long b = row / L;
        const float* qr = q + row * E;
        const uint8_t* mr = mask ? mask + (mask_per_query ? row : b) * (long)M : NULL;
        float* s = (float*)malloc(sizeof(float) * M);
        float* heads = (float*)malloc(sizeof(float) * E);
        float* gl = (float*)malloc(sizeof(float) * E);
        for (int h = 0; h < H; ++h) {
            float m = -INFINITY;
            for (int n = 0; n < M; ++n) {
                const float* kr = K + (b * M + n) * (long)E + h * D;
                float acc = 0.0f;
                for (int d = 0; d < D; ++d) acc = fmaf(qr[h * D + d], kr[d], acc);
                acc = acc * scale;
                if (mask_inner && mr && !mr[n]) acc = -INFINITY;
                s[n] = acc;
                if (acc > m) m = acc;
            }
            float z[64], a[64];
            for (int l = 0; l < 64; ++l) {
                z[l] = 0.0f;
                for (int n = l; n < M; n += 64) { s[n] = (s[n] == -INFINITY) ? 0.0f : d_expf(s[n] - m); z[l] = z[l] + s[n]; }
            }
            const float Z = lane_tree(z, 64);
            for (int d = 0; d < D; ++d) {
                for (int l = 0; l < 64; ++l) {
                    a[l] = 0.0f;
                    for (int n = l; n < M; n += 64) a[l] = fmaf(s[n], V[(b * M + n) * (long)E + h * D + d], a[l]);
                }
                heads[h * D + d] = lane_tree(a, 64) / Z;
            }
        }
        for (int o = 0; o < E; ++o) {
            float acc = bout ? bout[o] : 0.0f;
            for (int i = 0; i < E; ++i) acc = fmaf(heads[i], Wout[(long)o * E + i], acc);
            gl[o] = acc;
        }
        const float inv = sqrtf((float)E);
        for (int n = 0; n < M; ++n) {
            const float* lr = Lk + (b * M + n) * (long)E;
            float acc = 0.0f;
            for (int o = 0; o < E; ++o) acc = fmaf(gl[o], lr[o], acc);
            logits[row * M + n] = acc / inv;
        }
        free(s); free(heads); free(gl);
    }
}

/* x += y (residual) */
ORC_API void orc_add_inplace(float* x, const float* y, long n)
{
    for (long i = 0; i < n; ++i) x[i] = x[i] + y[i];
}

/* BatchNorm1d eval: scale = gamma / sqrt(var + eps); shift = beta - mean*scale; y = fma(x, scale, shift) */
ORC_API void orc_batchnorm_eval(float* x, long rows, int E, const float* gamma, const float* beta,
                                const float* mean, const float* var, float eps)
{
    float* scale = (float*)malloc(sizeof(float) * E);
    float* shift = (float*)malloc(sizeof(float) * E);
    for (int e = 0; e < E; ++e) {
        scale[e] = gamma[e] / sqrtf(var[e] + eps);
        float ms = mean[e] * scale[e];
        shift[e] = beta[e] - ms;
    }
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r)
        for (int e = 0; e < E; ++e) x[r * E + e] = fmaf(x[r * E + e], scale[e], shift[e]);
    free(scale);
    free(shift);
}

/* BatchNorm1d with BATCH statistics (training mode, rl4co/models/nn/ops.py:45-47): per channel, chunks of 128 rows
 * summed sequentially, chunk sums added ascending; mean = sum / n; variance = (sum of fma(d, d, .), d = x - mean, same
 * chunking) / n; then the eval formula with those statistics.  running_* (may be NULL) updated as torch does:
 * running = (1 - momentum) * running + momentum * stat, the variance one unbiased (sum / (n - 1)). */
ORC_API void orc_batchnorm_train(float* x, long rows, int E, const float* gamma, const float* beta, float* running_mean,
                                 float* running_var, float momentum, float eps, float* save_mean, float* save_var)
{
    const long nch = (rows + 127) / 128;
    const float n = (float)rows;
    for (int e = 0; e < E; ++e) {
        float tot = 0.0f;
        for (long c = 0; c < nch; ++c) {
            const long r1 = (c + 1) * 128 < rows ? (c + 1) * 128 : rows;
            float s = 0.0f;
            for (long r = c * 128; r < r1; ++r) s = s + x[r * E + e];
            tot = tot + s;
        }
        const float mean = tot / n;
        float vt = 0.0f;
        for (long c = 0; c < nch; ++c) {
            const long r1 = (c + 1) * 128 < rows ? (c + 1) * 128 : rows;
            float s = 0.0f;
            for (long r = c * 128; r < r1; ++r) { const float d = x[r * E + e] - mean; s = fmaf(d, d, s); }
            vt = vt + s;
        }
        save_mean[e] = mean;
        save_var[e] = vt / n;
        if (running_mean) {
            const float keep = 1.0f - momentum;
            const float a = keep * running_mean[e], b = momentum * mean;
            running_mean[e] = a + b;
            const float unb = rows > 1 ? vt / (n - 1.0f) : vt / n;
            const float c2 = keep * running_var[e], d2 = momentum * unb;
            running_var[e] = c2 + d2;
        }
    }
    orc_batchnorm_eval(x, rows, E, gamma, beta, save_mean, save_var, eps);
}

/* InstanceNorm1d(affine) over nodes per (instance, channel): sequential sums, biased variance. */
ORC_API void orc_instancenorm(float* x, long B, int N, int E, const float* gamma, const float* beta, float eps)
{
#pragma omp parallel for schedule(static)
    for (long b = 0; b < B; ++b)
        for (int e = 0; e < E; ++e) {
            float* col = x + b * (long)N * E + e;
            float s = 0.0f;
            for (int n = 0; n < N; ++n) s = s + col[(long)n * E];
            float mean = s / (float)N;
            float v = 0.0f;
            for (int n = 0; n < N; ++n) { float d = col[(long)n * E] - mean; v = fmaf(d, d, v); }
            float inv = 1.0f / sqrtf(v / (float)N + eps);
            for (int n = 0; n < N; ++n) {
                float d = col[(long)n * E] - mean;
                col[(long)n * E] = fmaf(d * inv, gamma[e], beta[e]);
            }
        }
}

/* mean over nodes: sequential sum / M */
ORC_API void orc_mean_nodes(const float* emb, float* out, long B, int M, int E)
{
    for (long b = 0; b < B; ++b)
        for (int e = 0; e < E; ++e) {
            float s = 0.0f;
            for (int n = 0; n < M; ++n) s = s + emb[(b * M + n) * (long)E + e];
            out[b * E + e] = s / (float)M;
        }
}

/* ------------------------------------------------------------------------------------------
 * environments (integer / bool state machines)
 * ---------------------------------------------------------------------------------------- */
/* TSPEnv._step: mask is the action mask (1 = not visited). */
ORC_API void orc_tsp_step(uint8_t* mask, int64_t* first, int64_t* cur, int64_t* istep,
                          const int64_t* action, uint8_t* done, long R, int N)
{
    for (long r = 0; r < R; ++r) {
        int64_t a = action[r];
        if (istep[r] == 0) first[r] = a;
        cur[r] = a;
        mask[r * N + a] = 0;
        istep[r] += 1;
        int any = 0;
        for (int n = 0; n < N; ++n) any |= mask[r * N + n];
        done[r] = !any;
    }
}

/* CVRPEnv.get_action_mask; M = N + 1 (depot = node 0) */
ORC_API void orc_cvrp_mask(const uint8_t* visited, const float* used, const float* vcap, const float* demand,
                           const int64_t* cur, uint8_t* mask, long R, long Binst, int N)
{
    const int M = N + 1;
    for (long r = 0; r < R; ++r) {
        const float* dem = demand + (r % Binst) * N;
        float lim = vcap[r] + 1e-5f;
        int any_free = 0;
        for (int j = 0; j < N; ++j) {
            float load = dem[j] + used[r];
            int blocked = (visited[r * M + 1 + j] != 0) | (load > lim);
            mask[r * M + 1 + j] = !blocked;
            any_free |= !blocked;
        }
        int mask_depot = (cur[r] == 0) && any_free;
        mask[r * M] = !mask_depot;
    }
}

/* CVRPEnv._step (+ get_action_mask) */
ORC_API void orc_cvrp_step(uint8_t* visited, float* used, const float* vcap, const float* demand,
                           int64_t* cur, const int64_t* action, uint8_t* mask, uint8_t* done,
                           long R, long Binst, int N)
{
    const int M = N + 1;
    for (long r = 0; r < R; ++r) {
        int64_t a = action[r];
        int64_t di = a - 1; if (di < 0) di = 0; if (di > N - 1) di = N - 1;
        float d = demand[(r % Binst) * N + di];
        float nz = (a != 0) ? 1.0f : 0.0f;
        used[r] = (used[r] + d) * nz;
        visited[r * M + a] = 1;
        cur[r] = a;
        int cnt = 0;
        for (int n = 0; n < M; ++n) cnt += visited[r * M + n];
        done[r] = (cnt == M);
    }
    orc_cvrp_mask(visited, used, vcap, demand, cur, mask, R, Binst, N);
}

/* SDVRPEnv.get_action_mask  [rl4co/envs/routing/sdvrp/env.py:137-146].  rem [R][M] = demand_with_depot (remaining
 * demand, depot slot 0), used / vcap [R], cur [R] -> mask [R][M] (1 = feasible). */
ORC_API void orc_sdvrp_mask(const float* rem, const float* used, const float* vcap, const int64_t* cur,
                            uint8_t* mask, long R, int M)
{
    for (long r = 0; r < R; ++r) {
        const int full = used[r] >= vcap[r];
        int any_free = 0;
        for (int n = 1; n < M; ++n) {
            int blocked = (rem[r * M + n] == 0.0f) | full;
            mask[r * M + n] = !blocked;
            any_free |= !blocked;
        }
        mask[r * M] = !((cur[r] == 0) && any_free);
    }
}

/* SDVRPEnv._step (+ mask)  [sdvrp/env.py:58-92]: deliver min(remaining demand, free capacity) */
ORC_API void orc_sdvrp_step(float* rem, float* used, const float* vcap, int64_t* cur, const int64_t* action,
                            uint8_t* mask, uint8_t* done, long R, int M)
{
    for (long r = 0; r < R; ++r) {
        const int64_t a = action[r];
        const float sel = rem[r * M + a];
        const float free_cap = vcap[r] - used[r];
        const float delivered = sel < free_cap ? sel : free_cap;          /* torch.min */
        used[r] = (used[r] + delivered) * (a != 0 ? 1.0f : 0.0f);
        rem[r * M + a] = sel + (-delivered);                             /* scatter_add of -delivered */
        cur[r] = a;
        int any = 0;
        for (int n = 0; n < M; ++n) any |= (rem[r * M + n] > 0.0f);
        done[r] = !any;
    }
    orc_sdvrp_mask(rem, used, vcap, cur, mask, R, M);
}

/* Euclidean distance as torch's `.norm(p=2, dim=-1)` computes it for two components on the CPU: sqrtf(fmaf(dy, dy, dx*dx))
 * (checked bit for bit on 2e6 random pairs); the same expression as a tour leg of orc_tour_length. */
static float dist2(const float* a, const float* b)
{
    const float dx = a[0] - b[0], dy = a[1] - b[1];
    return sqrtf(fmaf(dy, dy, dx * dx));
}

/* OPEnv.get_action_mask  [rl4co/envs/routing/op/env.py:149-165].  locs [Binst][M][2], maxlen [Binst][M] (the per-node
 * arrival limit of the reset state: max_length - distance to the depot - 1e-6), tour_len [R]. */
ORC_API void orc_op_mask(const uint8_t* visited, const float* tour_len, const int64_t* cur, const float* locs,
                         const float* maxlen, uint8_t* mask, long R, long Binst, int M)
{
    for (long r = 0; r < R; ++r) {
        const float* L = locs + (r % Binst) * (long)M * 2;
        const float* ml = maxlen + (r % Binst) * (long)M;
        const uint8_t* v = visited + r * M;
        for (int n = 1; n < M; ++n) {
            const int exceeds = (tour_len[r] + dist2(L + 2 * n, L + 2 * cur[r])) > ml[n];
            mask[r * M + n] = !(v[n] | v[0] | exceeds);
        }
        mask[r * M] = 1;        /* the depot can always be visited */
    }
}

/* OPEnv._step (+ mask)  [op/env.py:69-102].  prize / prize_tot may be NULL (bookkeeping only) */
ORC_API void orc_op_step(uint8_t* visited, float* tour_len, float* prize_tot, const float* prize, const float* locs,
                         const float* maxlen, int64_t* cur, int64_t* istep, const int64_t* action, uint8_t* mask,
                         uint8_t* done, long R, long Binst, int M)
{
    for (long r = 0; r < R; ++r) {
        const int64_t a = action[r];
        const float* L = locs + (r % Binst) * (long)M * 2;
        tour_len[r] = tour_len[r] + dist2(L + 2 * a, L + 2 * cur[r]);
        if (prize_tot) prize_tot[r] = prize_tot[r] + prize[(r % Binst) * M + a];
        visited[r * M + a] = 1;
        done[r] = (a == 0) && (istep[r] > 0);
        cur[r] = a;
        istep[r] += 1;
    }
    orc_op_mask(visited, tour_len, cur, locs, maxlen, mask, R, Binst, M);
}

/* CVRPTWEnv.get_action_mask  [rl4co/envs/routing/cvrptw/env.py:103-116]: the CVRP mask AND "reachable before the time
 * window closes".  tw [Binst][M][2] (start, end) as floats, time [R] = current_time. */
ORC_API void orc_cvrptw_mask(const uint8_t* visited, const float* used, const float* vcap, const float* demand,
                             const int64_t* cur, const float* time, const float* locs, const float* tw, uint8_t* mask,
                             long R, long Binst, int N)
{
    const int M = N + 1;
    orc_cvrp_mask(visited, used, vcap, demand, cur, mask, R, Binst, N);
    for (long r = 0; r < R; ++r) {
        const float* L = locs + (r % Binst) * (long)M * 2;
        const float* W = tw + (r % Binst) * (long)M * 2;
        for (int n = 0; n < M; ++n) {
            const int in_time = (time[r] + dist2(L + 2 * cur[r], L + 2 * n)) <= W[2 * n + 1];
            mask[r * M + n] = mask[r * M + n] && in_time;
        }
    }
}

/* CVRPTWEnv._step  [cvrptw/env.py:118-138]: time = (a != 0) * (max(time + dist(cur, a), tw_start[a]) + duration[a]),
 * then CVRPEnv._step and the mask above.  dur [Binst][M]. */
ORC_API void orc_cvrptw_step(uint8_t* visited, float* used, const float* vcap, const float* demand, int64_t* cur,
                             float* time, const float* locs, const float* tw, const float* dur, const int64_t* action,
                             uint8_t* mask, uint8_t* done, long R, long Binst, int N)
{
    const int M = N + 1;
    for (long r = 0; r < R; ++r) {
        const int64_t a = action[r];
        const float* L = locs + (r % Binst) * (long)M * 2;
        const float* W = tw + (r % Binst) * (long)M * 2;
        const float arrive = time[r] + dist2(L + 2 * cur[r], L + 2 * a);
        const float start = arrive > W[2 * a] ? arrive : W[2 * a];                 /* torch.max */
        time[r] = (a != 0 ? 1.0f : 0.0f) * (start + dur[(r % Binst) * (long)M + a]);
    }
    orc_cvrp_step(visited, used, vcap, demand, cur, action, mask, done, R, Binst, N);
    orc_cvrptw_mask(visited, used, vcap, demand, cur, time, locs, tw, mask, R, Binst, N);
}

/* PCTSPEnv.get_action_mask  [rl4co/envs/routing/pctsp/env.py:156-163].  visited [R][M], prize_tot [R] (cur_total_prize):
 * a customer is feasible until visited and until the depot has been visited; the depot is infeasible while the collected
 * prize is below 1 and an unvisited customer remains. */
ORC_API void orc_pctsp_mask(const uint8_t* visited, const float* prize_tot, uint8_t* mask, long R, int M)
{
    for (long r = 0; r < R; ++r) {
        const uint8_t* v = visited + r * M;
        int unvisited = 0;
        for (int n = 1; n < M; ++n) {
            mask[r * M + n] = !(v[n] | v[0]);
            unvisited |= !v[n];
        }
        mask[r * M] = !((prize_tot[r] < 1.0f) && unvisited);
    }
}

/* PCTSPEnv._step (+ mask)  [pctsp/env.py:64-97].  prize / penalty [Binst][M] with a zero depot slot; pen_tot may be NULL */
ORC_API void orc_pctsp_step(uint8_t* visited, float* prize_tot, float* pen_tot, const float* prize, const float* penalty,
                            int64_t* cur, int64_t* istep, const int64_t* action, uint8_t* mask, uint8_t* done,
                            long R, long Binst, int M)
{
    for (long r = 0; r < R; ++r) {
        const int64_t a = action[r];
        prize_tot[r] = prize_tot[r] + prize[(r % Binst) * M + a];
        if (pen_tot) pen_tot[r] = pen_tot[r] + penalty[(r % Binst) * M + a];
        visited[r * M + a] = 1;
        done[r] = (istep[r] > 0) && (a == 0);
        cur[r] = a;
        istep[r] += 1;
    }
    orc_pctsp_mask(visited, prize_tot, mask, R, M);
}

/* ------------------------------------------------------------------------------------------
 * decode step (decoder + process_logits + selection); one row = one instance or one (start,instance)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int env;            /* ORC_ENV_* */
    long R;             /* rows */
    long Binst;         /* instances owning the cache; row r uses instance r % Binst */
    int M, E, H;
    const float *K, *V, *Lp;      /* [Binst][M][E] glimpse key, glimpse value, folded logit key */
    const float *Pa, *Pb;         /* TSP: P1 (first) / P2 (current); CVRP: Pc / NULL   [Binst][M][E] */
    const float *cvec;            /* TSP: c0[E] (placeholder query); CVRP: wcap[E] */
    const float *gctx;            /* [Binst][E] or NULL */
    float clip, temp;
    const float* dyn;   /* SDVRP dynamic embedding [3][E]: glimpse-key, glimpse-value and folded logit-key vectors that the
                         * reference scales by a node's remaining demand and adds to its K / V / Lp rows (folded here,
                         * see decode_row)                                              nn/env_embeddings/dynamic.py:59-78 */
    int top_k;          /* process_logits top-k filtering (0 = off)                      utils/decoding.py:110-115 */
    float top_p;        /* process_logits nucleus filtering (0 or >= 1 = off)            utils/decoding.py:118-136 */
} orc_dec_t;

static int decode_row(const orc_dec_t* c, long r, int64_t first, int64_t cur, int64_t istep, float remaining, float now,
                      const float* rem, const uint8_t* mask, int mode, const float* noise, int64_t given,
                      int64_t* out_action, float* out_logp, float* out_logits, float* out_logprobs, float* scratch)
{
    const int M = c->M, E = c->E, H = c->H, D = E / H;
    const long bi = r % c->Binst;
    const float* K = c->K + bi * (long)M * E;
    const float* V = c->V + bi * (long)M * E;
    const float* Lp = c->Lp + bi * (long)M * E;
    float* q = scratch;                 /* E */
    float* heads = q + E;               /* E */
    float* w = heads + E;               /* H*M */
    float* x = w + (long)H * M;         /* M */
    float* ex = x + M;                  /* M */

    /* D1 query */
    for (int e = 0; e < E; ++e) {
        float g = c->gctx ? c->gctx[bi * E + e] : 0.0f;
        float ctx;
        if (c->env == ORC_ENV_TSP) {
            if (istep == 0) ctx = c->cvec[e];
            else ctx = c->Pa[(bi * M + first) * (long)E + e] + c->Pb[(bi * M + cur) * (long)E + e];
        } else {   /* CVRP and SDVRP: VRPContext */
            ctx = fmaf(c->cvec[e], remaining, c->Pa[(bi * M + cur) * (long)E + e]);
            /* VRPTWContext: one more state column, the current time (context.py:160-176); cvec = [2][E] */
            if (c->env == ORC_ENV_CVRPTW) ctx = fmaf(c->cvec[E + e], now, ctx);
        }
        q[e] = ctx + g;
    }
    /* SDVRP: the reference adds rem[n] * (a dynamic vector) to row n of the cached glimpse key / value / logit key
     * before every step.  That is a rank-1 update, so it is folded like the other weight products:
     *   score[h][n]  = chain_d(q, K[n]) + rem[n] * chain_d(q, wk)             (one fma on top of the K chain)
     *   glimpse[e]   = (sum_n w[n] V[n][e]  +  R_h * wv[e]) / Z,   R_h = lane_tree_n(w[n] * rem[n])
     *   partial[n][g]= chain_e(heads, Lp[n]) + rem[n] * chain_e(heads, lw)     (per column chunk g)
     * identical to the reference in exact arithmetic; this order is the definition all implementations share. */
    const float* dk = (c->env == ORC_ENV_SDVRP) ? c->dyn : NULL;
    const float* dv = dk ? c->dyn + E : NULL;
    const float* dl = dk ? c->dyn + 2 * E : NULL;
    /* D2-D4 glimpse */
    const int C = (M + ORC_NCHUNK - 1) / ORC_NCHUNK;
    const float qk_scale = 1.0f / sqrtf((float)D);
    for (int h = 0; h < H; ++h) {
        float* wh = w + (long)h * M;
        float mx = -INFINITY;
        float qw = 0.0f;
        if (dk) for (int d = 0; d < D; ++d) qw = fmaf(q[h * D + d], dk[h * D + d], qw);
        for (int n = 0; n < M; ++n) {
            if (!mask[n]) { wh[n] = -INFINITY; continue; }
            float acc = 0.0f;
            for (int d = 0; d < D; ++d) acc = fmaf(q[h * D + d], K[(long)n * E + h * D + d], acc);
            if (dk) acc = fmaf(rem[n], qw, acc);
            acc = acc * qk_scale;                 /* 1/sqrt(D); D=16 -> exactly 0.25 */
            wh[n] = acc;
            if (acc > mx) mx = acc;
        }
        for (int n = 0; n < M; ++n) wh[n] = mask[n] ? d_expf(wh[n] - mx) : 0.0f;
        float Z = 0.0f, Rw = 0.0f;
        for (int g = 0; g < ORC_NCHUNK; ++g) {
            /* per chunk: four interleaved partial sums by position in the chunk, (n - g C) mod 4, combined as (P0 + P1) + (P2 + P3) */
            float P[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int n = g * C; n < M && n < (g + 1) * C; ++n) P[(n - g * C) & 3] = P[(n - g * C) & 3] + wh[n];
            const float zg = (P[0] + P[1]) + (P[2] + P[3]);
            Z = (g == 0) ? zg : Z + zg;
        }
        if (dv) {                           /* ex[] is free until the log-softmax below */
            for (int n = 0; n < M; ++n) ex[n] = wh[n] * rem[n];
            Rw = lane_tree(ex, M);
        }
        for (int d = 0; d < D; ++d) {
            float A = 0.0f;
            for (int g = 0; g < ORC_NCHUNK; ++g) {
                float ag = 0.0f;
                for (int n = g * C; n < M && n < (g + 1) * C; ++n)
                    ag = fmaf(wh[n], V[(long)n * E + h * D + d], ag);
                A = (g == 0) ? ag : A + ag;
            }
            if (dv) A = fmaf(Rw, dv[h * D + d], A);
            heads[h * D + d] = A / Z;
        }
    }
    /* D5-D6 logits, clip, mask, temperature */
    const int EC = E / ORC_NCHUNK;
    const float inv_sqrtE = 1.0f / sqrtf((float)E);      /* one rounded constant; logit = u * inv_sqrtE (round 3: was u / sqrt(E)) */
    float mx = -INFINITY;
    int nan_seen = 0;
    float hl[ORC_NCHUNK];
    for (int g = 0; g < ORC_NCHUNK; ++g) {
        hl[g] = 0.0f;
        if (dl) for (int e = g * EC; e < (g + 1) * EC; ++e) hl[g] = fmaf(heads[e], dl[e], hl[g]);
    }
    for (int n = 0; n < M; ++n) {
        float u = 0.0f;
        for (int g = 0; g < ORC_NCHUNK; ++g) {
            float cg = 0.0f;
            for (int e = g * EC; e < (g + 1) * EC; ++e) cg = fmaf(heads[e], Lp[(long)n * E + e], cg);
            if (dl) cg = fmaf(rem[n], hl[g], cg);
            u = (g == 0) ? cg : u + cg;
        }
        float logit = u * inv_sqrtE;
        if (logit != logit) nan_seen = 1;
        if (out_logits) out_logits[n] = logit;
        float v = (c->clip > 0.0f) ? d_tanhf(logit) * c->clip : logit;
        if (!mask[n]) v = -INFINITY;
        v = v / c->temp;
        x[n] = v;
        if (v > mx) mx = v;
    }
    /* D6b top-k / top-p filtering of the scaled logits (process_logits, decoding.py:170-176).
     *   top-k: keep n iff fewer than k entries are strictly larger (== "logits < k-th largest" removed; ties kept).
     *   top-p: p = softmax(x) (d_expf(x - max) / lane_tree sum); entries in ascending (value, index) order; running sum c
     *          sequential in that order; remove while c <= (float)(1 - top_p). */
    if (c->top_k > 0) {
        const int k = c->top_k < M ? c->top_k : M;
        for (int n = 0; n < M; ++n) {
            int cnt = 0;
            for (int m = 0; m < M; ++m) cnt += (x[m] > x[n]);
            ex[n] = (cnt < k) ? 1.0f : 0.0f;
        }
        for (int n = 0; n < M; ++n) if (ex[n] == 0.0f) x[n] = -INFINITY;
    }
    if (c->top_p > 0.0f && c->top_p < 1.0f) {
        const float thr = (float)(1.0 - (double)c->top_p);
        float m2 = -INFINITY;
        for (int n = 0; n < M; ++n) if (x[n] > m2) m2 = x[n];
        for (int n = 0; n < M; ++n) ex[n] = (x[n] > -INFINITY) ? d_expf(x[n] - m2) : 0.0f;
        const float Z = lane_tree(ex, M);
        float* srt = w;                               /* the glimpse weights are no longer needed: H*M >= 2M floats */
        int* rk = (int*)(w + M);
        for (int n = 0; n < M; ++n) {
            int r0 = 0;
            for (int m = 0; m < M; ++m) r0 += (x[m] < x[n]) || (x[m] == x[n] && m < n);
            rk[n] = r0;
            srt[r0] = ex[n] / Z;
        }
        float cs = 0.0f;
        for (int j = 0; j < M; ++j) { cs = cs + srt[j]; srt[j] = (cs <= thr) ? 1.0f : 0.0f; }
        for (int n = 0; n < M; ++n) if (srt[rk[n]] != 0.0f) x[n] = -INFINITY;
    }
    if (c->top_k > 0 || (c->top_p > 0.0f && c->top_p < 1.0f)) {
        mx = -INFINITY;
        for (int n = 0; n < M; ++n) if (x[n] > mx) mx = x[n];
    }
    /* D7 log-softmax (an entry takes part iff it is still finite: masked and filtered entries are -inf) */
    for (int n = 0; n < M; ++n) ex[n] = (x[n] > -INFINITY) ? d_expf(x[n] - mx) : 0.0f;
    float lse = d_logf(lane_tree(ex, M));
    for (int n = 0; n < M; ++n) {
        x[n] = (x[n] > -INFINITY) ? (x[n] - mx) - lse : -INFINITY;
        if (out_logprobs) out_logprobs[n] = x[n];
    }
    /* D8 selection (lowest index wins ties, as torch.argmax) */
    int64_t a = 0;
    if (mode == ORC_EVALUATE) {
        a = given;
    } else if (mode == ORC_GREEDY) {
        float best = x[0];
        for (int n = 1; n < M; ++n) if (x[n] > best) { best = x[n]; a = n; }
    } else {
        float best = d_expf(x[0]) / noise[0];
        for (int n = 1; n < M; ++n) {
            float rr = d_expf(x[n]) / noise[n];
            if (rr > best) { best = rr; a = n; }
        }
    }
    *out_action = a;
    *out_logp = x[a];
    return nan_seen ? -1 : (mask[a] ? 0 : -2);
}

/* One decode step for R rows.  State arrays are per row.  mask: [R][M] u8 (1 = feasible).
 * remaining capacity (CVRP) = vcap - used is computed here, as VRPContext._state_embedding does.
 * noise: [R][M] or NULL; given: [R] or NULL; out_logits/out_logprobs: [R][M] or NULL.
 * Returns 0, -1 (NaN logits), -2 (infeasible action selected). */
ORC_API int orc_decode_step(int env, long R, long Binst, int M, int E, int H,
                            const float* K, const float* V, const float* Lp,
                            const float* Pa, const float* Pb, const float* cvec, const float* gctx,
                            const int64_t* first, const int64_t* cur, const int64_t* istep,
                            const float* used, const float* vcap, const uint8_t* mask,
                            const float* rem, const float* dyn, const float* time,
                            int mode, const float* noise, const int64_t* given, float clip, float temp,
                            int top_k, float top_p,
                            int64_t* out_action, float* out_logp, float* out_logits, float* out_logprobs)
{
    orc_dec_t c = { env, R, Binst, M, E, H, K, V, Lp, Pa, Pb, cvec, gctx, clip, temp, dyn, top_k, top_p };
    int status = 0;
#pragma omp parallel
    {
        float* scratch = (float*)malloc(sizeof(float) * (2 * (long)E + (long)H * M + 2 * (long)M));
#pragma omp for schedule(static)
        for (long r = 0; r < R; ++r) {
            float remaining = (env != ORC_ENV_TSP) ? (vcap[r] - used[r]) : 0.0f;
            /* PCTSPContext: clamp(prize_required - cur_total_prize, min=0)   nn/env_embeddings/context.py:194-208 */
            if (env == ORC_ENV_PCTSP && remaining < 0.0f) remaining = 0.0f;
            int st = decode_row(&c, r, first ? first[r] : 0, cur[r], istep ? istep[r] : 1, remaining, time ? time[r] : 0.0f,
                                rem ? rem + r * (long)M : NULL, mask + r * (long)M, mode, noise ? noise + r * (long)M : NULL,
                                given ? given[r] : 0, out_action + r, out_logp + r,
                                out_logits ? out_logits + r * (long)M : NULL,
                                out_logprobs ? out_logprobs + r * (long)M : NULL, scratch);
            if (st != 0) {
#pragma omp critical
                if (status == 0) status = st;
            }
        }
        free(scratch);
    }
    return status;
}

/* ------------------------------------------------------------------------------------------
 * reward and validity
 * ---------------------------------------------------------------------------------------- */
/* Closed tour length over `actions` [R][T] into locs [Binst][M][2]; with_depot prepends node 0.
 * out = -length (the reward). */
ORC_API void orc_tour_length(const float* locs, const int64_t* actions, float* reward,
                             long R, long Binst, int M, int T, int with_depot)
{
#pragma omp parallel
    {
        const int P = T + (with_depot ? 1 : 0);
        float* d = (float*)malloc(sizeof(float) * P);
#pragma omp for schedule(static)
        for (long r = 0; r < R; ++r) {
            const float* L = locs + (r % Binst) * (long)M * 2;
            for (int t = 0; t < P; ++t) {
                int64_t a0, a1;
                if (with_depot) {
                    a0 = (t == 0) ? 0 : actions[r * T + t - 1];
                    a1 = (t + 1 == P) ? 0 : actions[r * T + t];
                } else {
                    a0 = actions[r * T + t];
                    a1 = actions[r * T + ((t + 1 == P) ? 0 : t + 1)];
                }
                float dx = L[2 * a1] - L[2 * a0];
                float dy = L[2 * a1 + 1] - L[2 * a0 + 1];
                d[t] = sqrtf(fmaf(dy, dy, dx * dx));
            }
            reward[r] = -lane_tree(d, P);
        }
        free(d);
    }
}

/* PCTSPEnv._get_reward  [pctsp/env.py:165-187]: saved penalties - (tour length from / to the depot + all penalties).
 * penalty [Binst][M] with a zero depot slot.  Sums of penalties go through the lane tree like the tour length. */
ORC_API void orc_pctsp_reward(const float* locs, const float* penalty, const int64_t* actions, float* reward,
                              long R, long Binst, int M, int T)
{
    float* len = (float*)malloc(sizeof(float) * R);
    orc_tour_length(locs, actions, len, R, Binst, M, T, 1);          /* = -length */
    float* sv = (float*)malloc(sizeof(float) * (T > M ? T : M));
    for (long r = 0; r < R; ++r) {
        const float* pen = penalty + (r % Binst) * (long)M;
        for (int t = 0; t < T; ++t) sv[t] = pen[actions[r * T + t]];
        const float saved = lane_tree(sv, T);
        const float total = lane_tree(pen + 1, M - 1);
        reward[r] = saved - ((0.0f - len[r]) + total);
    }
    free(sv); free(len);
}

/* OPEnv._get_reward  [op/env.py:167-177]: sum of the collected prizes (lane tree over the steps). prize [Binst][M]. */
ORC_API void orc_op_reward(const float* prize, const int64_t* actions, float* reward, long R, long Binst, int M, int T)
{
    float* sv = (float*)malloc(sizeof(float) * T);
    for (long r = 0; r < R; ++r) {
        for (int t = 0; t < T; ++t) sv[t] = prize[(r % Binst) * (long)M + actions[r * T + t]];
        reward[r] = lane_tree(sv, T);
    }
    free(sv);
}

/* OPEnv.check_solution_validity  [op/env.py:179-212]: customers at most once; closed length over the actions
 * <= ((maxlen[n] + dist(depot, n)) + 1e-6) + 1e-5 for every node n.  Returns duplicate rows + 1000000 * over-length rows. */
ORC_API long orc_check_op(const int64_t* actions, const float* locs, const float* maxlen, long R, long Binst, int M, int T)
{
    long dup = 0, over = 0;
    uint8_t* seen = (uint8_t*)malloc(M);
    float* len = (float*)malloc(sizeof(float) * R);
    orc_tour_length(locs, actions, len, R, Binst, M, T, 0);          /* = -length, closed over the actions */
    for (long r = 0; r < R; ++r) {
        memset(seen, 0, M);
        int ok = 1;
        for (int t = 0; t < T; ++t) {
            int64_t a = actions[r * T + t];
            if (a < 0 || a >= M) { ok = 0; break; }
            if (a != 0) { if (seen[a]) { ok = 0; break; } seen[a] = 1; }
        }
        if (!ok) { ++dup; continue; }
        const float* L = locs + (r % Binst) * (long)M * 2;
        const float length = 0.0f - len[r];
        int ex = 0;
        for (int n = 0; n < M; ++n) {
            const float lim = ((maxlen[(r % Binst) * (long)M + n] + dist2(L, L + 2 * n)) + 1e-6f) + 1e-5f;
            ex |= !(length <= lim);
        }
        over += ex;
    }
    free(seen); free(len);
    return dup + 1000000 * over;
}

/* The time-window replay of CVRPTWEnv.check_solution_validity  [cvrptw/env.py:203-227]: arrival times are truncated to
 * integers there ((curr_time + dist).int()), service starts at max(arrival, window start) and must not be after the
 * window end; the depot resets the clock.  Returns the number of rows that miss a deadline. */
ORC_API long orc_check_cvrptw_time(const int64_t* actions, const float* locs, const float* tw, const float* dur,
                                   long R, long Binst, int M, int T)
{
    long late = 0;
    for (long r = 0; r < R; ++r) {
        const float* L = locs + (r % Binst) * (long)M * 2;
        const float* W = tw + (r % Binst) * (long)M * 2;
        float curr = 0.0f;
        int64_t node = 0;
        int bad = 0;
        for (int t = 0; t < T; ++t) {
            int64_t nx = actions[r * T + t];
            if (nx < 0 || nx >= M) { bad = 1; break; }
            int ct = (int)(curr + dist2(L + 2 * node, L + 2 * nx));
            const int ws = (int)W[2 * nx];
            if (ws > ct) ct = ws;
            if ((float)ct > W[2 * nx + 1]) bad = 1;
            curr = (float)ct + dur[(r % Binst) * (long)M + nx];
            node = nx;
            if (nx == 0) curr = 0.0f;
        }
        late += bad;
    }
    return late;
}

/* PCTSPEnv.check_solution_validity  [pctsp/env.py:189-205].  Returns rows with a customer visited twice (or an id out of
 * range) + 1000000 * rows that neither collect a total prize >= 1 - 1e-5 nor visit every customer. */
ORC_API long orc_check_pctsp(const int64_t* actions, const float* prize, long R, long Binst, int M, int T)
{
    long dup = 0, low = 0;
    uint8_t* seen = (uint8_t*)malloc(M);
    for (long r = 0; r < R; ++r) {
        memset(seen, 0, M);
        int ok = 1, cnt = 0;
        float p = 0.0f;
        for (int t = 0; t < T; ++t) {
            int64_t a = actions[r * T + t];
            if (a < 0 || a >= M) { ok = 0; break; }
            if (a != 0) { if (seen[a]) { ok = 0; break; } seen[a] = 1; ++cnt; }
            p = p + prize[(r % Binst) * M + a];
        }
        if (!ok) { ++dup; continue; }
        if (!((p >= (float)(1.0 - 1e-5)) || cnt == M - 1)) ++low;
    }
    free(seen);
    return dup + 1000000 * low;
}

/* sum of per-step selected log-probs, sequential over t */
ORC_API void orc_sum_logp(const float* logp, float* out, long R, int T)
{
    for (long r = 0; r < R; ++r) {
        float s = 0.0f;
        for (int t = 0; t < T; ++t) s = s + logp[r * T + t];
        out[r] = s;
    }
}

/* TSP: every node exactly once.  Returns the number of invalid rows. */
ORC_API long orc_check_tsp(const int64_t* actions, long R, int N)
{
    long bad = 0;
    uint8_t* seen = (uint8_t*)malloc(N);
    for (long r = 0; r < R; ++r) {
        memset(seen, 0, N);
        int ok = 1;
        for (int t = 0; t < N; ++t) {
            int64_t a = actions[r * N + t];
            if (a < 0 || a >= N || seen[a]) { ok = 0; break; }
            seen[a] = 1;
        }
        bad += !ok;
    }
    free(seen);
    return bad;
}

/* CVRP: customers exactly once, any number of depot visits, running load <= cap + 1e-5.
 * Returns invalid-tour rows + 1000000 * over-capacity rows. */
ORC_API long orc_check_cvrp(const int64_t* actions, const float* demand, const float* vcap,
                            long R, long Binst, int N, int T)
{
    long bad_tour = 0, bad_cap = 0;
    uint8_t* seen = (uint8_t*)malloc(N + 1);
    for (long r = 0; r < R; ++r) {
        memset(seen, 0, N + 1);
        int ok = 1;
        for (int t = 0; t < T; ++t) {
            int64_t a = actions[r * T + t];
            if (a < 0 || a > N) { ok = 0; break; }
            if (a != 0) { if (seen[a]) { ok = 0; break; } seen[a] = 1; }
        }
        for (int n = 1; n <= N && ok; ++n) if (!seen[n]) ok = 0;
        bad_tour += !ok;
        if (!ok) continue;
        float usedc = 0.0f, lim = vcap[r] + 1e-5f;
        int over = 0;
        for (int t = 0; t < T; ++t) {
            int64_t a = actions[r * T + t];
            float dd = (a == 0) ? -vcap[r] : demand[(r % Binst) * N + a - 1];
            usedc = usedc + dd;
            if (usedc < 0.0f) usedc = 0.0f;
            if (usedc > lim) over = 1;
        }
        bad_cap += over;
    }
    free(seen);
    return bad_tour + 1000000 * bad_cap;
}

/* ------------------------------------------------------------------------------------------
 * whole rollout: decode step + env step until every row is done (ConstructivePolicy.forward loop)
 * ---------------------------------------------------------------------------------------- */
/* State in/out as for the step functions.  actions/logps: [R][Tmax] (row-major, right-padded as the
 * reference does: finished CVRP rows keep choosing the depot).  noise: [R][Tmax][M] or NULL;
 * given: [R][Tgiven] or NULL (teacher forcing).  Returns the number of steps taken, or <0 on error. */
ORC_API int orc_rollout(int env, long R, long Binst, int M, int E, int H,
                        const float* K, const float* V, const float* Lp,
                        const float* Pa, const float* Pb, const float* cvec, const float* gctx,
                        int64_t* first, int64_t* cur, int64_t* istep,
                        float* used, const float* vcap, const float* demand,
                        uint8_t* mask, uint8_t* visited, uint8_t* done, float* rem, const float* dyn,
                        const float* locs,   /* OP, CVRPTW: [Binst][M][2]; OP: `demand` is then the arrival limit [Binst][M] */
                        const float* tw, const float* dur, float* time,   /* CVRPTW: [Binst][M][2], [Binst][M], [R] */
                        int mode, const float* noise, const int64_t* given, int Tgiven,
                        float clip, float temp, int top_k, float top_p, int Tmax,
                        int64_t* actions, float* logps)
{
    int64_t* a = (int64_t*)malloc(sizeof(int64_t) * R);
    float* lp = (float*)malloc(sizeof(float) * R);
    float* nz = noise ? (float*)malloc(sizeof(float) * R * M) : NULL;
    int64_t* gv = given ? (int64_t*)malloc(sizeof(int64_t) * R) : NULL;
    int t = 0, status = 0;
    for (;;) {
        int all_done = 1;
        for (long r = 0; r < R; ++r) if (!done[r]) { all_done = 0; break; }
        if (all_done || t >= Tmax) break;
        if (noise) for (long r = 0; r < R; ++r) memcpy(nz + r * M, noise + (r * (long)Tmax + t) * M, sizeof(float) * M);
        if (given) for (long r = 0; r < R; ++r) gv[r] = (t < Tgiven) ? given[r * (long)Tgiven + t] : 0;
        status = orc_decode_step(env, R, Binst, M, E, H, K, V, Lp, Pa, Pb, cvec, gctx, first, cur, istep,
                                 used, vcap, mask, rem, dyn, time, mode, nz, gv, clip, temp, top_k, top_p, a, lp, NULL, NULL);
        if (status != 0) break;
        for (long r = 0; r < R; ++r) { actions[r * (long)Tmax + t] = a[r]; logps[r * (long)Tmax + t] = lp[r]; }
        if (env == ORC_ENV_TSP) orc_tsp_step(mask, first, cur, istep, a, done, R, M);
        else if (env == ORC_ENV_CVRP) orc_cvrp_step(visited, used, vcap, demand, cur, a, mask, done, R, Binst, M - 1);
        else if (env == ORC_ENV_PCTSP) orc_pctsp_step(visited, used, NULL, demand, NULL, cur, istep, a, mask, done, R, Binst, M);
        else if (env == ORC_ENV_CVRPTW)
            orc_cvrptw_step(visited, used, vcap, demand, cur, time, locs, tw, dur, a, mask, done, R, Binst, M - 1);
        else if (env == ORC_ENV_OP) orc_op_step(visited, used, NULL, NULL, locs, demand, cur, istep, a, mask, done, R, Binst, M);
        else orc_sdvrp_step(rem, used, vcap, cur, a, mask, done, R, M);
        ++t;
    }
    free(a); free(lp); free(nz); free(gv);
    return status != 0 ? status : t;
}
