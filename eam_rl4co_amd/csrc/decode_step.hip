// Step-API decode kernels: one 256-thread workgroup per rollout row, cache streamed from HBM/L2.
//
//   k_decode_step<ENV>     one decode step (+ optionally the fused env step) for every row
//   k_rollout_stream<ENV>  whole decode loop per row, re-reading the row's K/V/Lp every step (any M)
//
// Reference path replaced: AttentionModelDecoder.forward + PointerAttention.forward +
// process_logits + Greedy/Sampling/Evaluate + TSPEnv._step / CVRPEnv._step
// (rl4co/models/zoo/am/decoder.py:161-198, rl4co/models/nn/attention.py:282-328,
//  rl4co/utils/decoding.py:140-190,346-465, rl4co/envs/routing/{tsp,cvrp}/env.py).
//
// HBM traffic per row-step (algorithmic): K, V, Lp rows of the still-feasible nodes
// (12*E B per node), 1-2 gathered context rows, mask row r/w.  Coalescing: V is read with
// lane = column (512 B contiguous per node); K and Lp are read as float4 runs of 64 B / 128 B per
// lane, consecutive lanes consecutive runs, so every fetched line is fully used.
#include "kernels.hpp"

namespace eamrl {

constexpr int BLOCK = 256;
constexpr int DU2 = 8;    // score stage: (node, head) pairs whose key loads are in flight together
#ifndef EAMRL_DU4
#define EAMRL_DU4 16
#endif
#ifndef EAMRL_DU5
#define EAMRL_DU5 2
#endif
constexpr int DU4 = EAMRL_DU4;   // glimpse stage: value loads in flight per thread
constexpr int DU5 = EAMRL_DU5;   // logit stage: (node, column chunk) pairs per trip, 8 float4 loads each
constexpr int NWAVE = BLOCK / EAMRL_WAVE;

#ifdef EAMRL_STAMPS   // development build only (tools/build_stamps.sh): per-stage cycle sums seen by thread 0 of each row
__device__ unsigned long long g_sstamps[16];
#define SSTAMP(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); \
                       atomicAdd(&g_sstamps[i], now_ - sst_t); sst_t = now_; } } while (0)
#define SSTAMP_DECL unsigned long long sst_t = __builtin_readcyclecounter()
#else
#define SSTAMP(i) do { } while (0)
#define SSTAMP_DECL do { } while (0)
#endif

struct RowState {
    int64_t first, cur, istep;
    float used, vcap;
    float now;          // CVRPTW: current_time
};

// block-wide max over values already reduced per thread; result broadcast to all threads
__device__ __forceinline__ float block_max(float v, float* red)
{
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    float m = red[0];
#pragma unroll
    for (int i = 1; i < NWAVE; ++i) m = __builtin_fmaxf(m, red[i]);
    return m;
}

// Decode one row with the whole workgroup.  l.msk holds the row's action mask.  Outputs are
// workgroup-uniform.  logprobs_row / logits_row may be null.
//
// SDVRP (ENV == EAMRL_ENV_SDVRP): `rem` (LDS, [M]) is the row's remaining demand and `dyn` (LDS, [3][E] + scratch) the
// dynamic-embedding vectors wk | wv | lw.  The reference adds rem[n] * vector to row n of the cached K / V / logit key
// every step (zoo/am/decoder.py:176-183); that rank-1 update is folded:
//   score[h][n] = fma(rem[n], q_h.wk_h, q_h.K[n]_h)      heads[e] = fma(lane_tree_n(w[n] * rem[n]), wv[e], sum_n w[n] V[n][e]) / Z
//   logit partial[n][c] = fma(rem[n], heads_c.lw_c, heads_c.Lp[n]_c)
// so K / V / Lp are read exactly as for CVRP.  Both pointers are unused otherwise.
template <int ENV>
__device__ void decode_row(const DecArgs& a, const RowLds& l, int64_t r, const RowState& s, const float* noise_row,
                           int64_t given, int64_t& out_a, float& out_lp, float* logprobs_row, float* logits_row,
                           const float* rem = nullptr, const float* dyn = nullptr)
{
    constexpr bool SD = ENV == EAMRL_ENV_SDVRP;
    float* sd_qw = const_cast<float*>(dyn) + 3 * a.E;          // [H]          q_h . wk_h
    float* sd_pr = sd_qw + a.H;                                // [H]          lane-tree sum of w * rem
    float* sd_hl = sd_pr + a.H;                                // [NCHUNK]     heads_c . lw_c
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int M = a.M, E = a.E, H = a.H, D = E / H;
    // LDS strides of q (per head) and heads (per column chunk): + 4 floats, so that the 8 heads / 4 chunks a wavefront reads
    // side by side start in different banks (with the dense layout they are 64 / 128 B apart: 4-way conflicts on every read)
    const int QS = D + RowLds::PAD, HS = E / EAMRL_NCHUNK + RowLds::PAD;
    const int64_t bi = r % a.B;
    const int64_t ld = a.ld;
    const float* K = a.K + bi * M * ld;
    const float* V = a.V + bi * M * ld;
    const float* Lp = a.Lp + bi * M * ld;

    SSTAMP_DECL;
    // ---- D1 context query ----------------------------------------------------------------------
    for (int e = tid; e < E; e += BLOCK) {
        float g = a.gctx ? a.gctx[bi * E + e] : 0.0f;
        float ctx;
        if (ENV == EAMRL_ENV_TSP) {
            if (s.istep == 0) ctx = a.cvec[e];
            else ctx = a.Pa[(bi * M + s.first) * ld + e] + a.Pb[(bi * M + s.cur) * ld + e];
        } else {
            // VRPContext: free capacity; PCTSPContext: clamp(prize_required - cur_total_prize, min=0)  (context.py:160-208)
            float state = s.vcap - s.used;
            if (ENV == EAMRL_ENV_PCTSP) state = state < 0.0f ? 0.0f : state;
            ctx = fma_(a.cvec[e], state, a.Pa[(bi * M + s.cur) * ld + e]);
            if (ENV == EAMRL_ENV_CVRPTW) ctx = fma_(a.cvec[E + e], s.now, ctx);      // VRPTWContext: + current time column
        }
        l.q[(e / D) * QS + (e % D)] = ctx + g;
    }
    __syncthreads();

    // ---- D2 scores s[h][n] = chain_d(q, K) / sqrt(D) ------------------------------------------------
    const float qk_scale = 1.0f / __builtin_sqrtf((float)D);
    if (SD) {       // q_h . wk_h, one thread per head
        if (tid < H) {
            float qw = 0.0f;
            for (int d = 0; d < D; ++d) qw = fma_(l.q[tid * QS + d], dyn[tid * D + d], qw);
            sd_qw[tid] = qw;
        }
        __syncthreads();
    }
    if (D == 16) {
    // DU2 (node, head) pairs per thread and trip: all 16 key loads of a trip are in flight before the first is used
    // (this kernel lives on HBM / L2 latency; a load-use loop would expose it once per pair)
    for (int p0 = tid; p0 < M * H; p0 += DU2 * BLOCK) {
        float4 kk[DU2][4];
        bool on[DU2];
        float rn[DU2];
#pragma unroll
        for (int u = 0; u < DU2; ++u) {
            const int p = p0 + u * BLOCK;
            const int pc = p < M * H ? p : M * H - 1;
            const int n = pc / H, h = pc - n * H;
            on[u] = p < M * H && l.msk[n] != 0;
            rn[u] = SD ? rem[n] : 0.0f;
            const float* kp = K + (int64_t)n * ld + h * D;
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4)
                kk[u][d4] = on[u] ? *reinterpret_cast<const float4*>(kp + 4 * d4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < DU2; ++u) {
            const int p = p0 + u * BLOCK;
            if (p >= M * H) break;
            const int n = p / H, h = p - n * H;
            const float* qp = l.q + h * QS;
            float acc = 0.0f;
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) {
                acc = fma_(qp[4 * d4], kk[u][d4].x, acc);
                acc = fma_(qp[4 * d4 + 1], kk[u][d4].y, acc);
                acc = fma_(qp[4 * d4 + 2], kk[u][d4].z, acc);
                acc = fma_(qp[4 * d4 + 3], kk[u][d4].w, acc);
            }
            if (SD) acc = fma_(rn[u], sd_qw[h], acc);
            l.w[h * M + n] = on[u] ? acc * qk_scale : -INFINITY;
        }
    }
        __syncthreads();
    } else {
    for (int p = tid; p < M * H; p += BLOCK) {
        const int n = p / H, h = p - n * H;
        float sc = -INFINITY;
        if (l.msk[n]) {
            const float* kp = K + (int64_t)n * ld + h * D;
            const float* qp = l.q + h * QS;
            float acc = 0.0f;
            for (int d = 0; d < D; d += 4) {
                float4 kk = *reinterpret_cast<const float4*>(kp + d);
                acc = fma_(qp[d], kk.x, acc);
                acc = fma_(qp[d + 1], kk.y, acc);
                acc = fma_(qp[d + 2], kk.z, acc);
                acc = fma_(qp[d + 3], kk.w, acc);
            }
            if (SD) acc = fma_(rem[n], sd_qw[h], acc);
            sc = acc * qk_scale;
        }
        l.w[h * M + n] = sc;
    }
        __syncthreads();
    }

    SSTAMP(0);      // D1 + D2 (keys)
    // ---- D3 per-head max, w = exp(s - max) -------------------------------------------------------------
    for (int h = wv; h < H; h += NWAVE) {
        float* wh = l.w + h * M;
        float m = -INFINITY;
        for (int n = lane; n < M; n += 64) m = __builtin_fmaxf(m, wh[n]);
        m = wave_max(m);
        float rs = 0.0f;
        for (int nb = 0; nb < M; nb += 64) {
            const int n = nb + lane;
            const float wn = (n < M && l.msk[n]) ? d_expf(wh[n] - m) : 0.0f;
            if (n < M) wh[n] = wn;
            if (SD) {       // R_h: 64-blocks of w * rem through the lane tree, blocks added left to right
                const float tr = wave_tree_sum(n < M ? wn * rem[n] : 0.0f);
                rs = nb == 0 ? tr : rs + tr;
            }
        }
        if (SD && lane == 0) sd_pr[h] = rs;
    }
    __syncthreads();

    SSTAMP(1);      // D3 (softmax weights)
    // ---- D4 glimpse: NCHUNK node chunks, sequential inside a chunk, chunks added left to right ----------
    const int C = (M + EAMRL_NCHUNK - 1) / EAMRL_NCHUNK;
    {
        // A thread owns four adjacent columns of one chunk (D % 4 == 0 is an entry requirement): float4 value loads, DU4
        // of them in flight.  One column per thread (4-byte loads, all threads busy) is 18 % slower on CVRP-500: the stage
        // is bound by the number of load instructions, not by idle lanes.  A masked node has w == +0 exactly, so adding
        // its terms unconditionally (with v = 0 in place of the skipped load) leaves the sums bit-identical.
        const int E4 = E / 4;
        for (int pp = tid; pp < E4 * EAMRL_NCHUNK; pp += BLOCK) {
            const int g = pp / E4, e = 4 * (pp - g * E4), h = e / D;
            const float* wh = l.w + h * M;
            const int n0 = g * C, n1 = min(M, n0 + C);
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
            // Z_g in the canonical order: four partial sums by position in the chunk ((n - n0) mod 4; DU4 is a multiple of 4, so slot u
            // of a trip is class u & 3: four independent add chains), combined as (P0 + P1) + (P2 + P3)
            static_assert(DU4 % 4 == 0, "DU4");
            float zc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int nb = n0; nb < n1; nb += DU4) {
                float4 vv[DU4];
                float ww[DU4];
#pragma unroll
                for (int u = 0; u < DU4; ++u) {
                    const int n = nb + u;
                    const bool onv = n < n1 && l.msk[n] != 0;
                    vv[u] = onv ? *reinterpret_cast<const float4*>(V + (int64_t)n * ld + e) : make_float4(0.f, 0.f, 0.f, 0.f);
                    ww[u] = onv ? wh[n] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < DU4; ++u) {
                    zc[u & 3] = zc[u & 3] + ww[u];        // (slots beyond n1 and masked nodes add 0)
                    a0 = fma_(ww[u], vv[u].x, a0);
                    a1 = fma_(ww[u], vv[u].y, a1);
                    a2 = fma_(ww[u], vv[u].z, a2);
                    a3 = fma_(ww[u], vv[u].w, a3);
                }
            }
            float* pa = l.partA + g * E + e;      // (partA is only 4-byte aligned for odd M)
            pa[0] = a0; pa[1] = a1; pa[2] = a2; pa[3] = a3;
            if (e - h * D == 0) l.partZ[g * H + h] = (zc[0] + zc[1]) + (zc[2] + zc[3]);
        }
    }
    __syncthreads();
    for (int e = tid; e < E; e += BLOCK) {
        const int h = e / D;
        float A = l.partA[e], Z = l.partZ[h];
#pragma unroll
        for (int g = 1; g < EAMRL_NCHUNK; ++g) { A = A + l.partA[g * E + e]; Z = Z + l.partZ[g * H + h]; }
        if (SD) A = fma_(sd_pr[h], dyn[E + e], A);
        l.heads[(e / (E / EAMRL_NCHUNK)) * HS + (e % (E / EAMRL_NCHUNK))] = A / Z;
    }
    __syncthreads();
    const int EC0 = E / EAMRL_NCHUNK;
    if (SD) {       // heads_c . lw_c, one thread per column chunk
        if (tid < EAMRL_NCHUNK) {
            float hl = 0.0f;
            for (int e = tid * EC0; e < (tid + 1) * EC0; ++e) hl = fma_(l.heads[tid * HS + (e - tid * EC0)], dyn[2 * E + e], hl);
            sd_hl[tid] = hl;
        }
        __syncthreads();
    }

    SSTAMP(2);      // D4 (values) + heads
    // ---- D5 logit partials over NCHUNK column chunks ---------------------------------------------------------
    const int EC = E / EAMRL_NCHUNK;
    if (EC == 32) {
        // DU5 (node, column chunk) pairs per trip: 8 * DU5 float4 loads in flight
        for (int p0 = tid; p0 < M * EAMRL_NCHUNK; p0 += DU5 * BLOCK) {
            float4 lv[DU5][8];
            bool on[DU5];
            float rn[DU5];
#pragma unroll
            for (int u = 0; u < DU5; ++u) {
                const int p = p0 + u * BLOCK;
                const int pc = p < M * EAMRL_NCHUNK ? p : M * EAMRL_NCHUNK - 1;
                const int n = pc / EAMRL_NCHUNK, c = pc - n * EAMRL_NCHUNK;
                on[u] = p < M * EAMRL_NCHUNK && l.msk[n] != 0;
                rn[u] = (SD && on[u]) ? rem[n] : 0.0f;
                const float* lp = Lp + (int64_t)n * ld + c * EC;
#pragma unroll
                for (int e4 = 0; e4 < 8; ++e4)
                    lv[u][e4] = on[u] ? *reinterpret_cast<const float4*>(lp + 4 * e4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < DU5; ++u) {
                const int p = p0 + u * BLOCK;
                if (p >= M * EAMRL_NCHUNK) break;
                const int c = p % EAMRL_NCHUNK;
                const float* hp = l.heads + c * HS;
                float cg = 0.0f;
#pragma unroll
                for (int e4 = 0; e4 < 8; ++e4) {
                    cg = fma_(hp[4 * e4], lv[u][e4].x, cg);
                    cg = fma_(hp[4 * e4 + 1], lv[u][e4].y, cg);
                    cg = fma_(hp[4 * e4 + 2], lv[u][e4].z, cg);
                    cg = fma_(hp[4 * e4 + 3], lv[u][e4].w, cg);
                }
                if (SD) cg = fma_(rn[u], sd_hl[c], cg);
                l.partL[p] = cg;       // masked node: all-zero operands -> chain of exact zeros, as before
            }
        }
    } else {
        for (int p = tid; p < M * EAMRL_NCHUNK; p += BLOCK) {
            const int n = p / EAMRL_NCHUNK, c = p - n * EAMRL_NCHUNK;
            float cg = 0.0f;
            if (l.msk[n]) {
                const float* lp = Lp + (int64_t)n * ld + c * EC;
                const float* hp = l.heads + c * HS;
                for (int e = 0; e < EC; e += 4) {
                    float4 v = *reinterpret_cast<const float4*>(lp + e);
                    cg = fma_(hp[e], v.x, cg);
                    cg = fma_(hp[e + 1], v.y, cg);
                    cg = fma_(hp[e + 2], v.z, cg);
                    cg = fma_(hp[e + 3], v.w, cg);
                }
                if (SD) cg = fma_(rem[n], sd_hl[c], cg);
            }
            l.partL[p] = cg;
        }
    }
    __syncthreads();

    SSTAMP(3);      // D5 (logit keys)
    // ---- D6 combine, /sqrt(E), tanh clip, mask, /temperature ------------------------------------------------------
    const float inv_sqrtE = 1.0f / __builtin_sqrtf((float)E);      // one rounded constant (canonical: logit = u * inv_sqrtE)
    float tmax = -INFINITY;
    bool nan_seen = false;
    for (int n = tid; n < M; n += BLOCK) {
        const float* pl = l.partL + n * EAMRL_NCHUNK;
        float u = pl[0];
#pragma unroll
        for (int c = 1; c < EAMRL_NCHUNK; ++c) u = u + pl[c];
        float logit = u * inv_sqrtE;
        const bool feas = l.msk[n] != 0;
        if (feas && logit != logit) nan_seen = true;
        if (logits_row) logits_row[n] = feas ? logit : -INFINITY;
        float v = (a.clip > 0.0f) ? d_tanhf(logit) * a.clip : logit;
        if (!feas) v = -INFINITY;
        v = v / a.temp;
        l.x[n] = v;
        tmax = __builtin_fmaxf(tmax, v);
    }
    if (nan_seen) atomicOr(a.status, EAMRL_ST_NAN_LOGITS);
    float mx = block_max(tmax, l.red);  // (syncs inside: partL reads above are complete)

    // ---- D6b top-k / top-p filtering of the scaled logits (process_logits, rl4co/utils/decoding.py:110-136,170-176) -----
    const bool use_topp = a.top_p > 0.0f && a.top_p < 1.0f;
    if (a.top_k > 0 || use_topp) {
        float* pp = l.partL;                      // [M] keep flags, then probabilities
        float* srt = l.partL + M;                 // [M] probabilities in ascending (value, index) order -> remove flags
        int* rk = reinterpret_cast<int*>(l.partL + 2 * M);
        if (a.top_k > 0) {                        // keep n iff fewer than k entries are strictly larger (ties kept)
            const int k = a.top_k < M ? a.top_k : M;
            for (int n = tid; n < M; n += BLOCK) {
                const float xn = l.x[n];
                int cnt = 0;
                for (int m = 0; m < M; ++m) cnt += (l.x[m] > xn);
                pp[n] = (cnt < k) ? 1.0f : 0.0f;
            }
            __syncthreads();
            for (int n = tid; n < M; n += BLOCK) if (pp[n] == 0.0f) l.x[n] = -INFINITY;
            __syncthreads();
        }
        if (use_topp) {                           // nucleus: drop the lower tail whose running probability <= 1 - top_p
            const float thr = (float)(1.0 - (double)a.top_p);
            float t2 = -INFINITY;
            for (int n = tid; n < M; n += BLOCK) t2 = __builtin_fmaxf(t2, l.x[n]);
            const float m2 = block_max(t2, l.red);
            for (int n = tid; n < M; n += BLOCK) pp[n] = (l.x[n] > -INFINITY) ? d_expf(l.x[n] - m2) : 0.0f;
            __syncthreads();
            const int nb2 = (M + 63) / 64;
            for (int b = wv; b < nb2; b += NWAVE) {
                const int n = b * 64 + lane;
                const float v = wave_tree_sum(n < M ? pp[n] : 0.0f);
                if (lane == 0) l.red[8 + b] = v;
            }
            __syncthreads();
            float Z = l.red[8];
            for (int b = 1; b < nb2; ++b) Z = Z + l.red[8 + b];
            for (int n = tid; n < M; n += BLOCK) {
                const float xn = l.x[n];
                int r0 = 0;
                for (int m = 0; m < M; ++m) { const float xm = l.x[m]; r0 += (xm < xn) | ((xm == xn) & (m < n)); }
                rk[n] = r0;
                srt[r0] = pp[n] / Z;
            }
            __syncthreads();
            if (tid == 0) {                       // the running sum is sequential by definition
                float cs = 0.0f;
                for (int j = 0; j < M; ++j) { cs = cs + srt[j]; srt[j] = (cs <= thr) ? 1.0f : 0.0f; }
            }
            __syncthreads();
            for (int n = tid; n < M; n += BLOCK) if (srt[rk[n]] != 0.0f) l.x[n] = -INFINITY;
            __syncthreads();
        }
        float t3 = -INFINITY;
        for (int n = tid; n < M; n += BLOCK) t3 = __builtin_fmaxf(t3, l.x[n]);
        mx = block_max(t3, l.red);
    }

    // ---- D7 log-softmax: lane-tree sum of exp(x - max); an entry takes part iff it is still finite --------------------------
    float* ex = l.partL;  // reuse
    for (int n = tid; n < M; n += BLOCK) ex[n] = (l.x[n] > -INFINITY) ? d_expf(l.x[n] - mx) : 0.0f;
    __syncthreads();
    const int nblk = (M + 63) / 64;
    for (int b = wv; b < nblk; b += NWAVE) {
        const int n = b * 64 + lane;
        float v = wave_tree_sum(n < M ? ex[n] : 0.0f);
        if (lane == 0) l.red[8 + b] = v;   // nblk <= 56
    }
    __syncthreads();
    float Zl = l.red[8];
    for (int b = 1; b < nblk; ++b) Zl = Zl + l.red[8 + b];
    const float lse = d_logf(Zl);
    __syncthreads();  // everyone has read red[] / ex[] before they are reused

    // ---- D8 log-probs + selection -----------------------------------------------------------------------------------------------
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int n = tid; n < M; n += BLOCK) {
        const bool feas = l.x[n] > -INFINITY;          // feasible and not filtered out
        float lpn = feas ? (l.x[n] - mx) - lse : -INFINITY;
        l.x[n] = lpn;
        if (logprobs_row) logprobs_row[n] = lpn;
        float key = lpn;
        if (a.mode == EAMRL_SAMPLE) key = d_expf(lpn) / noise_row[n];
        if (key > best || besti == 0x7fffffff) { best = key; besti = n; }  // n ascending per thread: first max kept
    }
    wave_argmax(best, besti);
    if (lane == 0) { l.red[wv] = best; l.redi[wv] = besti; }
    __syncthreads();
    int sel = l.redi[0];
    {
        float bv = l.red[0];
#pragma unroll
        for (int i = 1; i < NWAVE; ++i) {
            float ov = l.red[i];
            int oi = l.redi[i];
            if (ov > bv || (ov == bv && oi < sel)) { bv = ov; sel = oi; }
        }
    }
    if (a.mode == EAMRL_EVALUATE) sel = (int)given;
    const bool in_range = sel >= 0 && sel < M;
    if (!in_range || !l.msk[sel]) {
        if (tid == 0) atomicOr(a.status, EAMRL_ST_INFEASIBLE);
        if (!in_range) sel = 0;
    }
    out_a = sel;
    out_lp = l.x[sel];
    SSTAMP(4);      // D6-D8 (finish)
}

// Apply the env transition to the LDS copy of the row (msk, vis for CVRP, rem for SDVRP) and the uniform state.
template <int ENV>
__device__ bool env_step_row(const DecArgs& a, const RowLds& l, uint8_t* vis, float* rem, int64_t r, RowState& s,
                             int64_t act)
{
    const int tid = threadIdx.x;
    const int M = a.M;
    __syncthreads();  // all readers of msk / x are done
    if (ENV == EAMRL_ENV_SDVRP) {
        // SDVRPEnv._step + get_action_mask (sdvrp/env.py:58-92,137-146): deliver min(remaining demand, free capacity)
        const float sel = rem[act];
        const float free_cap = s.vcap - s.used;
        const float delivered = sel < free_cap ? sel : free_cap;
        s.used = (s.used + delivered) * (act != 0 ? 1.0f : 0.0f);
        s.cur = act;
        __syncthreads();  // everyone has read rem[act]
        if (tid == 0) rem[act] = sel + (-delivered);
        __syncthreads();
        const bool full = s.used >= s.vcap;
        int any_free = 0, any_rem = 0;
        for (int n = tid; n < M; n += BLOCK) {
            const float rv = rem[n];
            any_rem |= rv > 0.0f;
            if (n >= 1) {
                const int blocked = (rv == 0.0f) | full;
                l.msk[n] = !blocked;
                any_free |= !blocked;
            }
        }
        any_free = __syncthreads_or(any_free);
        any_rem = __syncthreads_or(any_rem);
        if (tid == 0) l.msk[0] = !((s.cur == 0) && any_free);
        return any_rem == 0;
    } else if (ENV == EAMRL_ENV_OP) {
        // OPEnv._step + get_action_mask (op/env.py:69-102,149-165); a.demand = per-node arrival limit, s.used = tour length
        const float* L = a.locs + (r % a.B) * (int64_t)M * 2;
        const float* ml = a.demand + (r % a.B) * (int64_t)M;
        const float cx = L[2 * act], cy = L[2 * act + 1];
        {
            const float dx = cx - L[2 * s.cur], dy = cy - L[2 * s.cur + 1];
            s.used = s.used + __builtin_sqrtf(fma_(dy, dy, dx * dx));
        }
        const bool done = (act == 0) && (s.istep > 0);
        s.cur = act;
        s.istep += 1;
        if (tid == 0) vis[act] = 1;
        __syncthreads();
        const int v0 = vis[0] != 0;
        for (int n = 1 + tid; n < M; n += BLOCK) {
            const float dx = L[2 * n] - cx, dy = L[2 * n + 1] - cy;
            const int exceeds = (s.used + __builtin_sqrtf(fma_(dy, dy, dx * dx))) > ml[n];
            l.msk[n] = !((vis[n] != 0) | v0 | exceeds);
        }
        if (tid == 0) l.msk[0] = 1;
        return done;
    } else if (ENV == EAMRL_ENV_PCTSP) {
        // PCTSPEnv._step + get_action_mask (pctsp/env.py:64-97,156-163); a.demand = real_prize [B][M]
        const float* prize = a.demand + (r % a.B) * M;
        s.used = s.used + prize[act];
        const bool done = (s.istep > 0) && (act == 0);
        s.cur = act;
        s.istep += 1;
        if (tid == 0) vis[act] = 1;
        __syncthreads();
        const int v0 = vis[0] != 0;
        int unvisited = 0;
        for (int n = 1 + tid; n < M; n += BLOCK) {
            const int v = vis[n] != 0;
            l.msk[n] = !(v | v0);
            unvisited |= !v;
        }
        unvisited = __syncthreads_or(unvisited);
        if (tid == 0) l.msk[0] = !((s.used < 1.0f) && unvisited);
        return done;
    } else if (ENV == EAMRL_ENV_TSP) {
        if (s.istep == 0) s.first = act;
        s.cur = act;
        s.istep += 1;
        if (tid == 0) l.msk[act] = 0;
        __syncthreads();
        int any = 0;
        for (int n = tid; n < M; n += BLOCK) any |= l.msk[n];
        return __syncthreads_or(any) == 0;
    } else {      // CVRP, and CVRPTW = CVRP + clock (cvrptw/env.py:103-138)
        constexpr bool TW = ENV == EAMRL_ENV_CVRPTW;
        const int N = M - 1;
        const int64_t bi = r % a.B;
        const float* dem = a.demand + bi * N;
        const float* L = TW ? a.locs + bi * (int64_t)M * 2 : nullptr;
        const float* W = TW ? a.tw + bi * (int64_t)M * 2 : nullptr;
        float cx = 0.0f, cy = 0.0f;
        if (TW) {
            cx = L[2 * act]; cy = L[2 * act + 1];
            const float dx = L[2 * s.cur] - cx, dy = L[2 * s.cur + 1] - cy;
            const float arrive = s.now + __builtin_sqrtf(fma_(dy, dy, dx * dx));
            const float start = arrive > W[2 * act] ? arrive : W[2 * act];
            s.now = (act != 0 ? 1.0f : 0.0f) * (start + a.dur[bi * (int64_t)M + act]);
        }
        int64_t di = act - 1;
        di = di < 0 ? 0 : (di > N - 1 ? N - 1 : di);
        s.used = (s.used + dem[di]) * (act != 0 ? 1.0f : 0.0f);
        s.cur = act;
        if (tid == 0) vis[act] = 1;
        __syncthreads();
        const float lim = s.vcap + 1e-5f;
        int any_free = 0, all_vis = vis[0] != 0;
        for (int j = tid; j < N; j += BLOCK) {
            const int v = vis[j + 1] != 0;
            const float load = dem[j] + s.used;
            const int blocked = v | (load > lim);
            int ok = !blocked;
            if (TW) {       // reachable before the window closes
                const float dx = cx - L[2 * (j + 1)], dy = cy - L[2 * (j + 1) + 1];
                ok &= (s.now + __builtin_sqrtf(fma_(dy, dy, dx * dx))) <= W[2 * (j + 1) + 1];
            }
            l.msk[j + 1] = ok;
            any_free |= !blocked;        // the depot rule looks at the CVRP mask only
            all_vis &= v;
        }
        any_free = __syncthreads_or(any_free);
        all_vis = __syncthreads_and(all_vis);
        if (tid == 0) {
            int ok0 = !((s.cur == 0) && any_free);
            if (TW) {
                const float dx = cx - L[0], dy = cy - L[1];
                ok0 &= (s.now + __builtin_sqrtf(fma_(dy, dy, dx * dx))) <= W[1];
            }
            l.msk[0] = ok0;
        }
        return all_vis != 0;
    }
}

template <int ENV>
__device__ __forceinline__ void load_row_state(const DecArgs& a, int64_t r, RowState& s)
{
    s.cur = a.cur[r];
    if (ENV == EAMRL_ENV_TSP) {
        s.first = a.first[r];
        s.istep = a.istep[r];
        s.used = 0.0f;
        s.vcap = 0.0f;
    } else {
        s.first = 0;
        s.istep = (ENV == EAMRL_ENV_PCTSP || ENV == EAMRL_ENV_OP) ? a.istep[r] : 1;
        s.used = a.used[r];
        s.vcap = a.vcap[r];
    }
    s.now = (ENV == EAMRL_ENV_CVRPTW) ? a.time[r] : 0.0f;
}

template <int ENV>
__device__ __forceinline__ void store_row_state(const DecArgs& a, const RowLds& l, const uint8_t* vis, const float* rem,
                                                int64_t r, const RowState& s, bool done)
{
    const int tid = threadIdx.x;
    __syncthreads();
    for (int n = tid; n < a.M; n += BLOCK) {
        a.mask[r * a.M + n] = l.msk[n];
        if (ENV == EAMRL_ENV_CVRP || ENV == EAMRL_ENV_PCTSP || ENV == EAMRL_ENV_OP || ENV == EAMRL_ENV_CVRPTW)
            a.visited[r * a.M + n] = vis[n];
        if (ENV == EAMRL_ENV_SDVRP) a.rem[r * a.M + n] = rem[n];
    }
    if (tid == 0) {
        a.cur[r] = s.cur;
        a.done[r] = done ? 1 : 0;
        if (ENV == EAMRL_ENV_TSP) { a.first[r] = s.first; a.istep[r] = s.istep; }
        else a.used[r] = s.used;
        if (ENV == EAMRL_ENV_PCTSP || ENV == EAMRL_ENV_OP) a.istep[r] = s.istep;
        if (ENV == EAMRL_ENV_CVRPTW) a.time[r] = s.now;
    }
}

// SDVRP rows keep two more LDS arrays behind `vis`: remaining demand [M] and the dynamic vectors [3][E].
template <int ENV>
__device__ __forceinline__ void load_row_lds(const DecArgs& a, const RowLds& l, uint8_t* vis, float* rem, float* dyn,
                                             int64_t r, bool want_vis)
{
    const int tid = threadIdx.x;
    for (int n = tid; n < a.M; n += BLOCK) {
        l.msk[n] = a.mask[r * a.M + n];
        if ((ENV == EAMRL_ENV_CVRP || ENV == EAMRL_ENV_PCTSP || ENV == EAMRL_ENV_OP || ENV == EAMRL_ENV_CVRPTW) && want_vis)
            vis[n] = a.visited[r * a.M + n];
        if (ENV == EAMRL_ENV_SDVRP) rem[n] = a.rem[r * a.M + n];
    }
    if (ENV == EAMRL_ENV_SDVRP)
        for (int e = tid; e < 3 * a.E; e += BLOCK) dyn[e] = a.dyn[e];
}

template <int ENV>
__global__ __launch_bounds__(BLOCK) void k_decode_step(DecArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const RowLds l = carve_row_lds(smem, a.M, a.E, a.H);
    const int Mp = (a.M + 15) & ~15;
    uint8_t* vis = l.msk + Mp;
    float* rem = reinterpret_cast<float*>(vis + Mp);
    float* dyn = rem + Mp;
    const int64_t r = blockIdx.x;
    const int tid = threadIdx.x;
    RowState s;
    load_row_state<ENV>(a, r, s);
    load_row_lds<ENV>(a, l, vis, rem, dyn, r, a.fuse_env != 0);
    __syncthreads();
    int64_t act;
    float lp;
    decode_row<ENV>(a, l, r, s, a.noise ? a.noise + r * a.M : nullptr, a.given ? a.given[r] : 0, act, lp,
                    a.logprobs_all ? a.logprobs_all + r * a.M : nullptr,
                    a.logits_raw ? a.logits_raw + r * a.M : nullptr, rem, dyn);
    if (tid == 0) { a.action[r] = act; a.logp[r] = lp; }
    if (a.fuse_env) {
        const bool done = env_step_row<ENV>(a, l, vis, rem, r, s, act);
        store_row_state<ENV>(a, l, vis, rem, r, s, done);
    }
}

// Whole decode loop for one row per workgroup; K/V/Lp re-read from HBM/L2 each step (works for any M).
template <int ENV>
__global__ __launch_bounds__(BLOCK) void k_rollout_stream(DecArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const RowLds l = carve_row_lds(smem, a.M, a.E, a.H);
    const int Mp = (a.M + 15) & ~15;
    uint8_t* vis = l.msk + Mp;
    float* rem = reinterpret_cast<float*>(vis + Mp);
    float* dyn = rem + Mp;
    const int64_t r = blockIdx.x;
    const int tid = threadIdx.x;
    RowState s;
    load_row_state<ENV>(a, r, s);
    const int64_t i0 = s.istep;
    load_row_lds<ENV>(a, l, vis, rem, dyn, r, true);
    bool done = a.done[r] != 0;
    __syncthreads();
    int t = 0;
    while (!done && t < a.t_max) {
        int64_t act;
        float lp;
        const float* nz = a.noise ? a.noise + (r * a.t_max + t) * (int64_t)a.M : nullptr;
        const int64_t gv = a.given ? (t < a.t_given ? a.given[r * a.t_given + t] : 0) : 0;
        decode_row<ENV>(a, l, r, s, nz, gv, act, lp, nullptr, nullptr, rem, dyn);
        if (tid == 0) { a.action[r * a.t_max + t] = act; a.logp[r * a.t_max + t] = lp; }
        done = env_step_row<ENV>(a, l, vis, rem, r, s, act);
        ++t;
    }
    if (ENV == EAMRL_ENV_PCTSP || ENV == EAMRL_ENV_OP) s.istep = i0;      // k_rollout_pad adds the batch's step count
    store_row_state<ENV>(a, l, vis, rem, r, s, done);
    if (tid == 0) {
        atomicMax(a.steps_out, t);
        if (!done) atomicOr(a.status, EAMRL_ST_STEP_OVERRUN);
    }
}

// After the loop the reference keeps stepping finished CVRP rows with the depot until the slowest row
// is done: rows whose last real action was a customer end at the depot with an
// empty vehicle.  actions/logps are already right-padded with 0; this fixes the state tensors.
// PCTSP / OP: the reference keeps counting `i` for finished rows too, so every row ends at i0 + T; the rollout kernels
// store i0 and this adds the batch's step count.
__global__ void k_rollout_pad(DecArgs a, int env)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.R) return;
    const int T = *a.steps_out;
    if (env == EAMRL_ENV_PCTSP || env == EAMRL_ENV_OP) { a.istep[r] += T; return; }
    if (T <= 0 || !a.done[r]) return;
    // A row that needed fewer than T steps never wrote column T-1 (host pre-zeroed = depot); stepping a
    // finished row with the depot makes (cur, used) = (0, 0) and leaves mask/visited unchanged.
    if (a.cur[r] != 0 && a.action[r * a.t_max + (T - 1)] == 0) {
        a.cur[r] = 0;
        a.used[r] = 0.0f;
        if (env == EAMRL_ENV_CVRPTW) a.time[r] = 0.0f;       // the depot resets the clock
    }
}

void launch_rollout_pad(int env, const DecArgs& a, hipStream_t st)
{
    if (env != EAMRL_ENV_TSP)
        hipLaunchKernelGGL(k_rollout_pad, dim3((unsigned)((a.R + 255) / 256)), dim3(256), 0, st, a, env);
}

static int launch_decode(int env, const DecArgs& a, bool rollout, hipStream_t st)
{
    const size_t Mp = ((size_t)a.M + 15) & ~(size_t)15;
    size_t lds = row_lds_bytes(a.M, a.E, a.H) + Mp;                       // + visited bytes
    if (env == EAMRL_ENV_SDVRP)                                            // + remaining demand + dynamic vectors + folds
        lds += 4 * Mp + 12 * (size_t)a.E + 4 * (2 * (size_t)a.H + EAMRL_NCHUNK);
    if (lds > 160 * 1024) return EAMRL_E_ARG;
    dim3 grid((unsigned)a.R), block(BLOCK);
    void (*k)(DecArgs);
    if (rollout)
        k = env == EAMRL_ENV_TSP ? k_rollout_stream<EAMRL_ENV_TSP>
          : env == EAMRL_ENV_CVRP ? k_rollout_stream<EAMRL_ENV_CVRP>
          : env == EAMRL_ENV_SDVRP ? k_rollout_stream<EAMRL_ENV_SDVRP>
          : env == EAMRL_ENV_PCTSP ? k_rollout_stream<EAMRL_ENV_PCTSP>
          : env == EAMRL_ENV_OP ? k_rollout_stream<EAMRL_ENV_OP> : k_rollout_stream<EAMRL_ENV_CVRPTW>;
    else
        k = env == EAMRL_ENV_TSP ? k_decode_step<EAMRL_ENV_TSP>
          : env == EAMRL_ENV_CVRP ? k_decode_step<EAMRL_ENV_CVRP>
          : env == EAMRL_ENV_SDVRP ? k_decode_step<EAMRL_ENV_SDVRP>
          : env == EAMRL_ENV_PCTSP ? k_decode_step<EAMRL_ENV_PCTSP>
          : env == EAMRL_ENV_OP ? k_decode_step<EAMRL_ENV_OP> : k_decode_step<EAMRL_ENV_CVRPTW>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return EAMRL_E_LAUNCH;
    }
    hipLaunchKernelGGL(k, grid, block, lds, st, a);
    if (rollout) launch_rollout_pad(env, a, st);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_decode_step(int env, const DecArgs& a, hipStream_t st) { return launch_decode(env, a, false, st); }
int launch_rollout_stream(int env, const DecArgs& a, hipStream_t st) { return launch_decode(env, a, true, st); }

}  // namespace eamrl

#ifdef EAMRL_STAMPS
extern "C" __attribute__((visibility("default"))) int eamrl_debug_read_stream_stamps(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(eamrl::g_sstamps), sizeof(eamrl::g_sstamps)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(eamrl::g_sstamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
