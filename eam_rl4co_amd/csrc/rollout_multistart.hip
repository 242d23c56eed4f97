// Start-sharing whole-rollout kernel on fp32 MFMA for multistart batches (POMO: R = S * B rows in "(s b)" order, TSP).
//
// One workgroup of 8 wavefronts owns ONE INSTANCE and rolls out all its S starts together: at every decode step the S
// context queries go against the instance's glimpse keys / values / folded logit keys as dense [S,16]x[16,M], [S,M]x[M,16]
// and [S,128]x[128,M] products on v_mfma_f32_16x16x4_f32 (S queries sharing one K/V/L tile is the fp32-FLOP
// bound case).  K, V and Lp live in registers as MFMA fragments for the whole episode; LDS holds the per-start state (mask
// bit sets, current / first node) and the 16-query tiles being processed.
//
// The arithmetic is the canonical order of DESIGN.md 2, bit for bit what k_rollout_resident / k_decode_step / the oracle
// compute -- the MFMA accumulates its k dimension as an ordered fma chain, so only the tiling had to be arranged:
//   scores   s = chain_d(q, K[n]) / 4: q is pre-scaled by the power of two; S^T = K Q^T tiles (lane = query, registers = keys)
//   glimpse  four node chunks of ceil(M / 4): one accumulator per chunk for A_g = chain_n(w, V); Z_g = (P0 + P1) + (P2 + P3) with
//            P_r the sequential sum of the weights at the chunk's positions r (mod 4) -- a lane holds exactly one such class; a k-step that straddles a chunk boundary is issued for both chunks with
//            the other chunk's weights zeroed (fma(0, v, acc) == acc);  heads = (((A0+A1)+A2)+A3) / (((Z0+Z1)+Z2)+Z3)
//   logits   four column chunks of 32: one accumulator per chunk (8 k-steps each), u = ((c0+c1)+c2)+c3
//   finish   (round 3) the logit tiles are transposed through LDS so that a HALF-WAVEFRONT owns all keys of one query (lane =
//            four consecutive keys): u * (1 / sqrt(E)), 10 tanh, mask, / temperature, log-softmax with the lane tree (two
//            in-lane adds, four DPP levels = a 64-key block per 16 lanes, the two blocks added), greedy = first node equal to the
//            maximum, sampling = first node with the largest p / noise, and the env transition of that start -- all reductions
//            inside the half-wavefront, no barrier between them (it was four cross-wave reductions with a barrier each, on seven
//            of the eight wavefronts)
// Noise: the caller's [R][t_max][M] tensor, or computed in place from (seed, row, step, node) (exp1_noise4, dmath.hpp).
//
// Reference loop replaced: rl4co/models/common/constructive/base.py:236-250 with the multistart layout of
// rl4co/utils/decoding.py:284-344 and rl4co/models/zoo/am/decoder.py:183-198 (K/V/L shared by the S queries of an instance).
#include "kernels.hpp"

namespace eamrl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
// a * b for operands below 2^32 (rows, steps per row: rollout_ms_mfma_supports): one v_mad_u64_u32 where the 64 x 64 product of two
// int64 is three quarter-rate multiplies -- per-lane row offsets are computed once per tile in the finish phase
__device__ __forceinline__ int64_t mul32w(int64_t a, int64_t b) { return (int64_t)((uint64_t)(uint32_t)a * (uint32_t)b); }

#ifdef EAMRL_STAMPS   // development build only (tools/build_stamps.sh): per-phase cycle sums over all wavefronts
__device__ unsigned long long g_ms_stamps[24];
#define MSTAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); st_acc[i] += now_ - st_t; st_t = now_; } while (0)
#else
#define MSTAMP(i) do { } while (0)
#endif

namespace {

constexpr int ME = 128, MH = 8;
constexpr int TS = 140, TG = 34;      // A-layout tile buffers [16][TS]: element (j, c) at j * TS + (c & 3) * TG + (c >> 2)
constexpr int SMAX = 128;             // starts per instance

__device__ __forceinline__ f32x4 mf(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 z4() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }

// combine over the four lane groups (lanes l, l^16, l^32, l^48): first the ^16 partner, then the ^32 one -- the order of the
// lane tree's levels 4 and 8 for keys laid out 4 per lane
__device__ __forceinline__ float group_max(float v)
{
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = vmax_raw(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return vmax_raw(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float group_sum(float v)
{
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// Sum over the four lane groups of one chunk's partial sums in the canonical order: lane group G holds the nodes 4 t + G, i.e. the
// chunk's position class (G - start) & 3, and Z_g = (P0 + P1) + (P2 + P3) over POSITION classes.  Even chunk start: the pairs are
// the lane groups {G, G ^ 1} (group_sum).  Odd start: {G, G ^ 3} = lanes l and l ^ 48, then the two pair sums (partner l ^ 16).
// (v_permlane32_swap(x, x): result 0 = x of the lower half in both halves, result 1 = the upper half's; v_permlane16_swap alike per row pair.)
__device__ __forceinline__ float group_sum_odd(float v, int lane)
{
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const uint32_t p32 = (lane & 32) ? b[0] : b[1];                      // v of lane l ^ 32
    auto a = __builtin_amdgcn_permlane16_swap(p32, p32, false, false);
    const uint32_t p48 = (lane & 16) ? a[0] : a[1];                      // ... of lane l ^ 48
    const float s = v + __uint_as_float(p48);
    auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(c[0]) + __uint_as_float(c[1]);
}
__device__ __forceinline__ int group_min(int v)
{
    auto a = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    v = min((int)a[0], (int)a[1]);
    auto b = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return min((int)b[0], (int)b[1]);
}

// Reductions over the 32 lanes of a half-wavefront (both halves at once): the four DPP levels inside a 16-lane row, then the two
// rows of the half through v_permlane16_swap (result 0 = the even row's value, 1 = the odd row's, in both rows).
__device__ __forceinline__ float half_max(float v)
{
    EAMRL_MAX_DPP(v, "quad_perm:[1,0,3,2]");
    EAMRL_MAX_DPP(v, "quad_perm:[2,3,0,1]");
    EAMRL_MAX_DPP(v, "row_half_mirror");
    EAMRL_MAX_DPP(v, "row_mirror");
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return vmax_raw(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// the lane tree over keys laid out four per lane: levels 4 .. 32 inside a row (one 64-key block per 16 lanes), then block 0 + block 1
__device__ __forceinline__ float half_tree_sum(float v)
{
    v = v + dpp_f<DPP_XOR1>(v);
    v = v + dpp_f<DPP_XOR2>(v);
    v = v + dpp_f<DPP_HALF_MIRROR>(v);
    v = v + dpp_f<DPP_MIRROR>(v);
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ int half_min(int v)
{
    v = min(v, dpp_i<DPP_XOR1>(v));
    v = min(v, dpp_i<DPP_XOR2>(v));
    v = min(v, dpp_i<DPP_HALF_MIRROR>(v));
    v = min(v, dpp_i<DPP_MIRROR>(v));
    auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return min((int)r[0], (int)r[1]);
}

// CC: the node-chunk length ceil(M / 4) as a compile-time value (0: run time).  With it every "does k-step t touch chunk g"
// decision folds away; left to run time the ~100 uniform conditions are hoisted out of the loops into SGPRs that spill
// (measured: 256 VGPRs + scratch vs 167 VGPRs), so the common sizes get their own instantiation.
// ENV: EAMRL_ENV_TSP, _CVRP, and (round 3) _CVRPTW, _PCTSP, _OP (state machine of the starts: bit-set masks either way; the depot envs add the visited set, the
// load of the vehicle and the per-step mask recomputation, cvrp/env.py:68-144).
template <int RTT, int CC, int ENV>
__global__ __launch_bounds__(512, 2) void k_rollout_ms_mfma(DecArgs a, int S, int nsplit)
{
    constexpr bool TW = ENV == EAMRL_ENV_CVRPTW;        // CVRP + clock and time windows
    constexpr bool CV = ENV == EAMRL_ENV_CVRP || TW;    // vehicle load against the demands
    constexpr bool PC = ENV == EAMRL_ENV_PCTSP;         // prize collecting: used = collected prize, cap = required prize
    constexpr bool OP = ENV == EAMRL_ENV_OP;            // orienteering: used = tour length, s_dem = arrival limit per node
    constexpr bool SD = ENV == EAMRL_ENV_SDVRP;         // split delivery: the state is the remaining-demand row, and the decoder adds
                                                        // rem[n] * (wk | wv | lw) to the node's key / value / logit key
    constexpr bool DEP = ENV != EAMRL_ENV_TSP;          // depot envs: visited set, half-wavefront transition
    constexpr bool XY = OP || TW;                       // the transition needs distances
    __shared__ __attribute__((aligned(16))) float QT[16 * TS];
    __shared__ __attribute__((aligned(16))) float HT[16 * TS];
    constexpr int US = 16 * RTT + 4;                                // row stride of UT (16-byte aligned rows)
    __shared__ __attribute__((aligned(16))) float UT[16 * US];      // raw logit sums u[query][key] of the tile (transposed tiles)
    __shared__ float LPSEL[16];                                     // log-prob of each query's pick
    __shared__ __attribute__((aligned(16))) uint32_t s_bits[SMAX][4];
    __shared__ int s_cur[SMAX], s_first[SMAX], s_istep[SMAX], s_cnt[SMAX], s_done[SMAX];
    __shared__ __attribute__((aligned(16))) uint32_t s_vis[DEP ? SMAX : 1][4];      // depot envs: visited nodes (depot = bit 0)
    __shared__ float s_used[DEP ? SMAX : 1], s_cap[DEP ? SMAX : 1], s_dem[DEP ? 128 : 1];  // load / prize / length, its bound; per-node
                                                                                           // demand (CVRP), prize (PCTSP), arrival limit (OP)
    __shared__ float s_time[TW ? SMAX : 1];                                          // CVRPTW: clock of each start
    __shared__ int s_i0[(PC || OP) ? SMAX : 1];                                      // PCTSP / OP: the step counter the launch started from
    __shared__ float s_xy[XY ? 256 : 2], s_tw0[TW ? 128 : 1], s_tw1[TW ? 128 : 1], s_dur[TW ? 128 : 1];   // coordinates, windows, service
    __shared__ uint32_t s_flags;
    __shared__ __attribute__((aligned(16))) float s_dyn[SD ? 3 * ME : 4];           // SDVRP: wk | wv | lw
    extern __shared__ __attribute__((aligned(16))) float LPF[];    // [RTT waves][32 k-steps][64 lanes]: the logit-key (Lp) A fragments
                                                                    // (kept in LDS, lane-linear: 32 VGPRs fewer per wave)
    // SDVRP: remaining demands of every start, [SMAX][16 RTT] in accumulator order -- node n at (n >> 4) * 16 + (n & 3) * 4 + ((n & 15) >> 2),
    // so that lane (start j, G) of the glimpse reads the nodes 16 kt + 4 r + G (r = 0..3) of its registers as one float4
    float* SREM = LPF + RTT * 32 * 64;
    auto rperm = [](int n) -> int { return (n >> 4) * 16 + (n & 3) * 4 + ((n & 15) >> 2); };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, G = lane >> 4, pi = 4 * (j & 3) + (j >> 2);
    const int M = a.M;
    // nsplit workgroups share an instance (small batches: fills the CUs a 64-instance batch would leave idle): workgroup
    // `part` owns the query tiles [qt0, qt1) -- tiles never interact, each start's state is private to its tile
    const int64_t b = blockIdx.x / nsplit;
    const int part = (int)(blockIdx.x - b * nsplit);
    const int64_t ld = a.ld;
    const int C = CC > 0 ? CC : (M + EAMRL_NCHUNK - 1) / EAMRL_NCHUNK;
    const float inv_sqrtE = 1.0f / __builtin_sqrtf((float)ME);     // one rounded constant (canonical: logit = u * inv_sqrtE)
    const uint64_t seed = a.seed ^ ((a.use_rng && a.seed_dev) ? *a.seed_dev : 0ull);
    const int b1 = C, b2 = 2 * C, b3 = 3 * C;      // node chunk boundaries
    uint64_t cgbits = 0;                            // chunk id (2 bits) of this lane's node 4 t + G, t = 0 .. 4 RTT - 1
#pragma unroll
    for (int t = 0; t < 4 * RTT; ++t) {
        const int n = 4 * t + G;
        cgbits |= (uint64_t)((n >= b1) + (n >= b2) + (n >= b3)) << (2 * t);
    }

    // ---- instance operands as MFMA fragments (once) ------------------------------------------------------------------------
    // wave h: kf[kt][t'] = K[16 kt + pi(j)][16 h + 4 t' + G] (scores A operand, keys in pi order so that accumulator register r
    //         of lane group G is key 16 kt + 4 r + G = the B operand of value k-step 4 kt + r);  vtf[t] = V[4 t + G][16 h + j]
    // wave w < RTT: lpf[t] = Lp[16 w + j][4 t + G] (logit A operand, keys in natural order)
    float kf[RTT][4], vtf[4 * RTT];
#pragma unroll
    for (int kt = 0; kt < RTT; ++kt) {
        const int n = 16 * kt + pi;
#pragma unroll
        for (int t = 0; t < 4; ++t) kf[kt][t] = n < M ? a.K[(b * M + n) * ld + 16 * wv + 4 * t + G] : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < 4 * RTT; ++t) {
        const int n = 4 * t + G;
        vtf[t] = n < M ? a.V[(b * M + n) * ld + 16 * wv + j] : 0.0f;
    }
    float* lpf = LPF + wv * 32 * 64 + lane;       // lpf[64 * t]  (kept in LDS: with the ~165 VGPRs of the compile-time-chunk
                                                  // variants they would fit in registers, measured 1 % slower -- r03j)
    if (wv < RTT) {
        const int n = 16 * wv + j;
#pragma unroll
        for (int t = 0; t < 32; ++t) lpf[64 * t] = n < M ? a.Lp[(b * M + n) * ld + 4 * t + G] : 0.0f;
    }
    // ---- per-start state ----------------------------------------------------------------------------------------------------
    const int nqt_all = (S + 15) >> 4;
    const int qt0 = nqt_all * part / nsplit, qt1 = nqt_all * (part + 1) / nsplit;
    const int s_lo = 16 * qt0, s_hi = min(S, 16 * qt1);
    for (int s = s_lo + tid; s < s_hi; s += blockDim.x) {
        const int64_t r = (int64_t)s * a.B + b;
        uint32_t w[4] = {0, 0, 0, 0};
        int cnt = 0;
        for (int n = 0; n < M; ++n)
            if (a.mask[r * M + n]) { w[n >> 5] |= 1u << (n & 31); ++cnt; }
        s_bits[s][0] = w[0]; s_bits[s][1] = w[1]; s_bits[s][2] = w[2]; s_bits[s][3] = w[3];
        s_cur[s] = (int)a.cur[r];
        if (SD) {
            for (int n = 0; n < 16 * RTT; ++n) SREM[s * (16 * RTT) + rperm(n)] = n < M ? a.rem[r * M + n] : 0.0f;
            s_used[s] = a.used[r];
            s_cap[s] = a.vcap[r];
        } else if (DEP) {
            uint32_t v[4] = {0, 0, 0, 0};
            cnt = 0;                                   // CVRP: nodes visited so far (the episode ends at M, depot included);
            for (int n = 0; n < M; ++n)                // PCTSP: customers visited so far
                if (a.visited[r * M + n]) { v[n >> 5] |= 1u << (n & 31); if (!(PC && n == 0)) ++cnt; }
            s_vis[s][0] = v[0]; s_vis[s][1] = v[1]; s_vis[s][2] = v[2]; s_vis[s][3] = v[3];
            s_used[s] = a.used[r];
            s_cap[s] = a.vcap[r];
            if (TW) s_time[s] = a.time[r];
            if (PC || OP) { s_istep[s] = (int)a.istep[r]; s_i0[s] = (int)a.istep[r]; }
        } else {
            s_first[s] = (int)a.first[r];
            s_istep[s] = (int)a.istep[r];
        }
        s_cnt[s] = cnt;
        s_done[s] = a.done[r] != 0;
    }
    if (SD)
        for (int i = tid; i < 3 * ME; i += blockDim.x) s_dyn[i] = a.dyn[i];
    if (DEP && !SD)
        for (int n = tid; n < 128; n += blockDim.x) {
            if (CV) s_dem[n] = (n >= 1 && n < M) ? a.demand[b * (M - 1) + n - 1] : 0.0f;     // demand of customer n
            else s_dem[n] = n < M ? a.demand[b * M + n] : 0.0f;                                // prize / arrival limit of node n
            if (XY) { s_xy[2 * n] = n < M ? a.locs[(b * M + n) * 2] : 0.0f; s_xy[2 * n + 1] = n < M ? a.locs[(b * M + n) * 2 + 1] : 0.0f; }
            if (TW) {
                s_tw0[n] = n < M ? a.tw[(b * M + n) * 2] : 0.0f;
                s_tw1[n] = n < M ? a.tw[(b * M + n) * 2 + 1] : 0.0f;
                s_dur[n] = n < M ? a.dur[b * M + n] : 0.0f;
            }
        }
    if (tid == 0) s_flags = 0;
    __syncthreads();

    const int nqt = qt1 - qt0;
    int t = 0;
    // thread (jq, e4): four columns 4 e4 .. 4 e4 + 3 of query row jq of a tile
    const int jq = tid >> 5, e4 = tid & 31;
    // this thread's columns of the instance's projection rows / graph context / context vectors as per-lane pointers, made once: a
    // row is then base + node * ld with a full-rate 24-bit multiply (node < 128, ld < 2^24: rollout_ms_mfma_supports) -- the int64
    // (b M + node) ld per load was three quarter-rate multiplies behind SGPRs the loop had to reload from spill lanes
    const float* const pa_b = a.Pa + b * M * ld + 4 * e4;
    const float* const pb_b = a.Pb ? a.Pb + b * M * ld + 4 * e4 : nullptr;
    const float* const gq_b = a.gctx ? a.gctx + b * ME + 4 * e4 : nullptr;
    const float* const cv_b = a.cvec + 4 * e4;
    const uint32_t ld24 = (uint32_t)ld;
    // A tile's query rows in two halves: q_fetch reads the start's state and ISSUES the row loads (branch-free, nothing computed on
    // what they return), q_combine does the arithmetic.  Round 3: as one function, called ahead of the logit phase to hide the
    // loads behind it, the compiler put the additions -- and a wait for every load -- in front of the MFMAs: an L2 latency per
    // tile with the whole workgroup stalled.
    struct QRaw { float4 a, b, g, c2; float fr, now; int fl; };     // fl: bit 0 = the start is live, bit 1 = TSP before the first pick
    auto q_fetch = [&](int qtile) -> QRaw {
        QRaw r;
        const int s = 16 * qtile + jq, sc = s < S ? s : 0;
        // (the start's state words read together: one LDS latency, not one per word)
        const int st_done = s_done[sc], st_cur = s_cur[sc], st_first = DEP ? 0 : s_first[sc], st_istep = DEP ? 1 : s_istep[sc];
        r.fl = (s < S && !st_done) ? 1 : 0;
        r.g = make_float4(0.f, 0.f, 0.f, 0.f);
        r.c2 = r.g;
        r.fr = r.now = 0.0f;
        if (gq_b) r.g = *reinterpret_cast<const float4*>(gq_b);
        const uint32_t curn = (uint32_t)min(max(st_cur, 0), M - 1);
        if (DEP) {
            r.a = *reinterpret_cast<const float4*>(cv_b);
            r.b = *reinterpret_cast<const float4*>(pa_b + __umul24(curn, ld24));
            float fr = s_cap[sc] - s_used[sc];
            if (PC) fr = fr < 0.0f ? 0.0f : fr;
            r.fr = fr;
            if (TW) {
                r.c2 = *reinterpret_cast<const float4*>(cv_b + ME);
                r.now = s_time[sc];
            }
        } else {
            const uint32_t firstn = (uint32_t)min(max(st_first, 0), M - 1);
            if (st_istep == 0) r.fl |= 2;
            r.a = *reinterpret_cast<const float4*>(pa_b + __umul24(firstn, ld24));
            r.b = *reinterpret_cast<const float4*>(pb_b + __umul24(curn, ld24));
            r.c2 = *reinterpret_cast<const float4*>(cv_b);
        }
        return r;
    };
    auto q_combine = [&](const QRaw& r) -> float4 {
        float4 v;
        if (DEP) {              // EnvContext: fma(state column, state scalar, Pa[current]) + graph context -- free capacity (VRPContext),
                                // prize still to collect clamped at 0 (PCTSPContext), length still allowed (OPContext); CVRPTW: + the clock column
            float4 y = make_float4(fma_(r.a.x, r.fr, r.b.x), fma_(r.a.y, r.fr, r.b.y), fma_(r.a.z, r.fr, r.b.z), fma_(r.a.w, r.fr, r.b.w));
            if (TW) y = make_float4(fma_(r.c2.x, r.now, y.x), fma_(r.c2.y, r.now, y.y), fma_(r.c2.z, r.now, y.z), fma_(r.c2.w, r.now, y.w));
            v = make_float4(y.x + r.g.x, y.y + r.g.y, y.z + r.g.z, y.w + r.g.w);
        } else if (r.fl & 2) {
            v = make_float4(r.c2.x + r.g.x, r.c2.y + r.g.y, r.c2.z + r.g.z, r.c2.w + r.g.w);
        } else {
            v = make_float4((r.a.x + r.b.x) + r.g.x, (r.a.y + r.b.y) + r.g.y, (r.a.z + r.b.z) + r.g.z, (r.a.w + r.b.w) + r.g.w);
        }
        return (r.fl & 1) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto q_store = [&](const float4 v) {
        float* p = QT + jq * TS + e4;
        p[0] = 0.25f * v.x; p[TG] = 0.25f * v.y; p[2 * TG] = 0.25f * v.z; p[3 * TG] = 0.25f * v.w;
    };
    const bool pre = nqt > 1;
    bool have_q = false;                // QT already holds the tile about to be processed
#ifdef EAMRL_STAMPS
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_readcyclecounter();
#endif
    for (;;) {
        __syncthreads();                 // the last tile's transition (its done flags) before anyone looks at them
        int active = 0;
        for (int s = s_lo + tid; s < s_hi; s += blockDim.x) active |= !s_done[s];
        if (!__syncthreads_or(active) || t >= a.t_max) break;
        for (int qt = qt0; qt < qt1; ++qt) {
            // ---- q~ tile: 0.25 * ((Pa[first] + Pb[cur]) + gctx), or 0.25 * (c0 + gctx) before the first node is chosen ----------
            // With more than one tile per step the NEXT tile's rows are fetched behind this tile's logit phase and stored
            // before its last barrier (q_next below): a tile's starts are touched by no other tile, so their state is final
            // since their own transition one round earlier, and QT is free once every wave has left the glimpse phase.
            if (!have_q) q_store(q_combine(q_fetch(qt)));
            const int sq = 16 * qt + j;                           // this lane's start (query j of the tile)
            const bool live = sq < S && !s_done[sq];
            uint4 mb = make_uint4(0, 0, 0, 0);
            if (live) mb = *reinterpret_cast<const uint4*>(&s_bits[sq][0]);
            MSTAMP(0);
            __syncthreads();        // QT is published (stored above, or during the previous tile's finish phase)
            MSTAMP(1);
            // ---- glimpse of head wv ------------------------------------------------------------------------------------------
            {
                const float* qp = QT + j * TS + G * TG + 4 * wv;
                const float2 qlo = *reinterpret_cast<const float2*>(qp), qhi = *reinterpret_cast<const float2*>(qp + 2);
                f32x4 s[RTT];
#pragma unroll
                for (int kt = 0; kt < RTT; ++kt) s[kt] = mf(kf[kt][0], qlo.x, z4());
#pragma unroll
                for (int kt = 0; kt < RTT; ++kt) s[kt] = mf(kf[kt][1], qlo.y, s[kt]);
#pragma unroll
                for (int kt = 0; kt < RTT; ++kt) s[kt] = mf(kf[kt][2], qhi.x, s[kt]);
#pragma unroll
                for (int kt = 0; kt < RTT; ++kt) s[kt] = mf(kf[kt][3], qhi.y, s[kt]);
                // SDVRP: s[n] = fma(rem[n], q~_h . wk_h, s[n]) -- the 16-term chain in column order (q~ carries the 1/4 of the scores,
                // a power of two: the products and sums are the canonical ones scaled)
                const float* remq = SREM + (sq < SMAX ? sq : 0) * (16 * RTT) + 4 * G;       // this lane's float4 column, + 16 kt
                if (SD) {
                    float qw = 0.0f;
                    float4 qg[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) qg[g] = *reinterpret_cast<const float4*>(QT + j * TS + g * TG + 4 * wv);   // q~[16 h + 4 i + g]
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float4 wk4 = *reinterpret_cast<const float4*>(s_dyn + 16 * wv + 4 * i);
                        const float qi[4] = {i == 0 ? qg[0].x : i == 1 ? qg[0].y : i == 2 ? qg[0].z : qg[0].w,
                                             i == 0 ? qg[1].x : i == 1 ? qg[1].y : i == 2 ? qg[1].z : qg[1].w,
                                             i == 0 ? qg[2].x : i == 1 ? qg[2].y : i == 2 ? qg[2].z : qg[2].w,
                                             i == 0 ? qg[3].x : i == 1 ? qg[3].y : i == 2 ? qg[3].z : qg[3].w};
                        qw = fma_(qi[0], wk4.x, qw); qw = fma_(qi[1], wk4.y, qw);
                        qw = fma_(qi[2], wk4.z, qw); qw = fma_(qi[3], wk4.w, qw);
                    }
#pragma unroll
                    for (int kt = 0; kt < RTT; ++kt) {
                        const float4 rv = *reinterpret_cast<const float4*>(remq + 16 * kt);
                        s[kt][0] = fma_(rv.x, qw, s[kt][0]); s[kt][1] = fma_(rv.y, qw, s[kt][1]);
                        s[kt][2] = fma_(rv.z, qw, s[kt][2]); s[kt][3] = fma_(rv.w, qw, s[kt][3]);
                    }
                }
                float m = -INFINITY;
                {
                    // infeasible keys: the score is OR-ed with all ones (a quiet NaN) where the key's mask bit is clear -- v_bfe_i32 of
                    // the inverted, pre-shifted mask word + v_or_b32, two instructions per key instead of three (bit test, compare,
                    // select).  A NaN score behaves like the -inf it replaces everywhere it goes: v_max_f32 / v_max3_f32 return the
                    // other operand(s), and d_expf4_nonpos turns NaN - m into the exact 0 (its final select is an ordered compare).
                    const uint32_t iw[4] = {~mb.x >> G, ~mb.y >> G, ~mb.z >> G, ~mb.w >> G};
#pragma unroll
                    for (int kt = 0; kt < RTT; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int n0 = 16 * kt + 4 * r;          // key n0 + G: bit (n0 & 31) of the shifted word (n0 & 31 <= 28, G <= 3)
                            const int bad = __builtin_amdgcn_sbfe(iw[n0 >> 5], n0 & 31, 1);      // 0 or -1
                            s[kt][r] = __uint_as_float(__float_as_uint(s[kt][r]) | (uint32_t)bad);
                        }
#pragma unroll
                    for (int kt = 0; kt < RTT; ++kt) m = vmax5_raw(m, s[kt][0], s[kt][1], s[kt][2], s[kt][3]);
                }
                m = group_max(m);
                MSTAMP(2);
                // softmax weights; a masked node has s = -inf and d_expf2_nonpos gives it exactly 0 (as the canonical select does)
#pragma unroll
                for (int kt = 0; kt < RTT; ++kt) {
                    // (compile-time chunk length: keys from 4 CC on do not exist -- their weight is the 0 a masked key gets)
                    const bool pad01 = CC > 0 && 16 * kt >= 4 * CC, pad23 = CC > 0 && 16 * kt + 8 >= 4 * CC;
                    if (pad01) {
                        s[kt] = z4();
                    } else if (pad23) {
                        const f32x2 e01 = d_expf2_nonpos((f32x2){s[kt][0] - m, s[kt][1] - m});
                        s[kt] = (f32x4){e01.x, e01.y, 0.0f, 0.0f};
                    } else {
                        s[kt] = d_expf4_nonpos(s[kt] - splat4(m));     // both pairs step by step (dmath.hpp)
                    }
                }
                MSTAMP(3);
                // SDVRP: R_h = the lane tree (canonical order: adjacent pairs over the NODE index, block 0 + block 1) of w[n] rem[n].  Node
                // n = 16 kt + 4 r + G: levels 1, 2 are the lane groups G (one row swap each, per register), 4 and 8 the registers r,
                // 16 and 32 the key tiles kt & 3, the two 64-node blocks kt >> 2
                float Rh = 0.0f;
                if (SD) {
                    float tk[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kt = 0; kt < RTT; ++kt) {
                        const float4 rv = *reinterpret_cast<const float4*>(remq + 16 * kt);
                        float pr[4] = {s[kt][0] * rv.x, s[kt][1] * rv.y, s[kt][2] * rv.z, s[kt][3] * rv.w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            auto x = __builtin_amdgcn_permlane16_swap(__float_as_uint(pr[r]), __float_as_uint(pr[r]), false, false);
                            pr[r] = __uint_as_float(x[0]) + __uint_as_float(x[1]);
                            auto y = __builtin_amdgcn_permlane32_swap(__float_as_uint(pr[r]), __float_as_uint(pr[r]), false, false);
                            pr[r] = __uint_as_float(y[0]) + __uint_as_float(y[1]);
                        }
                        tk[kt] = (pr[0] + pr[1]) + (pr[2] + pr[3]);
                    }
                    Rh = ((tk[0] + tk[1]) + (tk[2] + tk[3])) + ((tk[4] + tk[5]) + (tk[6] + tk[7]));
                }
                // Value product per node chunk g = [g C, (g+1) C), chunks in ascending order: `cur` accumulates the chunk in
                // progress (A_g against V^T, Z_g against a row of ones), `tot` the finished ones as ((A0 + A1) + A2) + A3.
                // K-step t holds nodes 4 t .. 4 t + 3 (this lane: 4 t + G); where it straddles a chunk boundary it is issued
                // once per chunk with the other chunk's weights zeroed.
                f32x4 tot_o = z4();
                float tot_z = 0.0f;
                if (CC > 0) {
                    // compile-time chunk length: chunk g is the nodes [g CC, (g + 1) CC) -- slots beyond M carry zero weights and
                    // zero values --, so which k-steps touch it, and which of them straddle a boundary, is known per (g, t4) and
                    // everything but the accumulations folds away (round 3: with the chunk in progress as a loop-carried variable
                    // the unrolled code kept ~70 branches and a group sum per possible boundary position).
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nlo = g * CC, nhi = (g + 1) * CC;
                        f32x4 co = z4();
                        float cz = 0.0f;        // Z_g: this lane's nodes are one residue class mod 4 -> P_gG, then the lane groups
#pragma unroll
                        for (int t4 = 0; t4 < 4 * RTT; ++t4) {
                            if (4 * t4 + 3 >= nlo && 4 * t4 < nhi) {                        // the k-step holds nodes of the chunk
                                const float w = s[t4 >> 2][t4 & 3];
                                const bool whole = 4 * t4 >= nlo && 4 * t4 + 3 < nhi;       // ... and only such nodes
                                const int n = 4 * t4 + G;
                                const float wg = whole ? w : ((n >= nlo && n < nhi) ? w : 0.0f);
                                co = mf(vtf[t4], wg, co);
                                cz = cz + wg;
                            }
                        }
                        const float zg = (nlo & 1) ? group_sum_odd(cz, lane) : group_sum(cz);
                        tot_o = g == 0 ? co : tot_o + co;
                        tot_z = g == 0 ? zg : tot_z + zg;
                    }
                } else {
                f32x4 cur_o = z4();
                float cur_z = 0.0f;        // Z_g: this lane's nodes are one residue class mod 4 -> P_gG, then the lane groups
                int curc = 0;
                bool first = true;
#pragma unroll
                for (int t4 = 0; t4 < 4 * RTT; ++t4) {
                    if (4 * t4 < M) {
                        const float w = s[t4 >> 2][t4 & 3];
                        const int cme = (int)((cgbits >> (2 * t4)) & 3ull);            // chunk of this lane's node 4 t4 + G
                        // CC > 0: node slots beyond M carry zero weights and zero values, so the k-step's last slot decides
                        const int nlast = (CC > 0 || 4 * t4 + 3 < M) ? 4 * t4 + 3 : M - 1;
                        const int cA = (4 * t4 >= b1) + (4 * t4 >= b2) + (4 * t4 >= b3);
                        const int cB = (nlast >= b1) + (nlast >= b2) + (nlast >= b3);
#pragma unroll
                        for (int sub = 0; sub < 4; ++sub) {
                            const int c = cA + sub;
                            if (c <= cB) {                                                // (uniform)
                                if (c != curc) {                                          // chunk curc is complete
                                    const float zg = ((curc * C) & 1) ? group_sum_odd(cur_z, lane) : group_sum(cur_z);
                                    tot_o = first ? cur_o : tot_o + cur_o;
                                    tot_z = first ? zg : tot_z + zg;
                                    first = false; curc = c; cur_o = z4(); cur_z = 0.0f;
                                }
                                const float wg = (cme == c) ? w : 0.0f;
                                cur_o = mf(vtf[t4], wg, cur_o);
                                cur_z = cur_z + wg;
                            }
                        }
                    }
                }
                {
                    const float zg = ((curc * C) & 1) ? group_sum_odd(cur_z, lane) : group_sum(cur_z);
                    tot_o = first ? cur_o : tot_o + cur_o;
                    tot_z = first ? zg : tot_z + zg;
                }
                }
                // lane (query j, G), register r -> head column e = 4 G + r -> A layout (g = r, t = 4 wv + G)
                float* hp = HT + j * TS + 4 * wv + G;
                float hv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (SD) tot_o[r] = fma_(Rh, s_dyn[ME + 16 * wv + 4 * G + r], tot_o[r]);      // heads_h += R_h wv_h
                    hv[r] = tot_o[r] / tot_z;
                    hp[r * TG] = hv[r];
                }
                if (a.heads_out && sq < S) {     // training: keep the step's glimpse output for the backward (zeros for a done row)
                    const int64_t row = mul32w(sq, a.B) + b;
                    *reinterpret_cast<float4*>(a.heads_out + (mul32w(row, a.t_max) + t) * 128 + 16 * wv + 4 * G) =
                        live ? make_float4(hv[0], hv[1], hv[2], hv[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            MSTAMP(4);
            __syncthreads();
            MSTAMP(5);
            QRaw q_raw;
            if (pre) q_raw = q_fetch(qt + 1 == qt1 ? qt0 : qt + 1);       // loads in flight across the logit phase
            // ---- logits of key tile wv: u[query][key] = ((c0 + c1) + c2) + c3, transposed into UT ---------------------------------
            if (wv < RTT) {
                const float* hp = HT + j * TS + G * TG;
                f32x4 c[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    c[g] = z4();
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const float2 lo = *reinterpret_cast<const float2*>(hp + 8 * g + 4 * u);
                        const float2 hi = *reinterpret_cast<const float2*>(hp + 8 * g + 4 * u + 2);
                        c[g] = mf(lpf[64 * (8 * g + 4 * u + 0)], lo.x, c[g]);
                        c[g] = mf(lpf[64 * (8 * g + 4 * u + 1)], lo.y, c[g]);
                        c[g] = mf(lpf[64 * (8 * g + 4 * u + 2)], hi.x, c[g]);
                        c[g] = mf(lpf[64 * (8 * g + 4 * u + 3)], hi.y, c[g]);
                    }
                }
                if (SD) {
                    // c_g[n] = fma(rem[n], hl_g, c_g[n]) with hl_g = heads . lw over the column chunk g (32-term chain in column order):
                    // lane group G computes the chain of chunk G, three row swaps hand every lane all four
                    float hl = 0.0f;
                    const float* hq = HT + j * TS + 8 * G;                  // heads[j][32 G + 4 i + g2] at g2 * TG + i
                    float4 hg[4][2];
#pragma unroll
                    for (int g2 = 0; g2 < 4; ++g2) {
                        hg[g2][0] = *reinterpret_cast<const float4*>(hq + g2 * TG);
                        hg[g2][1] = *reinterpret_cast<const float4*>(hq + g2 * TG + 4);
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float4 lw4 = *reinterpret_cast<const float4*>(s_dyn + 2 * ME + 32 * G + 4 * i);
                        float hh[4];
#pragma unroll
                        for (int g2 = 0; g2 < 4; ++g2) {
                            const float4 v = hg[g2][i >> 2];
                            hh[g2] = (i & 3) == 0 ? v.x : (i & 3) == 1 ? v.y : (i & 3) == 2 ? v.z : v.w;
                        }
                        hl = fma_(hh[0], lw4.x, hl); hl = fma_(hh[1], lw4.y, hl);
                        hl = fma_(hh[2], lw4.z, hl); hl = fma_(hh[3], lw4.w, hl);
                    }
                    // rows of 16 lanes = lane groups: the 16-lane swap leaves (even group's, odd group's) value in every lane, the 32-lane
                    // swap of each (groups 0-1's, groups 2-3's)
                    const auto eo = __builtin_amdgcn_permlane16_swap(__float_as_uint(hl), __float_as_uint(hl), false, false);
                    const auto e2 = __builtin_amdgcn_permlane32_swap(eo[0], eo[0], false, false);
                    const auto o2 = __builtin_amdgcn_permlane32_swap(eo[1], eo[1], false, false);
                    const float hlg[4] = {__uint_as_float(e2[0]), __uint_as_float(o2[0]), __uint_as_float(e2[1]), __uint_as_float(o2[1])};
                    // this lane's keys 16 wv + 4 G + r: node n -> SREM slot (n >> 4) * 16 + (n & 3) * 4 + ((n & 15) >> 2) = 16 wv + 4 r + G
                    const float* rq = SREM + (sq < SMAX ? sq : 0) * (16 * RTT) + 16 * wv + G;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float rn = rq[4 * r];
#pragma unroll
                        for (int g = 0; g < 4; ++g) c[g][r] = fma_(rn, hlg[g], c[g][r]);
                    }
                }
                const f32x4 u = ((c[0] + c[1]) + c[2]) + c[3];          // lane (query j, G): keys 16 wv + 4 G + (0..3)
                *reinterpret_cast<float4*>(UT + j * US + 16 * wv + 4 * G) = make_float4(u[0], u[1], u[2], u[3]);
            }
            MSTAMP(6);
            __syncthreads();
            MSTAMP(9);
            // ---- finish: half-wavefront hw2 = query 2 wv + (lane >> 5) of the tile; lane l32 holds the keys 4 l32 .. 4 l32 + 3 --------
            {
                const int l32 = lane & 31;
                const int jq2 = 2 * wv + (lane >> 5), s2 = 16 * qt + jq2;
                const bool live2 = s2 < S && !s_done[s2];
                const int nb4 = 4 * l32;                                 // this lane's keys: nb4 + r
                const bool inq = l32 < 4 * RTT;                          // UT holds 16 RTT keys per query
                const int64_t r2 = mul32w(s2, a.B) + b;
                uint4 mb2 = make_uint4(0, 0, 0, 0);
                if (live2) mb2 = *reinterpret_cast<const uint4*>(&s_bits[s2][0]);
                float4 u4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (inq) u4 = *reinterpret_cast<const float4*>(UT + jq2 * US + nb4);
                float nz[4] = {1.0f, 1.0f, 1.0f, 1.0f};
                if (a.mode == EAMRL_SAMPLE && live2 && inq) {
                    if (a.use_rng) {
                        // the seed goes through an opaque register pair: otherwise the ten round keys of the generator are computed
                        // once, kept in 20 SGPRs across the whole kernel and spilled (a v_readlane per key, per tile)
                        uint64_t sd = seed;
                        asm volatile("" : "+s"(sd));
                        exp1_noise4(sd, r2, t, l32, nz);
                    } else {
                        const float* np_ = a.noise + (mul32w(r2, a.t_max) + t) * (int64_t)M;
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr)
                            if (nb4 + rr < M) nz[rr] = np_[nb4 + rr];
                    }
                }
                MSTAMP(7);
                f32x4 v;
                {
                    const f32x4 l = (f32x4){u4.x, u4.y, u4.z, u4.w} * splat4(inv_sqrtE);
                    uint32_t w = (nb4 >> 5) == 0 ? mb2.x : (nb4 >> 5) == 1 ? mb2.y : (nb4 >> 5) == 2 ? mb2.z : mb2.w;
                    w = inq ? ~w >> (nb4 & 31) : ~0u;              // bit r set: key nb4 + r is infeasible (or does not exist)
                    v = (a.clip > 0.0f) ? d_tanhf4(l) * splat4(a.clip) : l;       // the four keys step by step (dmath.hpp)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = ((w >> r) & 1u) ? -INFINITY : v[r];     // (no lane masks kept: see vsum below)
                    if (a.temp != 1.0f) {
                        asm volatile("" ::: "memory");
                        v = v / splat4(a.temp);
                    }
                }
                // a NaN among the FEASIBLE logits (tanh, the clip and the temperature pass it on; infeasible keys are -inf by now): the sum
                // of the lane's four values is a NaN exactly then (-inf + x = -inf for every other x the clipped logits can take)
                const float vsum = (v[0] + v[1]) + (v[2] + v[3]);
                const bool nan_seen = vsum != vsum;
                const float mq = half_max(vmax5_raw(v[0], v[1], v[2], v[3], v[3]));
                MSTAMP(8);
                const f32x4 d = v - splat4(mq);
                float lse, lp[4];
                {
                    // an infeasible key has v = -inf, so d = -inf, its weight is the exact 0 and its log-prob -inf without further selects
                    const f32x4 e = d_expf4_nonpos(d);
                    const float Zl = half_tree_sum((e[0] + e[1]) + (e[2] + e[3]));     // levels 1, 2 in lane; 4 .. 32 in the row; block 0 + block 1
                    lse = d_logf(Zl);
#pragma unroll
                    for (int r = 0; r < 4; ++r) lp[r] = d[r] - lse;
                }
                MSTAMP(10);
                // ---- selection: greedy = first node whose log-prob equals the maximum (mx - mx) - lse; sampling = first node with
                //      the largest p / noise --------------------------------------------------------------------------------------
                int cand = 1 << 20;
                if (a.mode == EAMRL_SAMPLE) {
                    const f32x4 kq = d_expf4_nonpos((f32x4){lp[0], lp[1], lp[2], lp[3]}) / (f32x4){nz[0], nz[1], nz[2], nz[3]};
                    float key[4];
                    // (a key that is infeasible or does not exist has p = exp(-inf) = 0, so its ratio is 0 and below the row's best)
#pragma unroll
                    for (int r = 0; r < 4; ++r) key[r] = kq[r];
                    const float top = half_max(vmax5_raw(key[0], key[1], key[2], key[3], key[3]));
#pragma unroll
                    for (int r = 3; r >= 0; --r)
                        if (key[r] == top) cand = nb4 + r;
                } else {
                    const float top = 0.0f - lse;
#pragma unroll
                    for (int r = 3; r >= 0; --r)
                        if (lp[r] == top) cand = nb4 + r;           // (-inf for a key that is infeasible or does not exist)
                }
                int sel = half_min(cand);
                if (a.mode == EAMRL_EVALUATE && live2) sel = (t < a.t_given) ? (int)a.given[mul32w(r2, a.t_given) + t] : 0;
                uint32_t fl = 0;
                if (live2) {
                    if (nan_seen) fl |= EAMRL_ST_NAN_LOGITS;
                    if (sel < 0 || sel >= M) { fl |= EAMRL_ST_INFEASIBLE; sel = 0; }
                }
                if (fl) atomicOr(&s_flags, fl);
                if (live2 && inq && sel >= nb4 && sel < nb4 + 4) LPSEL[jq2] = lp[sel - nb4];      // the lane that holds the pick
                __builtin_amdgcn_wave_barrier();        // LPSEL is read by lanes of the same wavefront: LDS keeps its accesses in order
                MSTAMP(12);
                // ---- env transition (TSPEnv._step, tsp/env.py:62-88): lane 0 of the half-wavefront ---------------------------------
                if (!DEP && l32 == 0 && live2) {
                    const int s = s2, sl = sel;
                    a.action[mul32w(r2, a.t_max) + t] = sl;
                    a.logp[mul32w(r2, a.t_max) + t] = LPSEL[jq2];
                    const uint32_t bit = 1u << (sl & 31);
                    const uint32_t wd = s_bits[s][sl >> 5];
                    if (!(wd & bit)) atomicOr(&s_flags, EAMRL_ST_INFEASIBLE);
                    s_bits[s][sl >> 5] = wd & ~bit;
                    if (s_istep[s] == 0) s_first[s] = sl;
                    s_cur[s] = sl;
                    s_istep[s] += 1;
                    s_cnt[s] -= (wd & bit) != 0;
                    s_done[s] = s_cnt[s] == 0;
                }
                // ---- depot envs: the env's _step + get_action_mask (cvrp/env.py:68-144, cvrptw/env.py:103-138, pctsp/env.py:64-97,156-163,
                // op/env.py:69-102,149-165; the expressions of k_rollout_resident's finish): every lane of the half-wavefront knows the
                // pick, lane l tests nodes l, l + 32, l + 64, l + 96, the ballots are the new mask words ----------------------------------
                if (DEP && live2) {
                    const int s = s2, sl = sel, hw = lane >> 5;
                    const float lpv = LPSEL[jq2];
                    const uint4 ob = mb2;
                    uint4 vw = *reinterpret_cast<const uint4*>(&s_vis[s][0]);
                    const uint32_t bit = 1u << (sl & 31);
                    const int wi = sl >> 5;
                    const uint32_t oword = wi == 0 ? ob.x : wi == 1 ? ob.y : wi == 2 ? ob.z : ob.w;
                    const uint32_t vword = wi == 0 ? vw.x : wi == 1 ? vw.y : wi == 2 ? vw.z : vw.w;
                    const bool was_vis = (vword & bit) != 0;
                    if (wi == 0) vw.x |= bit; else if (wi == 1) vw.y |= bit; else if (wi == 2) vw.z |= bit; else vw.w |= bit;
                    const int curn = s_cur[s];
                    float u = s_used[s], now = 0.0f, cx = 0.0f, cy = 0.0f;
                    int cnt = s_cnt[s], ist = 0;
                    bool done_new;
                    if (XY) { cx = s_xy[2 * sl]; cy = s_xy[2 * sl + 1]; }
                    float sd_left = 0.0f;
                    if (SD) {           // SDVRPEnv._step (sdvrp/env.py:58-92): deliver min(remaining demand, free capacity)
                        const float selrem = SREM[s * (16 * RTT) + rperm(sl)];
                        const float free_cap = s_cap[s] - u;
                        const float delivered = selrem < free_cap ? selrem : free_cap;
                        u = (u + delivered) * (sl != 0 ? 1.0f : 0.0f);
                        sd_left = selrem + (-delivered);
                        done_new = false;       // (set from the remaining demands below)
                    } else if (OP) {
                        ist = s_istep[s];
                        const float dx = cx - s_xy[2 * curn], dy = cy - s_xy[2 * curn + 1];
                        u = u + __builtin_sqrtf(fma_(dy, dy, dx * dx));
                        done_new = (sl == 0) && (ist > 0);
                    } else if (PC) {
                        ist = s_istep[s];
                        u = u + s_dem[sl];
                        done_new = (ist > 0) && (sl == 0);
                        cnt += (sl != 0 && !was_vis);
                    } else {
                        if (TW) {       // clock (cvrptw/env.py:118-138)
                            now = s_time[s];
                            const float dx = s_xy[2 * curn] - cx, dy = s_xy[2 * curn + 1] - cy;
                            const float arrive = now + __builtin_sqrtf(fma_(dy, dy, dx * dx));
                            const float ws = s_tw0[sl];
                            const float start = arrive > ws ? arrive : ws;
                            now = (sl != 0 ? 1.0f : 0.0f) * (start + s_dur[sl]);
                        }
                        int di = sl - 1;
                        di = di < 0 ? 0 : (di > M - 2 ? M - 2 : di);
                        u = (u + s_dem[di + 1]) * (sl != 0 ? 1.0f : 0.0f);
                        cnt += was_vis ? 0 : 1;
                        done_new = cnt == M;
                    }
                    const bool v0 = (vw.x & 1u) != 0;               // the depot has been visited (after this step)
                    const float lim = s_cap[s] + 1e-5f;
                    uint32_t nb[4], fr = 0, anyrem = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int n = 32 * k + l32;
                        const uint32_t vk = k == 0 ? vw.x : k == 1 ? vw.y : k == 2 ? vw.z : vw.w;
                        const bool vis_n = (vk >> l32) & 1u;
                        const bool cust = n >= 1 && n < M;
                        bool ok, freeb = false;
                        if (SD) {       // get_action_mask (sdvrp/env.py:137-146): nothing left to deliver there, or the vehicle is full
                            float rv = (n < 16 * RTT) ? SREM[s * (16 * RTT) + rperm(n)] : 0.0f;
                            rv = n == sl ? sd_left : rv;
                            freeb = cust && !((rv == 0.0f) | (u >= s_cap[s]));
                            ok = freeb;
                            anyrem |= (uint32_t)(__ballot(n < M && rv > 0.0f) >> (32 * hw));
                        } else if (PC) {
                            ok = cust && !(vis_n | v0);
                        } else if (OP) {
                            const float dx = s_xy[2 * n] - cx, dy = s_xy[2 * n + 1] - cy;
                            const bool exceeds = (u + __builtin_sqrtf(fma_(dy, dy, dx * dx))) > s_dem[n];
                            ok = cust && !(vis_n | v0 | exceeds);
                        } else {
                            freeb = cust && !(vis_n | ((s_dem[n] + u) > lim));
                            ok = freeb;
                            if (TW) {
                                const float dx = cx - s_xy[2 * n], dy = cy - s_xy[2 * n + 1];
                                ok = ok && (now + __builtin_sqrtf(fma_(dy, dy, dx * dx))) <= s_tw1[n];
                            }
                        }
                        nb[k] = (uint32_t)(__ballot(ok) >> (32 * hw));
                        if (CV || SD) fr |= (uint32_t)(__ballot(freeb) >> (32 * hw));
                    }
                    if (SD) done_new = anyrem == 0u;            // done = no demand left anywhere (sdvrp/env.py:84-86)
                    // the depot's bit
                    if (PC) {
                        if (!((u < 1.0f) && (cnt < M - 1))) nb[0] |= 1u;     // opens once the prize is collected (or everyone visited)
                    } else if (OP) {
                        nb[0] |= 1u;                                          // always feasible (and ends the episode)
                    } else {
                        bool ok0 = !((sl == 0) && fr != 0u);                  // closed only while at it with customers left
                        if (TW) {
                            const float dx = cx - s_xy[0], dy = cy - s_xy[1];
                            ok0 = ok0 && (now + __builtin_sqrtf(fma_(dy, dy, dx * dx))) <= s_tw1[0];
                        }
                        if (ok0) nb[0] |= 1u;
                    }
                    if (l32 == 0) {
                        a.action[mul32w(r2, a.t_max) + t] = sl;
                        a.logp[mul32w(r2, a.t_max) + t] = lpv;
                        if (!(oword & bit)) atomicOr(&s_flags, EAMRL_ST_INFEASIBLE);
                        *reinterpret_cast<uint4*>(&s_bits[s][0]) = make_uint4(nb[0], nb[1], nb[2], nb[3]);
                        if (!SD) *reinterpret_cast<uint4*>(&s_vis[s][0]) = vw;
                        if (SD) SREM[s * (16 * RTT) + rperm(sl)] = sd_left;     // (every lane's reads of the row are above: one wavefront, LDS in order)
                        s_used[s] = u;
                        s_cur[s] = sl;
                        s_cnt[s] = cnt;
                        s_done[s] = done_new;
                        if (TW) s_time[s] = now;
                        if (PC || OP) s_istep[s] = ist + 1;
                    }
                }
            }
            if (pre) q_store(q_combine(q_raw)); // QT is free since the glimpse barrier; the next tile's first barrier publishes it
            have_q = pre;
            // (UT / LPSEL are rewritten by the next tile only behind its glimpse barrier, which every wavefront reaches after this
            //  finish phase; the state of these 16 starts is next read one round later, or behind the tile-start barrier)
            MSTAMP(14);
        }
        ++t;
    }
#ifdef EAMRL_STAMPS
    if (lane == 0) {
        for (int i = 0; i < 16; ++i) atomicAdd(&g_ms_stamps[i], st_acc[i]);
        atomicAdd(&g_ms_stamps[16], 1ull);
    }
#endif
    // ---- final state -------------------------------------------------------------------------------------------------------------
    __syncthreads();
    if (a.heads_out) {          // the steps this instance did not take: zeros (the caller's buffer is not initialised)
        const int per = (a.t_max - t) * 32;              // float4 per row
        for (int s = s_lo; s < s_hi; ++s) {
            float4* hz = reinterpret_cast<float4*>(a.heads_out + (((int64_t)s * a.B + b) * a.t_max + t) * 128);
            for (int i = tid; i < per; i += blockDim.x) hz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    for (int s = s_lo + tid; s < s_hi; s += blockDim.x) {
        const int64_t r = (int64_t)s * a.B + b;
        for (int n = 0; n < M; ++n) a.mask[r * M + n] = (s_bits[s][n >> 5] >> (n & 31)) & 1u;
        a.cur[r] = s_cur[s];
        if (SD) {
            for (int n = 0; n < M; ++n) a.rem[r * M + n] = SREM[s * (16 * RTT) + rperm(n)];
            a.used[r] = s_used[s];
        } else if (DEP) {
            for (int n = 0; n < M; ++n) a.visited[r * M + n] = (s_vis[s][n >> 5] >> (n & 31)) & 1u;
            a.used[r] = s_used[s];
            if (TW) a.time[r] = s_time[s];
            if (PC || OP) a.istep[r] = s_i0[s];         // launch_rollout_pad adds the batch's step count
        } else {
            a.first[r] = s_first[s];
            a.istep[r] = s_istep[s];
        }
        a.done[r] = s_done[s] ? 1 : 0;
        if (!s_done[s]) atomicOr(&s_flags, EAMRL_ST_STEP_OVERRUN);
    }
    __syncthreads();
    if (tid == 0) {
        atomicMax(a.steps_out, t);
        if (s_flags) atomicOr(a.status, s_flags);
    }
}

template <int RTT, int CC, int ENV>
int launch_t(const DecArgs& a, int S, hipStream_t st)
{
    const size_t lds = ((size_t)RTT * 32 * 64 + (ENV == EAMRL_ENV_SDVRP ? (size_t)SMAX * 16 * RTT : 0)) * sizeof(float);
    auto k = k_rollout_ms_mfma<RTT, CC, ENV>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(lds + (ENV == EAMRL_ENV_SDVRP ? 0 : 32 * 1024))) != hipSuccess)
        return EAMRL_E_LAUNCH;
    // one workgroup per CU is resident (register budget): a batch smaller than the chip splits each instance's query tiles
    const int nqt = (S + 15) / 16;
    int nsplit = a.B > 0 ? (int)(256 / a.B) : 1;
    nsplit = nsplit < 1 ? 1 : (nsplit > nqt ? nqt : nsplit);
    if (g_debug[13]) nsplit = 1;
    hipLaunchKernelGGL(k, dim3((unsigned)(a.B * nsplit)), dim3(512), lds, st, a, S, nsplit);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

template <int ENV>
int launch_env_t(const DecArgs& a, int S, int C, hipStream_t st)
{
    // compile-time chunk lengths for the common sizes: TSP-20 / 50 / 100 (C = 5, 13, 25) and the depot envs' 21 / 51 / 101 nodes
    // (C = 6, 13, 26); everything else runs the run-time variant
    constexpr bool T = ENV == EAMRL_ENV_TSP || ENV == EAMRL_ENV_CVRP;       // (the sibling envs: fewer instantiations)
    if (a.M <= 32) {
        if (T && C == 5) return launch_t<2, T ? 5 : 6, ENV>(a, S, st);
        return C == 6 ? launch_t<2, 6, ENV>(a, S, st) : launch_t<2, 0, ENV>(a, S, st);
    }
    if (a.M <= 64) return C == 13 ? launch_t<4, 13, ENV>(a, S, st) : launch_t<4, 0, ENV>(a, S, st);
    if (T && C == 25) return launch_t<7, T ? 25 : 26, ENV>(a, S, st);
    return C == 26 ? launch_t<7, 26, ENV>(a, S, st) : launch_t<7, 0, ENV>(a, S, st);
}

}  // namespace

#ifdef EAMRL_STAMPS
extern "C" __attribute__((visibility("default"))) int eamrl_debug_read_ms_stamps(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ms_stamps), sizeof(g_ms_stamps)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[24] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_ms_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// shape_only: the question eamrl_rollout_rng_native asks before the state exists (cache shape and row count alone)
bool rollout_ms_mfma_supports(int env, const DecArgs& a, bool shape_only)
{
    const bool sd = env == EAMRL_ENV_SDVRP;
    const bool depot_env = env == EAMRL_ENV_CVRP || env == EAMRL_ENV_CVRPTW || env == EAMRL_ENV_PCTSP || env == EAMRL_ENV_OP;
    if (sd) {
        if (g_debug[14] || a.E != ME || a.H != MH || a.M < 2 || a.M > 112 || a.ld % 4 != 0 || a.ld >= (1 << 24)) return false;
        if (!shape_only && (!a.dyn || !a.rem || !a.used || !a.vcap)) return false;
        if (a.R % a.B != 0 || a.R >= (1ll << 31)) return false;
        const int64_t S = a.R / a.B;
        return S >= 2 && S <= SMAX && a.top_k == 0 && !(a.top_p > 0.0f && a.top_p < 1.0f);
    }
    if ((env != EAMRL_ENV_TSP && !depot_env) || a.E != ME || a.H != MH || a.M < 2 || a.M > 112 || a.ld % 4 != 0 || a.ld >= (1 << 24)) return false;
    if (depot_env && (g_debug[14] || (!shape_only && (!a.visited || !a.used || !a.vcap || !a.demand)))) return false;
    if (!shape_only && (env == EAMRL_ENV_OP || env == EAMRL_ENV_CVRPTW) && !a.locs) return false;
    if (!shape_only && env == EAMRL_ENV_CVRPTW && (!a.time || !a.tw || !a.dur)) return false;
    if (!shape_only && (env == EAMRL_ENV_PCTSP || env == EAMRL_ENV_OP) && !a.istep) return false;
    if (a.R % a.B != 0 || a.R >= (1ll << 31)) return false;    // mul32w below
    const int64_t S = a.R / a.B;
    return S >= 2 && S <= SMAX && a.top_k == 0 && !(a.top_p > 0.0f && a.top_p < 1.0f);
}

int launch_rollout_ms_mfma(int env, const DecArgs& a, hipStream_t st)
{
    const int S = (int)(a.R / a.B);
    const int C = (a.M + EAMRL_NCHUNK - 1) / EAMRL_NCHUNK;
    if (env == EAMRL_ENV_TSP) return launch_env_t<EAMRL_ENV_TSP>(a, S, C, st);
    const int rc = env == EAMRL_ENV_CVRP ? launch_env_t<EAMRL_ENV_CVRP>(a, S, C, st)
                 : env == EAMRL_ENV_CVRPTW ? launch_env_t<EAMRL_ENV_CVRPTW>(a, S, C, st)
                 : env == EAMRL_ENV_PCTSP ? launch_env_t<EAMRL_ENV_PCTSP>(a, S, C, st)
                 : env == EAMRL_ENV_SDVRP ? launch_env_t<EAMRL_ENV_SDVRP>(a, S, C, st) : launch_env_t<EAMRL_ENV_OP>(a, S, C, st);
    if (rc) return rc;
    launch_rollout_pad(env, a, st);              // rows that finished early end at the depot (and PCTSP / OP get their step counters)
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------------
// the same Exp(1) draws as a tensor (for kernels without in-place noise, and to pin the in-place path)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void k_exp1_noise(uint64_t seed0, const uint64_t* __restrict__ seed_dev, float* __restrict__ noise, int64_t R, int T,
                             int M)
{
    const uint64_t seed = seed0 ^ (seed_dev ? *seed_dev : 0ull);
    const int nq = (M + 3) >> 2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * T * nq) return;
    const int q = (int)(idx % nq);
    const int64_t rt = idx / nq;
    const int t = (int)(rt % T);
    const int64_t r = rt / T;
    float v[4];
    exp1_noise4(seed, r, t, q, v);
    for (int i = 0; i < 4 && 4 * q + i < M; ++i) noise[rt * M + 4 * q + i] = v[i];
}

int launch_exp1_noise(uint64_t seed, const uint64_t* seed_dev, float* noise, int64_t R, int T, int M, hipStream_t st)
{
    const int64_t n = R * T * ((M + 3) >> 2);
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_exp1_noise, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, seed, seed_dev, noise, R, T, M);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

}  // namespace eamrl
