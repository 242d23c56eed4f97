// Teacher-forced re-evaluation of a finished rollout, forward and backward, on fp32 MFMA (the gradient path of training).
//
// Reference: rl4co/models/common/constructive/base.py:203-229 + rl4co/utils/decoding.py:452-465 (`policy(..., actions=)`,
// decode type "evaluate"): for every decode step t of every row r the log-probability of the action actually taken,
//     q      = Pa[b][ia] + Pb[b][ib] + gctx[b] + sum_k sc_k[r,t] C_k              (the context query; weight folds of DESIGN.md 2)
//     heads  = MHA(q; K[b], V[b] | mask[r,t])                                    (8 heads of 16, scores / 4)
//     u[n]   = heads . Lp[b][n] / sqrt(E);  z = clip tanh(u) (or u), masked -> -inf, / temperature
//     logp   = z[a] - logsumexp(z)
// -- rl4co/models/zoo/am/decoder.py:133-198, rl4co/models/nn/attention.py:282-328, rl4co/utils/decoding.py:140-190 -- and its
// gradient with respect to K, V, Lp, Pa, Pb, gctx and the state columns C_k.  Because the actions are known, every step's
// state is a prefix function of the action row: the feasibility masks arrive as bit sets (one 128-bit word per (row, step),
// produced by replaying the env kernels) and all S x T queries of an instance share its K / V / Lp -- dense contractions.
//
// This is NOT the parity-critical rollout path (that is decode_step.hip / rollout_resident.hip, bit-exact to the oracle):
// values here are held to 1e-5 of the native rollout's log-probs and gradients to 1e-4 of autograd (tests/test_gpu_train.py),
// so the kernels use the hardware exp / log and whatever summation order the tiles give.
//
// Structure (all kernels): one workgroup of 8 wavefronts per (instance, chunk of its rows); 16 queries per tile; operands of
// the instance (K, V, Lp slices) live in REGISTERS as MFMA fragments for the whole launch, LDS only stages the tile's query /
// head vectors.  Score tiles are computed transposed (S^T = K Q^T: lane = query, registers = keys) with the keys placed on
// MFMA rows so that the accumulators are directly the B operand of the products that contract over the keys.
#include "kernels.hpp"

namespace eamrl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int RE = 128, RH = 8;      // embed dim, heads (D = 16)
constexpr int TS = 140, TG = 34;     // A-layout tile buffers [16][TS]: element (j, c) at j * TS + (c & 3) * TG + (c >> 2)
constexpr float LOG2E = 1.44269504088896341f;

__device__ __forceinline__ f32x4 mf(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 z4() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * LOG2E); }

// combine over the four lane groups (lanes l, l^16, l^32, l^48) that share a query
__device__ __forceinline__ float group_max(float v)
{
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float group_sum(float v)
{
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// ---- SDVRP: the dynamic embedding (rl4co/models/nn/env_embeddings/dynamic.py:59-78) ---------------------------------------
// Every step adds rem[n] * (wk | wv | wl) to row n of the glimpse key / value / logit key, rem = the remaining demands of the
// row at that step (demand_with_depot).  The update is rank one, so it never touches the instance's K / V / Lp fragments:
//   scores   s[n] += rem[n] (q~ . wk_h)                    heads_h += (sum_n a[n] rem[n]) wv_h
//   logits   u[n] += rem[n] (heads . lw),  lw = wl folded through project_out like Lp
// and in the backward  dheads += (sum_n du[n] rem[n]) lw,  da[n] += rem[n] (wv_h . dO_h),  dq~_h += (sum_n ds[n] rem[n]) wk_h,
//   dlw += sum_q (sum_n du rem) heads_q,  dwv_h += sum_q (sum_n a rem) dO_h,  dwk_h += sum_q (sum_n ds rem) q~_h.
// a.dyn = wk | wv | lw [3][E]; a.rem [R][T][RMP] (zero padded rows, recorded by eamrl_replay_states_sdvrp); a.ddyn accumulates.
// A tile's rem rows are staged in LDS so that lane (query j, G) reads the keys of its accumulator registers, 16 kt + 4 r + G
// (r = 0..3), as one float4: element (q, n) at q * 16 RTT + (n >> 4) * 16 + (n & 3) * 4 + ((n & 15) >> 2).

// wk | wv live in LDS (DYNL[2][E], filled once per workgroup) and are read where they are used: the RTT = 7 backward kernels
// have no registers to hold a lane's 16 entries across a tile (volatile: or the compiler hoists the reads and spills instead).
__device__ __forceinline__ float lds_vread(const float* p)
{
    return *reinterpret_cast<const volatile __attribute__((address_space(3))) float*>((const __attribute__((address_space(3))) float*)p);
}
struct DynLane {
    const float* base;      // DYNL
    int h, G;
    // [16 h + 4 i + G]: the layout of the staged q~ / dO operands;  [16 h + 4 G + r]: that of the dq~ / heads accumulators
    __device__ __forceinline__ float wk_in(int i) const { return lds_vread(base + 16 * h + 4 * i + G); }
    __device__ __forceinline__ float wv_in(int i) const { return lds_vread(base + RE + 16 * h + 4 * i + G); }
    __device__ __forceinline__ float wk_out(int r) const { return lds_vread(base + 16 * h + 4 * G + r); }
    __device__ __forceinline__ float wv_out(int r) const { return lds_vread(base + RE + 16 * h + 4 * G + r); }
};
__device__ __forceinline__ void fill_dyn_lds(const float* dyn, float* DYNL)     // (a barrier before the first use)
{
    for (int i = threadIdx.x; i < 2 * RE; i += blockDim.x) DYNL[i] = dyn[i];
}
// thread (query jq, float4 e4 of the row): rem[q][4 e4 .. 4 e4 + 3] -> the staged tile
template <int RTT>
__device__ __forceinline__ void stage_rem(float* REM, int jq, int e4, const float4& v, bool ok)
{
    if (e4 < 4 * RTT) {            // n = 4 e4 + i: key tile e4 >> 2, register e4 & 3, lane group i
        float* p = REM + jq * (16 * RTT) + (e4 >> 2) * 16 + (e4 & 3);
        p[0] = ok ? v.x : 0.0f; p[4] = ok ? v.y : 0.0f; p[8] = ok ? v.z : 0.0f; p[12] = ok ? v.w : 0.0f;
    }
}
template <int RTT>
__device__ __forceinline__ f32x4 rem_quad(const float* REM, int j, int kt, int G)
{
    const float4 v = *reinterpret_cast<const float4*>(REM + j * (16 * RTT) + kt * 16 + G * 4);
    return (f32x4){v.x, v.y, v.z, v.w};
}
template <int RTT>
__device__ __forceinline__ const float* rem_lane(const float* REM, int j, int G) { return REM + j * (16 * RTT) + G * 4; }
__device__ __forceinline__ f32x4 rem_at(const float* remq, int kt)
{
    // (read through a volatile LDS pointer: otherwise the compiler keeps the value of the first read in a register for the later ones)
    return *reinterpret_cast<const volatile __attribute__((address_space(3))) f32x4*>(
        (const __attribute__((address_space(3))) float*)remq + kt * 16);
}

struct Q {        // the tile's query of this lane (query j = lane & 15)
    int64_t qi;   // r * T + t, or -1
    int64_t r;
    int t;
    bool active;
};

__device__ __forceinline__ Q tile_query(const ReevalArgs& a, int64_t b, int s0, int64_t nq, int64_t tile, int j)
{
    Q q;
    const int64_t l = tile * 16 + j;
    q.active = false; q.qi = -1; q.r = 0; q.t = 0;
    if (l < nq) {
        const int64_t sl = l / a.T;
        q.t = (int)(l - sl * a.T);
        q.r = (s0 + sl) * a.B + b;
        q.qi = q.r * a.T + q.t;
        q.active = q.t >= a.tstart;
    }
    return q;
}

// ---- graphs above 112 nodes: key chunks ---------------------------------------------------------------------------------------
// The kernels below hold ONE chunk of at most 112 keys (K / V / Lp fragments in registers).  A larger graph is cut into
// nkc = ceil(M / 112) chunks and every (instance, row chunk) gets one workgroup per key chunk; `chunk_view` turns the arguments
// into those of a 112-node instance made of the chunk's keys (pointer arithmetic only: the kernels' index expressions
// (b * M + n) * ld stay as they are), the mask words and the action index become chunk-local, and what a softmax needs from the
// other chunks travels through small per-query statistics (k_reeval_mc_* below).
constexpr int KCH = 112;
constexpr int RMP = 128;      // floats per remaining-demand row (SDVRP), per key chunk
struct Blk { int64_t b; int ch, c; };
__device__ __forceinline__ Blk decode_block(const ReevalArgs& a)
{
    Blk k;
    int64_t blk = blockIdx.x;
    k.c = (int)(blk % a.nkc);
    blk /= a.nkc;
    k.b = blk / a.nchunk;
    k.ch = (int)(blk - k.b * a.nchunk);
    return k;
}
__device__ __forceinline__ void chunk_view(ReevalArgs& a, int64_t b, int c)
{
    if (a.nkc <= 1) return;
    const int koff = KCH * c, Mc = min(KCH, a.M - koff);
    const int64_t adj = b * (a.M - Mc);            // the body computes (b * Mc + n) * ld: make that row b * M + koff + n
    a.K += (adj + koff) * a.ld; a.V += (adj + koff) * a.ld; a.Lp += (adj + koff) * a.ld;
    a.Pa += adj * a.ld;                            // context rows are indexed by any node of the instance
    if (a.Pb) a.Pb += adj * a.ld;
    if (a.dK) { a.dK += (adj + koff) * a.ldg; a.dV += (adj + koff) * a.ldg; a.dLp += (adj + koff) * a.ldg; }
    a.maskbits += 4 * c;
    if (a.rem) a.rem += RMP * c;                  // SDVRP: rem [R][T][nkc][RMP], chunk-local rows
    a.mc_koff = koff;
    a.M = Mc;
}

// q~ tile = 0.25 * (Pa[ia] + Pb[ib] + gctx + sum_k sc_k C_k)  ->  QT (A layout); 512 threads: 32 float4 per query
__device__ __forceinline__ void build_query_tile(const ReevalArgs& a, int64_t b, int s0, int64_t nq, int64_t tile, float* QT)
{
    const int jq = threadIdx.x >> 5, e4 = threadIdx.x & 31;
    const Q q = tile_query(a, b, s0, nq, tile, jq);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q.qi >= 0) {
        const int ia = a.idxA[q.qi];
        if (ia >= 0) v = *reinterpret_cast<const float4*>(a.Pa + (b * a.M + ia) * a.ld + 4 * e4);
        if (a.idxB) {
            const int ib = a.idxB[q.qi];
            if (ib >= 0) {
                const float4 w = *reinterpret_cast<const float4*>(a.Pb + (b * a.M + ib) * a.ld + 4 * e4);
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
        }
        if (a.gctx) {
            const float4 w = *reinterpret_cast<const float4*>(a.gctx + b * RE + 4 * e4);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        for (int k = 0; k < a.NC; ++k) {
            const float s = a.sc[(int64_t)k * a.R * a.T + q.qi];
            const float4 w = *reinterpret_cast<const float4*>(a.Cvec + k * RE + 4 * e4);
            v.x = fmaf(s, w.x, v.x); v.y = fmaf(s, w.y, v.y); v.z = fmaf(s, w.z, v.z); v.w = fmaf(s, w.w, v.w);
        }
    }
    float* p = QT + jq * TS + e4;
    p[0] = 0.25f * v.x; p[TG] = 0.25f * v.y; p[2 * TG] = 0.25f * v.z; p[3 * TG] = 0.25f * v.w;
}

// Fragments of head h of instance b, loaded once per workgroup.
//   kf[kt][t']  scores A operand: row rho of key tile kt holds key 16 kt + pi(rho), pi(rho) = 4 (rho & 3) + (rho >> 2),
//               so that accumulator register r of lane group G is key 16 kt + 4 r + G
//   vtf[t]      value A operand (V^T): row e, k index = key 4 t + g
template <int RTT>
__device__ __forceinline__ void load_head_frags(const ReevalArgs& a, int64_t b, int h, int lane, float (&kf)[RTT][4],
                                                float (&vtf)[4 * RTT])
{
    const int j = lane & 15, G = lane >> 4, pi = 4 * (j & 3) + (j >> 2);
#pragma unroll
    for (int kt = 0; kt < RTT; ++kt) {
        const int n = 16 * kt + pi;
#pragma unroll
        for (int t = 0; t < 4; ++t) kf[kt][t] = n < a.M ? a.K[(b * a.M + n) * a.ld + 16 * h + 4 * t + G] : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < 4 * RTT; ++t) {
        const int n = 4 * t + G;
        vtf[t] = n < a.M ? a.V[(b * a.M + n) * a.ld + 16 * h + j] : 0.0f;
    }
}

// softmax weights of head h for the tile's queries: s[kt][r] <- w (unnormalised), returns 1 / Z (0 when nothing is feasible)
// DYN: remq = this lane's float4 column of the staged remaining demands (rem_lane), re-read at every use instead of held in 28
// registers (the RTT = 7 backward kernels have none to spare)
template <int RTT, bool DYN = false>
__device__ __forceinline__ float head_softmax(const float (&kf)[RTT][4], const float* QT, int h, int lane, const uint4& mb, int M,
                                              f32x4 (&s)[RTT], const float* remq = nullptr, const DynLane* dl = nullptr,
                                              float* stats = nullptr, const float* given = nullptr)
{
    // stats (key chunks, forward): -> (maximum or -inf, sum of the weights) of THIS chunk; given (backward): (maximum, 1 / sum) over
    // ALL chunks -- the weights are then exp(s - given maximum) and nothing is reduced here
    const int j = lane & 15, G = lane >> 4;
    const float* qp = QT + j * TS + G * TG + 4 * h;
    const float2 qlo = *reinterpret_cast<const float2*>(qp), qhi = *reinterpret_cast<const float2*>(qp + 2);
#pragma unroll
    for (int kt = 0; kt < RTT; ++kt) s[kt] = mf(kf[kt][0], qlo.x, z4());
#pragma unroll
    for (int kt = 0; kt < RTT; ++kt) s[kt] = mf(kf[kt][1], qlo.y, s[kt]);
#pragma unroll
    for (int kt = 0; kt < RTT; ++kt) s[kt] = mf(kf[kt][2], qhi.x, s[kt]);
#pragma unroll
    for (int kt = 0; kt < RTT; ++kt) s[kt] = mf(kf[kt][3], qhi.y, s[kt]);
    if (DYN) {
        const float c = group_sum(fmaf(qhi.y, dl->wk_in(3), fmaf(qhi.x, dl->wk_in(2), fmaf(qlo.y, dl->wk_in(1), qlo.x * dl->wk_in(0)))));   // q~_h . wk_h
#pragma unroll
        for (int kt = 0; kt < RTT; ++kt) {
            const f32x4 rv = rem_at(remq, kt);
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][r] = fmaf(rv[r], c, s[kt][r]);
        }
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < RTT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n0 = 16 * kt + 4 * r;                      // + G at run time: same 32-bit word
            const uint32_t w = (n0 >> 5) == 0 ? mb.x : (n0 >> 5) == 1 ? mb.y : (n0 >> 5) == 2 ? mb.z : mb.w;
            const bool ok = (w >> ((n0 & 31) + G)) & 1u;
            s[kt][r] = ok ? s[kt][r] : -INFINITY;
            m = fmaxf(m, s[kt][r]);
        }
    if (given) {
        m = given[0];
#pragma unroll
        for (int kt = 0; kt < RTT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][r] = fexp(s[kt][r] - m);
        return given[1];
    }
    m = group_max(m);
    if (stats) stats[0] = m;
    if (m == -INFINITY) m = 0.0f;
    float z = 0.0f;
#pragma unroll
    for (int kt = 0; kt < RTT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[kt][r] = fexp(s[kt][r] - m);
            z += s[kt][r];
        }
    z = group_sum(z);
    if (stats) stats[1] = z;
    return z > 0.0f ? 1.0f / z : 0.0f;
}

// heads tile of all heads -> HT (A layout): wave h computes head h
template <int RTT, bool DYN = false>
__device__ __forceinline__ void glimpse_tile(const float (&kf)[RTT][4], const float (&vtf)[4 * RTT], const float* QT, float* HT,
                                             int h, int lane, const uint4& mb, int M, const float* REM = nullptr,
                                             const DynLane* dl = nullptr)
{
    const int j = lane & 15, G = lane >> 4;
    f32x4 s[RTT];
    const float* remq = DYN ? rem_lane<RTT>(REM, j, G) : nullptr;
    const float iz = head_softmax<RTT, DYN>(kf, QT, h, lane, mb, M, s, remq, dl);
    f32x4 o = z4();
#pragma unroll
    for (int t = 0; t < 4 * RTT; ++t) o = mf(vtf[t], s[t >> 2][t & 3], o);
    if (DYN) {
        float ra = 0.0f;
#pragma unroll
        for (int kt = 0; kt < RTT; ++kt) {
            const f32x4 rv = rem_at(remq, kt);
#pragma unroll
            for (int r = 0; r < 4; ++r) ra = fmaf(s[kt][r], rv[r], ra);
        }
        ra = group_sum(ra);                        // sum_n w[n] rem[n] (unnormalised, like o)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = fmaf(ra, dl->wv_out(r), o[r]);
    }
    float* hp = HT + j * TS + 4 * h + G;           // element (j, c = 16 h + 4 G + r) -> (g = r, t = 4 h + G)
#pragma unroll
    for (int r = 0; r < 4; ++r) hp[r * TG] = o[r] * iz;
}

// u^T tile of key tile kt against the heads tile: A = Lp rows (keys in pi order), B = heads^T
__device__ __forceinline__ f32x4 logit_tile(const float (&lpf)[32], const float* HT, int lane)
{
    const int j = lane & 15, G = lane >> 4;
    const float* hp = HT + j * TS + G * TG;
    f32x4 u = z4();
#pragma unroll
    for (int g4 = 0; g4 < 8; ++g4) {
        const float2 lo = *reinterpret_cast<const float2*>(hp + 4 * g4), hi = *reinterpret_cast<const float2*>(hp + 4 * g4 + 2);
        u = mf(lpf[4 * g4 + 0], lo.x, u);
        u = mf(lpf[4 * g4 + 1], lo.y, u);
        u = mf(lpf[4 * g4 + 2], hi.x, u);
        u = mf(lpf[4 * g4 + 3], hi.y, u);
    }
    return u;
}

// DYN: also hl = heads[j] . lw for the lane's query (lw_in[4 g4 + i] = lw[16 g4 + 4 i + G], the staged heads' layout; registers in
// the forward kernel, an LDS copy [G][32] in the backward one, which has none to spare)
template <typename LW>
__device__ __forceinline__ f32x4 logit_tile_dyn(const float (&lpf)[32], const float* HT, int lane, const LW& lw_in, float& hl)
{
    const int j = lane & 15, G = lane >> 4;
    const float* hp = HT + j * TS + G * TG;
    f32x4 u = z4();
    float acc = 0.0f;
#pragma unroll
    for (int g4 = 0; g4 < 8; ++g4) {
        const float2 lo = *reinterpret_cast<const float2*>(hp + 4 * g4), hi = *reinterpret_cast<const float2*>(hp + 4 * g4 + 2);
        u = mf(lpf[4 * g4 + 0], lo.x, u);
        u = mf(lpf[4 * g4 + 1], lo.y, u);
        u = mf(lpf[4 * g4 + 2], hi.x, u);
        u = mf(lpf[4 * g4 + 3], hi.y, u);
        acc = fmaf(hi.y, lw_in[4 * g4 + 3], fmaf(hi.x, lw_in[4 * g4 + 2], fmaf(lo.y, lw_in[4 * g4 + 1], fmaf(lo.x, lw_in[4 * g4], acc))));
    }
    hl = group_sum(acc);
    return u;
}
__device__ __forceinline__ void load_lw_in(const float* dyn, int G, float (&lw_in)[32])
{
#pragma unroll
    for (int g4 = 0; g4 < 8; ++g4)
#pragma unroll
        for (int i = 0; i < 4; ++i) lw_in[4 * g4 + i] = dyn[2 * RE + 16 * g4 + 4 * i + G];
}

__device__ __forceinline__ void load_lp_frags(const ReevalArgs& a, int64_t b, int kt, int lane, float (&lpf)[32])
{
    const int j = lane & 15, G = lane >> 4, n = 16 * kt + 4 * (j & 3) + (j >> 2);
#pragma unroll
    for (int t = 0; t < 32; ++t) lpf[t] = n < a.M ? a.Lp[(b * a.M + n) * a.ld + 4 * t + G] : 0.0f;
}

// z (processed logit) and dz/du of one accumulator value
__device__ __forceinline__ float process_logit(float u, float clip, float inv_temp, float& dzdu)
{
    u *= 0.08838834764831845f;                      // 1 / sqrt(128)
    float z = u, d = 1.0f;
    if (clip > 0.0f) {
        const float e2 = fexp(2.0f * u);
        const float th = 1.0f - 2.0f / (e2 + 1.0f);
        z = clip * th;
        d = clip * (1.0f - th * th);
    }
    dzdu = d * inv_temp * 0.08838834764831845f;
    return z * inv_temp;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// forward: logp[r][t], lse[r][t]
// ---------------------------------------------------------------------------------------------------------------------
// HEADS: the rollout's glimpse outputs (eamrl_reeval.heads) are staged instead of recomputed (training: the entropy pass)
// MC (key chunks): the logits half of a chunk -- heads come from the combined glimpse (HEADS), and instead of log-probs the
// chunk's (maximum, sum of exponentials, sum of e z, z[action] or -inf) go to mc_lpart for k_reeval_mc_combine_logits
template <int RTT, bool HEADS, bool DYN = false, bool MC = false>
__global__ __launch_bounds__(512, 2) void k_reeval_fwd(ReevalArgs a)
{
    __shared__ __attribute__((aligned(16))) float QT[16 * TS];
    __shared__ __attribute__((aligned(16))) float HT[16 * TS];
    __shared__ __attribute__((aligned(16))) float REM[DYN ? 16 * 16 * RTT : 4];
    __shared__ float DYNL[DYN ? 2 * RE : 4];
    __shared__ float RED[8][16], RED2[8][16], RED3[8][16], ZA[16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, G = lane >> 4;
    const Blk blk = decode_block(a);
    const int64_t b = blk.b;
    const int ch = blk.ch, kc = blk.c;
    chunk_view(a, b, kc);
    const int s0 = (int)((int64_t)a.S * ch / a.nchunk), s1 = (int)((int64_t)a.S * (ch + 1) / a.nchunk);
    const int64_t nq = (int64_t)(s1 - s0) * a.T;
    const int64_t ntiles = (nq + 15) / 16;

    float kf[RTT][4], vtf[4 * RTT], lpf[32], lw_in[32];
    const DynLane dl = {DYNL, wv, G};
    if (!HEADS) load_head_frags<RTT>(a, b, wv, lane, kf, vtf);
    if (wv < RTT) load_lp_frags(a, b, wv, lane, lpf);
    if (DYN) {
        fill_dyn_lds(a.dyn, DYNL);
        load_lw_in(a.dyn, G, lw_in);
    }
    const float inv_temp = 1.0f / a.temp;

    for (int64_t tile = 0; tile < ntiles; ++tile) {
        if (!HEADS) build_query_tile(a, b, s0, nq, tile, QT);
        if (DYN) {      // (the previous tile's readers of REM finished before its last barrier)
            const int jq = threadIdx.x >> 5, e4 = threadIdx.x & 31;
            const Q qq = tile_query(a, b, s0, nq, tile, jq);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qq.qi >= 0) v = *reinterpret_cast<const float4*>(a.rem + qq.qi * (RMP * a.nkc) + 4 * e4);
            stage_rem<RTT>(REM, jq, e4, v, qq.qi >= 0);
        }
        const Q q = tile_query(a, b, s0, nq, tile, j);
        uint4 mb = make_uint4(0, 0, 0, 0);
        int act = -1;
        if (q.qi >= 0) {
            mb = *reinterpret_cast<const uint4*>(a.maskbits + q.qi * a.mc_mstride);
            act = (int)a.actions[q.qi] - a.mc_koff;
        }
        if (MC && tid < 16) ZA[tid] = -INFINITY;     // (its readers of the previous tile: these very threads, earlier in program order)
        if (HEADS) {
            const int jq = threadIdx.x >> 5, e4 = threadIdx.x & 31;
            const Q qq = tile_query(a, b, s0, nq, tile, jq);
            const int th = qq.t - a.tstart;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qq.qi >= 0 && qq.active && th < a.heads_T)
                v = *reinterpret_cast<const float4*>(a.heads + (qq.r * a.heads_T + th) * RE + 4 * e4);
            __syncthreads();            // the previous tile's readers of HT are done
            float* hp = HT + jq * TS + e4;
            hp[0] = v.x; hp[TG] = v.y; hp[2 * TG] = v.z; hp[3 * TG] = v.w;
        } else {
            __syncthreads();
            glimpse_tile<RTT, DYN>(kf, vtf, QT, HT, wv, lane, mb, a.M, REM, &dl);
        }
        __syncthreads();
        f32x4 z = z4();
        float mx = -INFINITY;
        if (wv < RTT) {
            f32x4 u;
            if (DYN) {
                float hl;
                u = logit_tile_dyn(lpf, HT, lane, lw_in, hl);
                const f32x4 rw = rem_quad<RTT>(REM, j, wv, G);
#pragma unroll
                for (int r = 0; r < 4; ++r) u[r] = fmaf(rw[r], hl, u[r]);
            } else {
                u = logit_tile(lpf, HT, lane);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n0 = 16 * wv + 4 * r;
                const uint32_t w = (n0 >> 5) == 0 ? mb.x : (n0 >> 5) == 1 ? mb.y : (n0 >> 5) == 2 ? mb.z : mb.w;
                const bool ok = (w >> ((n0 & 31) + G)) & 1u;
                float d;
                z[r] = ok ? process_logit(u[r], a.clip, inv_temp, d) : -INFINITY;
                mx = fmaxf(mx, z[r]);
                if (n0 + G == act) ZA[j] = z[r];
            }
            mx = group_max(mx);
            if (G == 0) RED[wv][j] = mx;
        }
        __syncthreads();
        float M_ = -INFINITY;
#pragma unroll
        for (int w = 0; w < RTT; ++w) M_ = fmaxf(M_, RED[w][j]);
        if (M_ == -INFINITY) M_ = 0.0f;
        if (wv < RTT) {
            float sm = 0.0f, sz = 0.0f;          // sum of e^(z - M) and of e^(z - M) z (for the entropy)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = fexp(z[r] - M_);
                sm += e;
                if (z[r] != -INFINITY) sz = fmaf(e, z[r], sz);
            }
            sm = group_sum(sm);
            if (G == 0) RED2[wv][j] = sm;
            if (a.entropy) {
                sz = group_sum(sz);
                if (G == 0) RED3[wv][j] = sz;
            }
        }
        __syncthreads();
        if (tid < 16) {
            const Q qq = tile_query(a, b, s0, nq, tile, tid);
            if (qq.qi >= 0) {
                float sm = 0.0f, sz = 0.0f, mm = -INFINITY;
                for (int w = 0; w < RTT; ++w) { sm += RED2[w][tid]; mm = fmaxf(mm, RED[w][tid]); }
                if (MC) {        // this chunk's share of the query's log-softmax
                    for (int w = 0; w < RTT; ++w) sz += a.entropy ? RED3[w][tid] : 0.0f;
                    *reinterpret_cast<float4*>(a.mc_lpart + (qq.qi * a.nkc + kc) * 4) = make_float4(mm, sm, sz, ZA[tid]);
                    continue;
                }
                if (mm == -INFINITY) mm = 0.0f;
                const float lse = mm + __builtin_amdgcn_logf(sm) * 0.6931471805599453f;
                if (a.lse) a.lse[qq.qi] = lse;
                a.logp[qq.qi] = qq.active ? ZA[tid] - lse : 0.0f;
                if (a.entropy) {       // -sum_n p_n log p_n = lse - sum_n p_n z_n over the feasible nodes; 0 for a forced step
                    for (int w = 0; w < RTT; ++w) sz += RED3[w][tid];
                    a.entropy[qq.qi] = (qq.active && sm > 0.0f) ? lse - sz / sm : 0.0f;
                }
            }
        }
        // (the next tile's first barrier orders these reads before QT / RED / ZA are rewritten: ZA and RED are only
        //  written after that tile's second barrier)
    }
}

// ---- key chunks, forward: glimpse of one chunk -> per-(query, head) partials; combine -> heads + statistics ------------------------
// mc_part_s [(q * 8 + h) * nkc + c][2] = (maximum or -inf, sum of exp(s - maximum)) over the chunk's feasible keys,
// mc_part_o [(q * nkc + c)][128] = sum_n exp(s_n - maximum) V[n] (unnormalised head outputs, all heads side by side)
template <int RTT, bool DYN = false>
__global__ __launch_bounds__(512, 2) void k_reeval_mc_glimpse_part(ReevalArgs a)
{
    __shared__ __attribute__((aligned(16))) float QT[16 * TS];
    __shared__ __attribute__((aligned(16))) float REM[DYN ? 16 * 16 * RTT : 4];
    __shared__ float DYNL[DYN ? 2 * RE : 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, G = lane >> 4;
    const Blk blk = decode_block(a);
    const int64_t b = blk.b;
    const int ch = blk.ch, kc = blk.c;
    chunk_view(a, b, kc);
    const int s0 = (int)((int64_t)a.S * ch / a.nchunk), s1 = (int)((int64_t)a.S * (ch + 1) / a.nchunk);
    const int64_t nq = (int64_t)(s1 - s0) * a.T;
    const int64_t ntiles = (nq + 15) / 16;
    float kf[RTT][4], vtf[4 * RTT];
    load_head_frags<RTT>(a, b, wv, lane, kf, vtf);
    const DynLane dl = {DYNL, wv, G};
    if (DYN) fill_dyn_lds(a.dyn, DYNL);
    for (int64_t tile = 0; tile < ntiles; ++tile) {
        build_query_tile(a, b, s0, nq, tile, QT);
        if (DYN) {
            const int jq = threadIdx.x >> 5, e4 = threadIdx.x & 31;
            const Q qq = tile_query(a, b, s0, nq, tile, jq);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qq.qi >= 0) v = *reinterpret_cast<const float4*>(a.rem + qq.qi * (RMP * a.nkc) + 4 * e4);
            stage_rem<RTT>(REM, jq, e4, v, qq.qi >= 0);
        }
        const Q q = tile_query(a, b, s0, nq, tile, j);
        uint4 mb = make_uint4(0, 0, 0, 0);
        if (q.qi >= 0) mb = *reinterpret_cast<const uint4*>(a.maskbits + q.qi * a.mc_mstride);
        __syncthreads();
        f32x4 s[RTT];
        float st2[2];
        const float* remq = DYN ? rem_lane<RTT>(REM, j, G) : nullptr;
        head_softmax<RTT, DYN>(kf, QT, wv, lane, mb, a.M, s, remq, &dl, st2);
        f32x4 o = z4();
#pragma unroll
        for (int t = 0; t < 4 * RTT; ++t) o = mf(vtf[t], s[t >> 2][t & 3], o);
        if (DYN) {      // + (sum_n w[n] rem[n]) wv_h, unnormalised like o (the combine scales both by exp(m_c - m) / Z)
            float ra = 0.0f;
#pragma unroll
            for (int kt = 0; kt < RTT; ++kt) {
                const f32x4 rv = rem_at(remq, kt);
#pragma unroll
                for (int r = 0; r < 4; ++r) ra = fmaf(s[kt][r], rv[r], ra);
            }
            ra = group_sum(ra);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = fmaf(ra, dl.wv_out(r), o[r]);
        }
        if (q.qi >= 0) {
            if (G == 0) *reinterpret_cast<float2*>(a.mc_part_s + ((q.qi * RH + wv) * a.nkc + kc) * 2) = make_float2(st2[0], st2[1]);
            *reinterpret_cast<float4*>(a.mc_part_o + (q.qi * a.nkc + kc) * RE + 16 * wv + 4 * G) = make_float4(o[0], o[1], o[2], o[3]);
        }
        __syncthreads();            // QT is rebuilt by the next tile
    }
}

// heads [r][t - tstart][E] (the layout of eamrl_reeval.heads with heads_T = T) and mc_gstat [(q * 8 + h)][2] = (maximum, 1 / sum)
// over all chunks; one thread per (query, four columns)
__global__ void k_reeval_mc_combine_glimpse(ReevalArgs a, float* heads)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t qi = idx >> 5;
    const int e4 = (int)(idx & 31), h = e4 >> 2;
    if (qi >= a.R * a.T) return;
    const int64_t r = qi / a.T;
    const int t = (int)(qi - r * a.T);
    if (t < a.tstart) return;
    const float* ps = a.mc_part_s + (qi * RH + h) * a.nkc * 2;
    float m = -INFINITY;
    for (int c = 0; c < a.nkc; ++c) m = fmaxf(m, ps[2 * c]);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    float Z = 0.0f;
    if (m > -INFINITY)
        for (int c = 0; c < a.nkc; ++c) {
            const float mc = ps[2 * c];
            if (mc == -INFINITY) continue;
            const float f = fexp(mc - m);
            Z = fmaf(f, ps[2 * c + 1], Z);
            const float4 v = *reinterpret_cast<const float4*>(a.mc_part_o + (qi * a.nkc + c) * RE + 4 * e4);
            o.x = fmaf(f, v.x, o.x); o.y = fmaf(f, v.y, o.y); o.z = fmaf(f, v.z, o.z); o.w = fmaf(f, v.w, o.w);
        }
    const float iz = Z > 0.0f ? 1.0f / Z : 0.0f;
    *reinterpret_cast<float4*>(heads + (r * a.T + t - a.tstart) * RE + 4 * e4) = make_float4(o.x * iz, o.y * iz, o.z * iz, o.w * iz);
    if ((e4 & 3) == 0) *reinterpret_cast<float2*>(a.mc_gstat + (qi * RH + h) * 2) = make_float2(m > -INFINITY ? m : 0.0f, iz);
}

// log-softmax over the chunks: mc_lpart [(q * nkc + c)][4] = (maximum, sum of exp, sum of e z, z[action] or -inf) -> logp, lse, entropy
__global__ void k_reeval_mc_combine_logits(ReevalArgs a)
{
    const int64_t qi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= a.R * a.T) return;
    const int t = (int)(qi % a.T);
    const float4* lp = reinterpret_cast<const float4*>(a.mc_lpart) + qi * a.nkc;
    float mm = -INFINITY, za = -INFINITY;
    for (int c = 0; c < a.nkc; ++c) { mm = fmaxf(mm, lp[c].x); za = fmaxf(za, lp[c].w); }
    float sm = 0.0f, sz = 0.0f;
    if (mm > -INFINITY)
        for (int c = 0; c < a.nkc; ++c) {
            if (lp[c].x == -INFINITY) continue;
            const float f = fexp(lp[c].x - mm);
            sm = fmaf(f, lp[c].y, sm);
            sz = fmaf(f, lp[c].z, sz);
        }
    if (mm == -INFINITY) mm = 0.0f;
    const float lse = mm + __builtin_amdgcn_logf(sm) * 0.6931471805599453f;
    const bool active = t >= a.tstart;
    if (a.lse) a.lse[qi] = lse;
    a.logp[qi] = active ? za - lse : 0.0f;
    if (a.entropy) a.entropy[qi] = (active && sm > 0.0f) ? lse - sz / sm : 0.0f;
}

// backward, between the two kernels: dheads = sum of the chunks' partials (into partial 0), rs [(q * 8 + h)] = heads_h . dheads_h --
// the sum_n a[n] (V[n] . dO) over ALL keys that the glimpse backward subtracts; one thread per (query, four columns)
__global__ void k_reeval_mc_sum_dheads(ReevalArgs a, const float* heads)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t qi = idx >> 5, RT = a.R * a.T;
    const int e4 = (int)(idx & 31);
    const bool in = qi < RT;
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f), hv = d;
    if (in) {
        for (int c = 0; c < a.nkc; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(a.dheads + (c * RT + qi) * RE + 4 * e4);
            d.x += v.x; d.y += v.y; d.z += v.z; d.w += v.w;
        }
        *reinterpret_cast<float4*>(a.dheads + qi * RE + 4 * e4) = d;
        const int64_t r = qi / a.T;
        const int t = (int)(qi - r * a.T);
        if (t >= a.tstart) hv = *reinterpret_cast<const float4*>(heads + (r * a.T + t - a.tstart) * RE + 4 * e4);
    }
    float p = fmaf(hv.w, d.w, fmaf(hv.z, d.z, fmaf(hv.y, d.y, hv.x * d.x)));
    p += __shfl_xor(p, 1);
    p += __shfl_xor(p, 2);
    if (in && (e4 & 3) == 0) a.mc_rsq[qi * RH + (e4 >> 2)] = p;
}

template <int RTT>
static int launch_fwd_t(const ReevalArgs& a, hipStream_t st)
{
    if (a.dyn) hipLaunchKernelGGL((k_reeval_fwd<RTT, false, true>), dim3((unsigned)(a.B * a.nchunk)), dim3(512), 0, st, a);
    else if (a.heads) hipLaunchKernelGGL((k_reeval_fwd<RTT, true>), dim3((unsigned)(a.B * a.nchunk)), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((k_reeval_fwd<RTT, false>), dim3((unsigned)(a.B * a.nchunk)), dim3(512), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

bool reeval_supports(int M, int E, int H) { return M >= 1 && M <= 1024 && E == RE && H == RH; }

// ---- key chunks: scratch layout (floats) -------------------------------------------------------------------------------------------
//   heads [R T][E] | part_o [R T][nkc][E] | part_s [R T][8][nkc][2] | gstat [R T][8][2] | lpart [R T][nkc][4] | rsq [R T][8]
// (the per-chunk dq~ rows of the backward live behind the caller's dheads partials: dheads is [2 nkc][R][T][E])
int64_t reeval_scratch_floats(int64_t R, int T, int M)
{
    if (M <= KCH) return 0;
    const int64_t nkc = (M + KCH - 1) / KCH, RT = R * T;
    return RT * (RE + nkc * RE + RH * nkc * 2 + RH * 2 + nkc * 4 + RH);
}
static float* mc_bind(ReevalArgs& a)        // -> heads
{
    const int64_t RT = a.R * (int64_t)a.T;
    a.nkc = (a.M + KCH - 1) / KCH;
    a.mc_mstride = 4 * a.nkc;
    a.mc_koff = 0;
    float* p = a.scratch;
    float* heads = p; p += RT * RE;
    a.mc_part_o = p; p += RT * a.nkc * RE;
    a.mc_part_s = p; p += RT * RH * a.nkc * 2;
    a.mc_gstat = p; p += RT * RH * 2;
    a.mc_lpart = p; p += RT * a.nkc * 4;
    a.mc_rsq = p;
    a.mc_dq = a.dheads ? a.dheads + (int64_t)a.nkc * RT * RE : nullptr;
    return heads;
}

static int launch_fwd_mc(const ReevalArgs& a0, hipStream_t st)
{
    ReevalArgs a = a0;
    float* heads = mc_bind(a);
    const int64_t RT = a.R * (int64_t)a.T;
    const unsigned grid = (unsigned)(a.B * a.nchunk * a.nkc);
    if (a.dyn) hipLaunchKernelGGL((k_reeval_mc_glimpse_part<7, true>), dim3(grid), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((k_reeval_mc_glimpse_part<7>), dim3(grid), dim3(512), 0, st, a);
    hipLaunchKernelGGL(k_reeval_mc_combine_glimpse, dim3((unsigned)((RT * 32 + 255) / 256)), dim3(256), 0, st, a, heads);
    ReevalArgs l = a;
    l.heads = heads; l.heads_T = a.T;
    if (a.dyn) hipLaunchKernelGGL((k_reeval_fwd<7, true, true, true>), dim3(grid), dim3(512), 0, st, l);
    else hipLaunchKernelGGL((k_reeval_fwd<7, true, false, true>), dim3(grid), dim3(512), 0, st, l);
    hipLaunchKernelGGL(k_reeval_mc_combine_logits, dim3((unsigned)((RT + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_reeval_fwd(const ReevalArgs& a0, hipStream_t st)
{
    if (a0.B <= 0 || a0.S <= 0 || a0.T <= 0) return 0;
    if (a0.M > KCH) return launch_fwd_mc(a0, st);
    ReevalArgs a = a0;
    a.nkc = 1; a.mc_koff = 0; a.mc_mstride = 4;
    if (a.M <= 32) return launch_fwd_t<2>(a, st);
    if (a.M <= 64) return launch_fwd_t<4>(a, st);
    return launch_fwd_t<7>(a, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward 1/2: logits.  Recomputes heads and u per tile; du -> dheads (scratch in HBM, read by the glimpse kernel) and
// dLp[n][e] += sum_q du[q][n] heads[q][e] (accumulators of wave w: embedding columns 16 w .. 16 w + 15, all key tiles).
//
// Software pipeline over the tiles (one workgroup per CU, see the glimpse kernel): the query rows of tile i + 1 are
// fetched while tile i is multiplied, and the glimpse of tile i + 1 is computed in the same barrier interval as the
// products that consume tile i's du -- two barriers per tile (three when the normaliser is derived from the rollout's
// log-probs), with MFMA work of every wave on both sides of each.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int DS = 17;        // row stride of the du tile ([n][q])

struct QWalk {          // query l = first, first + 16, ... of the chunk as (start offset, step)
    int sl, t;
    __device__ __forceinline__ void init(int l, int T) { sl = l / T; t = l - sl * T; }
    __device__ __forceinline__ void next(int T)
    {
        t += 16;
        while (t >= T) { t -= T; ++sl; }
    }
};

// One thread's share (query jq, columns 4 e4 ..) of a tile's inputs.  The loads are branch-free: an exec-masked load
// into a register that the loop later overwrites makes the compiler wait for EVERY outstanding load before the overwrite
// (the pending-load state merges over the skipped branch), which serialises the prefetch.  Queries past the end of the
// chunk read query 0 / node 0 instead and are masked by `fl` when the tile is staged.
struct RowPre {
    float4 pa, pb, dh, rem;     // rem: SDVRP only (the row's remaining demands, float4 e4)
    float sc[2];        // state scalars (NC <= 2)
    int fl;             // bit 0: query exists, 1: idxA names a node, 2: idxB names a node, 3: step >= tstart
};

__device__ __forceinline__ void load_idx(const ReevalArgs& a, int qi, int& ia, int& ib)
{
    const int q = max(qi, 0);
    ia = a.idxA[q];
    ib = a.idxB ? a.idxB[q] : -1;
}

template <bool DH, bool DYN = false>
__device__ __forceinline__ void load_rows(const ReevalArgs& a, int64_t b, int e4, int qi, int t, int ia, int ib, RowPre& p)
{
    const int q = max(qi, 0);
    if (DYN) p.rem = *reinterpret_cast<const float4*>(a.rem + (int64_t)q * (RMP * a.nkc) + 4 * e4);
    p.fl = (qi >= 0 ? 1 : 0) | (ia >= 0 ? 2 : 0) | (ib >= 0 ? 4 : 0) | (t >= a.tstart ? 8 : 0);
    p.pa = *reinterpret_cast<const float4*>(a.Pa + (b * a.M + max(ia, 0)) * a.ld + 4 * e4);
    p.pb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.Pb) p.pb = *reinterpret_cast<const float4*>(a.Pb + (b * a.M + max(ib, 0)) * a.ld + 4 * e4);
    p.dh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (DH) p.dh = *reinterpret_cast<const float4*>(a.dheads + (int64_t)q * RE + 4 * e4);
#pragma unroll
    for (int k = 0; k < 2; ++k) p.sc[k] = k < a.NC ? a.sc[(int64_t)k * a.R * a.T + q] : 0.0f;
}

// q~ = 0.25 (Pa[ia] + Pb[ib] + gctx + sum_k sc_k C_k) of the thread's query -> QT (A layout) [, dheads -> DHT]
template <bool DH>
__device__ __forceinline__ void stage_rows(const RowPre& p, const float4& gc, const float* CV, float* QT, float* DHT, int jq, int e4)
{
    const bool ok = p.fl & 1, oa = (p.fl & 3) == 3, ob = (p.fl & 5) == 5;
    float4 v;
    v.x = (oa ? p.pa.x : 0.0f) + (ob ? p.pb.x : 0.0f); v.y = (oa ? p.pa.y : 0.0f) + (ob ? p.pb.y : 0.0f);
    v.z = (oa ? p.pa.z : 0.0f) + (ob ? p.pb.z : 0.0f); v.w = (oa ? p.pa.w : 0.0f) + (ob ? p.pb.w : 0.0f);
    v.x += gc.x; v.y += gc.y; v.z += gc.z; v.w += gc.w;
#pragma unroll
    for (int k = 0; k < 2; ++k) {       // (CV rows >= NC are zero)
        const float4 w = *reinterpret_cast<const float4*>(CV + k * RE + 4 * e4);
        v.x = fmaf(p.sc[k], w.x, v.x); v.y = fmaf(p.sc[k], w.y, v.y);
        v.z = fmaf(p.sc[k], w.z, v.z); v.w = fmaf(p.sc[k], w.w, v.w);
    }
    float* qp = QT + jq * TS + e4;
    qp[0] = ok ? 0.25f * v.x : 0.0f; qp[TG] = ok ? 0.25f * v.y : 0.0f;
    qp[2 * TG] = ok ? 0.25f * v.z : 0.0f; qp[3 * TG] = ok ? 0.25f * v.w : 0.0f;
    if (DH) {
        const bool od = (p.fl & 9) == 9;
        float* dp = DHT + jq * TS + e4;
        dp[0] = od ? p.dh.x : 0.0f; dp[TG] = od ? p.dh.y : 0.0f; dp[2 * TG] = od ? p.dh.z : 0.0f; dp[3 * TG] = od ? p.dh.w : 0.0f;
    }
}

__device__ __forceinline__ uint4 load_mask_words(const ReevalArgs& a, int qi)
{
    const uint4 m = *reinterpret_cast<const uint4*>(a.maskbits + (int64_t)max(qi, 0) * a.mc_mstride);
    return qi >= 0 ? m : make_uint4(0, 0, 0, 0);
}

template <int RTT, bool HEADS, bool DYN = false>
__global__ __launch_bounds__(512, 2) void k_reeval_bwd_logits(ReevalArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* QTB = lds;                               // [2][16][TS]   q~ tiles (A layout)
    float* HTB = QTB + 2 * 16 * TS;                 // [2][16][TS]   heads tiles
    float* LPT = HTB + 2 * 16 * TS;                 // [8 waves][RTT][64 lanes][4]  Lp^T fragments of the dheads product
    float* CV = LPT + 8 * RTT * 256;                // [2][128] state-column vectors
    float* DU = CV + 2 * RE;                        // [16 RTT][DS]
    float* LSE = DU + 16 * RTT * DS;                // [16]
    float* SDR = LSE + 16;                          // DYN: [2][16]  sum_n du[q][n] rem[q][n], summed over the key tiles (LDS atomics)
    float* REMB = SDR + 32;                         // DYN: [2][16][16 RTT]  staged remaining demands
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, G = lane >> 4;
    const int jq = tid >> 5, e4 = tid & 31;
    const Blk blk = decode_block(a);
    const int64_t b = blk.b;
    const int ch = blk.ch, kc = blk.c;
    chunk_view(a, b, kc);
    const int s0 = (int)((int64_t)a.S * ch / a.nchunk), s1 = (int)((int64_t)a.S * (ch + 1) / a.nchunk);
    const int ns = s1 - s0, T = a.T;
    const int ntiles = (ns * T + 15) / 16;
    if (a.nkc > 1) a.dheads += (int64_t)kc * a.R * a.T * RE;        // key chunks: this chunk's partial (summed by k_reeval_mc_sum_dheads)
    // lse == NULL: no forward pass was run -- logp holds the ROLLOUT's log-prob of the chosen node (the same quantity in the
    // rollout kernels' arithmetic), and the normaliser is recovered as z[action] - logp, one extra LDS hand-off per tile
    const bool derive_lse = a.lse == nullptr;

    float kf[RTT][4], vtf[4 * RTT], lpf[32], lw_out[4] = {0.f, 0.f, 0.f, 0.f}, dlw_acc = 0.0f;
    float* LWL = REMB + 2 * 16 * 16 * RTT;         // DYN: [4 lane groups][32]  lw in the staged heads' layout
    float* DYNL = LWL + 128;                        // DYN: [2][E]  wk | wv
    const float* lw_in = LWL + 32 * G;
    const DynLane dl = {DYNL, wv, G};
    if (!HEADS) load_head_frags<RTT>(a, b, wv, lane, kf, vtf);
    load_lp_frags(a, b, wv < RTT ? wv : RTT - 1, lane, lpf);       // (wave 7 of RTT = 7 computes no logits)
    if (DYN) {
        fill_dyn_lds(a.dyn, DYNL);
        if (tid < 128) LWL[tid] = a.dyn[2 * RE + 16 * ((tid & 31) >> 2) + 4 * (tid & 3) + (tid >> 5)];     // (visible after the CV barrier)
#pragma unroll
        for (int r = 0; r < 4; ++r) lw_out[r] = a.dyn[2 * RE + 16 * wv + 4 * G + r];      // the dheads accumulator's columns
    }
    float4* lpt = reinterpret_cast<float4*>(LPT) + wv * RTT * 64 + lane;
#pragma unroll
    for (int t4 = 0; t4 < RTT; ++t4) {              // Lp^T: row e = 16 wv + j, k index = key 4 t + G, t = 4 t4 + i
        float kk[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = 4 * (4 * t4 + i) + G;
            kk[i] = n < a.M ? a.Lp[(b * a.M + n) * a.ld + 16 * wv + j] : 0.0f;
        }
        lpt[t4 * 64] = make_float4(kk[0], kk[1], kk[2], kk[3]);
    }
    for (int i = tid; i < 2 * RE; i += blockDim.x) CV[i] = i < a.NC * RE ? a.Cvec[i] : 0.0f;
    float4 gc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.gctx) gc = *reinterpret_cast<const float4*>(a.gctx + b * RE + 4 * e4);
    f32x4 dLp[RTT];
#pragma unroll
    for (int nt = 0; nt < RTT; ++nt) dLp[nt] = z4();
    const float inv_temp = 1.0f / a.temp;

    // query index r * T + t as a 32-bit value (R * T < 2^31 is checked by the caller), -1 past the chunk's end
    auto qi_of = [&](const QWalk& w) -> int { return w.sl < ns ? ((s0 + w.sl) * (int)a.B + (int)b) * T + w.t : -1; };
    // HEADS: the rollout kept every step's glimpse output (eamrl_state.heads_out) -- the tile's heads rows are fetched
    // instead of the query rows and nothing of the glimpse is recomputed
    auto load_heads = [&](const QWalk& w, RowPre& p) {
        const int qi = qi_of(w);
        const int64_t r = (int64_t)(s0 + min(w.sl, ns - 1)) * a.B + b;
        const int th = max(w.t - a.tstart, 0);
        p.fl = (qi >= 0 ? 1 : 0) | (w.t >= a.tstart && th < a.heads_T ? 8 : 0);
        p.dh = *reinterpret_cast<const float4*>(a.heads + (r * a.heads_T + min(th, a.heads_T - 1)) * RE + 4 * e4);
        if (DYN) p.rem = *reinterpret_cast<const float4*>(a.rem + (int64_t)max(qi, 0) * (RMP * a.nkc) + 4 * e4);
    };
    auto stage_heads = [&](const RowPre& p, int buf) {
        const bool ok = (p.fl & 9) == 9;
        float* dp = HTB + buf * 16 * TS + jq * TS + e4;
        dp[0] = ok ? p.dh.x : 0.0f; dp[TG] = ok ? p.dh.y : 0.0f; dp[2 * TG] = ok ? p.dh.z : 0.0f; dp[3 * TG] = ok ? p.dh.w : 0.0f;
    };
    // ---- prologue: tile 0 staged, indices of tile 1 in flight --------------------------------------------------------
    QWalk wr, wj;                       // the staging role (query jq) and the MFMA role (query j) of this thread
    wr.init(jq, T);
    wj.init(j, T);
    int qr = qi_of(wr), ia = -1, ib = -1;
    {
        RowPre pre;
        if (HEADS) {
            load_heads(wr, pre);
            stage_heads(pre, 0);
            if (DYN) stage_rem<RTT>(REMB, jq, e4, pre.rem, pre.fl & 1);
        } else {
            load_idx(a, qr, ia, ib);
            load_rows<false, DYN>(a, b, e4, qr, wr.t, ia, ib, pre);
            __syncthreads();            // CV
            stage_rows<false>(pre, gc, CV, QTB, nullptr, jq, e4);
            if (DYN) stage_rem<RTT>(REMB, jq, e4, pre.rem, pre.fl & 1);
        }
    }
    wr.next(T);
    qr = qi_of(wr);
    if (!HEADS) load_idx(a, qr, ia, ib);    // tile 1
    int qj = qi_of(wj), tj = wj.t;      // tile 0
    uint4 mb = load_mask_words(a, qj);

    // iteration `tile`: logits, du and the products of tile `tile`, then the glimpse of tile + 1 (iteration -1: only that).
    // Nothing but the mask words, the indices of the next tile's rows and the walkers is live across iterations: each
    // iteration issues its loads first and consumes them behind its own MFMA work.
    for (int tile = -1; tile < ntiles; ++tile) {
        const int cur = tile & 1, nxt = cur ^ 1;
        const float* HT = HTB + cur * 16 * TS;
        uint4 mbn = mb;
        int qjn = qj, tjn = tj;
        if (tile >= 0) {
            RowPre pre;
            if (HEADS) {
                load_heads(wr, pre);                                // heads rows of tile + 1
                wr.next(T);
            } else {
                load_rows<false, DYN>(a, b, e4, qr, wr.t, ia, ib, pre);  // rows of tile + 1 (indices fetched one iteration ago)
                wr.next(T);
                qr = qi_of(wr);
                load_idx(a, qr, ia, ib);                            // indices of tile + 2
            }
            const int qjc = max(qj, 0);
            const bool live = qj >= 0 && tj >= a.tstart;
            const int act_l = (int)a.actions[qjc] - a.mc_koff;
            const float g_l = a.glogp[qjc], lse_l = derive_lse ? a.logp[qjc] : a.lse[qjc];
            wj.next(T);
            qjn = qi_of(wj);
            tjn = wj.t;
            mbn = load_mask_words(a, qjn);
            if (DYN && tid < 16) SDR[cur * 16 + tid] = 0.0f;       // (its last readers were two iterations ago)
            __syncthreads();            // heads of this tile complete; the previous tile's du has been consumed
            f32x4 u = z4(), rw = z4();
            if (wv < RTT) {
                if (DYN) {
                    float hl;
                    u = logit_tile_dyn(lpf, HT, lane, lw_in, hl);
                    rw = rem_quad<RTT>(REMB + cur * 16 * 16 * RTT, j, wv, G);
#pragma unroll
                    for (int r = 0; r < 4; ++r) u[r] = fmaf(rw[r], hl, u[r]);
                } else {
                    u = logit_tile(lpf, HT, lane);
                }
            }
            const int act = qj >= 0 ? act_l : -1;
            const float g = live ? g_l : 0.0f;
            float lse = live ? lse_l : 0.0f;
            float zr[4] = {0.f, 0.f, 0.f, 0.f}, dz[4] = {0.f, 0.f, 0.f, 0.f};
            if (wv < RTT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) zr[r] = process_logit(u[r], a.clip, inv_temp, dz[r]);
            }
            if (derive_lse) {           // the lane that owns the chosen node publishes z[action] - logp for its query
                if (wv < RTT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * wv + 4 * r + G == act && g != 0.0f) LSE[j] = zr[r] - lse;
                }
                __syncthreads();
                if (g != 0.0f) lse = LSE[j];
            }
            if (wv < RTT) {
                float sdr = 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n0 = 16 * wv + 4 * r;
                    const uint32_t w = (n0 >> 5) == 0 ? mb.x : (n0 >> 5) == 1 ? mb.y : (n0 >> 5) == 2 ? mb.z : mb.w;
                    const bool ok = (w >> ((n0 & 31) + G)) & 1u;
                    float du = 0.0f;
                    if (ok && g != 0.0f) {
                        const float p = fexp(zr[r] - lse);
                        du = g * ((n0 + G == act ? 1.0f : 0.0f) - p) * dz[r];
                    }
                    DU[(n0 + G) * DS + j] = du;
                    if (DYN) sdr = fmaf(du, rw[r], sdr);
                }
                if (DYN) {
                    sdr = group_sum(sdr);
                    if (G == 0) atomicAdd(&SDR[cur * 16 + j], sdr);
                }
            }
            if (HEADS) stage_heads(pre, nxt);
            else stage_rows<false>(pre, gc, CV, QTB + nxt * 16 * TS, nullptr, jq, e4);
            if (DYN) stage_rem<RTT>(REMB + nxt * 16 * 16 * RTT, jq, e4, pre.rem, pre.fl & 1);
            __syncthreads();            // du of this tile and q~ of the next one visible
            // wave wv: embedding columns 16 wv .. 16 wv + 15
            f32x4 dh = z4();
#pragma unroll
            for (int t4 = 0; t4 < RTT; ++t4) {
                const float4 ll = lpt[t4 * 64];
                dh = mf(ll.x, DU[(16 * t4 + G) * DS + j], dh);
                dh = mf(ll.y, DU[(16 * t4 + 4 + G) * DS + j], dh);
                dh = mf(ll.z, DU[(16 * t4 + 8 + G) * DS + j], dh);
                dh = mf(ll.w, DU[(16 * t4 + 12 + G) * DS + j], dh);
            }
            if (DYN) {                  // dheads += (sum_n du rem) lw
                const float sq = SDR[cur * 16 + j];
#pragma unroll
                for (int r = 0; r < 4; ++r) dh[r] = fmaf(sq, lw_out[r], dh[r]);
            }
            if (qj >= 0)
                *reinterpret_cast<float4*>(a.dheads + (int64_t)qj * RE + 16 * wv + 4 * G) = make_float4(dh[0], dh[1], dh[2], dh[3]);
            float hb[4];
            const int c = 16 * wv + j;
#pragma unroll
            for (int t = 0; t < 4; ++t) hb[t] = HT[(4 * t + G) * TS + (c & 3) * TG + (c >> 2)];
            if (DYN) {                  // dlw[c] += sum_q (sum_n du rem)[q] heads[q][c], this lane: the queries 4 t + G
#pragma unroll
                for (int t = 0; t < 4; ++t) dlw_acc = fmaf(SDR[cur * 16 + 4 * t + G], hb[t], dlw_acc);
            }
#pragma unroll
            for (int nt = 0; nt < RTT; ++nt)
#pragma unroll
                for (int t = 0; t < 4; ++t) dLp[nt] = mf(DU[(16 * nt + j) * DS + 4 * t + G], hb[t], dLp[nt]);
        } else {
            __syncthreads();            // q~ of tile 0 visible
        }
        mb = mbn; qj = qjn; tj = tjn;
        if (!HEADS && tile + 1 < ntiles)
            glimpse_tile<RTT, DYN>(kf, vtf, QTB + nxt * 16 * TS, HTB + nxt * 16 * TS, wv, lane, mb, a.M, REMB + nxt * 16 * 16 * RTT, &dl);
    }
    if (DYN) {
        dlw_acc = group_sum(dlw_acc);
        if (G == 0) atomicAdd(a.ddyn + 2 * RE + 16 * wv + j, dlw_acc);
    }
    // dLp: lane (column 16 wv + j, G), register r -> key 16 nt + 4 G + r
#pragma unroll
    for (int nt = 0; nt < RTT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = 16 * nt + 4 * G + r;
            if (n < a.M) atomicAdd(a.dLp + (b * a.M + n) * a.ldg + 16 * wv + j, dLp[nt][r]);
        }
}

// One tile (16 queries) of attention backward for head h, shared by the glimpse backward (queries = decode steps) and the
// encoder self-attention backward (queries = the instance's nodes): recomputes a = softmax(K q~ | mask), then
//   da = V dO,  ds = a (da - sum a da),  dq~ = K^T ds (handed to store_dq as soon as it exists, so that its global store
//   runs behind the remaining products: lane (query j, G), register r -> column 4 G + r),
//   dV += a^T dO,  dK += ds^T q~   (accumulators: lane (column 16 h + j, G), register r -> key 16 nt + 4 G + r).
// QT / DHT: the staged q~ and dO tiles (A layout); stg: this wave's 4 x 256-float transposition rows; ktl: this lane's K^T
// fragments in LDS ([t4] at stride 64 float4).
// DYN (SDVRP, see DynLane): REM = the tile's staged remaining demands, srw = 32 wave-private LDS floats, dwk / dwv = this lane's
// running sums of the gradients of wk[16 h + j] / wv[16 h + j] over the queries 4 t + G
// given (key chunks): (maximum, 1 / sum, rs) of the lane's query over ALL chunks -- the softmax is not renormalised here and the
// row sum rs = sum_n a[n] da[n] = heads_h . dO_h comes from k_reeval_mc_sum_dheads
template <int RTT, bool DYN = false, typename StoreDq>
__device__ __forceinline__ void attention_bwd_tile(const float (&kf)[RTT][4], const float (&vaf)[RTT][4], const float4* ktl,
                                                   const float* QT, const float* DHT, float* stg, int h, int lane,
                                                   const uint4& mb, int M, f32x4 (&dV)[RTT], f32x4 (&dK)[RTT], StoreDq store_dq,
                                                   const float* REM = nullptr, const DynLane* dl = nullptr, float* srw = nullptr,
                                                   float* dwk = nullptr, float* dwv = nullptr, const float* given = nullptr)
{
    const int j = lane & 15, G = lane >> 4;
    f32x4 s[RTT], da[RTT];
    const float* remq = DYN ? rem_lane<RTT>(REM, j, G) : nullptr;
    const float iz = head_softmax<RTT, DYN>(kf, QT, h, lane, mb, M, s, remq, dl, nullptr, given);
    const float* dp = DHT + j * TS + G * TG + 4 * h;
    const float2 dlo = *reinterpret_cast<const float2*>(dp), dhi = *reinterpret_cast<const float2*>(dp + 2);
#pragma unroll
    for (int kt = 0; kt < RTT; ++kt) da[kt] = mf(vaf[kt][0], dlo.x, z4());
#pragma unroll
    for (int kt = 0; kt < RTT; ++kt) da[kt] = mf(vaf[kt][1], dlo.y, da[kt]);
#pragma unroll
    for (int kt = 0; kt < RTT; ++kt) da[kt] = mf(vaf[kt][2], dhi.x, da[kt]);
#pragma unroll
    for (int kt = 0; kt < RTT; ++kt) da[kt] = mf(vaf[kt][3], dhi.y, da[kt]);
    if (DYN) {                                       // da[n] += rem[n] (wv_h . dO_h)
        const float e1 = group_sum(fmaf(dhi.y, dl->wv_in(3), fmaf(dhi.x, dl->wv_in(2), fmaf(dlo.y, dl->wv_in(1), dlo.x * dl->wv_in(0)))));
#pragma unroll
        for (int kt = 0; kt < RTT; ++kt) {
            const f32x4 rv = rem_at(remq, kt);
#pragma unroll
            for (int r = 0; r < 4; ++r) da[kt][r] = fmaf(rv[r], e1, da[kt][r]);
        }
    }
    float rs = 0.0f, ar = 0.0f, sr = 0.0f;
#pragma unroll
    for (int kt = 0; kt < RTT; ++kt) {
        f32x4 rv = z4();
        if (DYN) rv = rem_at(remq, kt);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[kt][r] *= iz;                          // a
            rs = fmaf(s[kt][r], da[kt][r], rs);
            if (DYN) ar = fmaf(s[kt][r], rv[r], ar);
        }
    }
    rs = given ? given[2] : group_sum(rs);
#pragma unroll
    for (int kt = 0; kt < RTT; ++kt) {
        f32x4 rv = z4();
        if (DYN) rv = rem_at(remq, kt);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            da[kt][r] = s[kt][r] * (da[kt][r] - rs);      // ds
            if (DYN) sr = fmaf(da[kt][r], rv[r], sr);
        }
    }
    if (DYN) {
        ar = group_sum(ar);                          // sum_n a[n] rem[n]
        sr = group_sum(sr);                          // sum_n ds[n] rem[n]
        if (G == 0) { srw[j] = sr; srw[16 + j] = ar; }
    }
    // staging of key tile 0 (element (key kappa, query q) of a tile at kappa * 16 + (q & 3) * 4 + (q >> 2))
    float* wp = stg + G * 16 + (j & 3) * 4 + (j >> 2);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        wp[r * 64] = s[0][r];
        wp[256 + r * 64] = da[0][r];
    }
    f32x4 dq = z4();
#pragma unroll
    for (int t4 = 0; t4 < RTT; ++t4) {
        const float4 kk = ktl[t4 * 64];
        dq = mf(kk.x, da[t4][0], dq);
        dq = mf(kk.y, da[t4][1], dq);
        dq = mf(kk.z, da[t4][2], dq);
        dq = mf(kk.w, da[t4][3], dq);
    }
    if (DYN) {                                       // dq~_h += (sum_n ds rem) wk_h
#pragma unroll
        for (int r = 0; r < 4; ++r) dq[r] = fmaf(sr, dl->wk_out(r), dq[r]);
    }
    store_dq(dq);
    // B operands of the two "sum over queries" products: dO_h [q][e] and q~_h [q][d], column j, k index q = 4 t + G
    float dhb[4], qb[4];
    const int c = 16 * h + j;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        dhb[t] = DHT[(4 * t + G) * TS + (c & 3) * TG + (c >> 2)];
        qb[t] = QT[(4 * t + G) * TS + (c & 3) * TG + (c >> 2)];
    }
    if (DYN) {          // dwk[c] += sum_q (sum_n ds rem)[q] q~[q][c],  dwv[c] += sum_q (sum_n a rem)[q] dO[q][c]
        __builtin_amdgcn_wave_barrier();            // srw: written above by this wavefront (LDS keeps a wave's accesses in order)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            *dwk = fmaf(srw[4 * t + G], qb[t], *dwk);
            *dwv = fmaf(srw[16 + 4 * t + G], dhb[t], *dwv);
        }
    }
#pragma unroll
    for (int nt = 0; nt < RTT; ++nt) {
        if (nt + 1 < RTT) {
            float* wq = wp + ((nt + 1) & 1) * 512;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                wq[r * 64] = s[nt + 1][r];
                wq[256 + r * 64] = da[nt + 1][r];
            }
        }
        __builtin_amdgcn_wave_barrier();
        const float* rp = stg + (nt & 1) * 512 + j * 16 + G * 4;
        const float4 at = *reinterpret_cast<const float4*>(rp);
        const float4 dt = *reinterpret_cast<const float4*>(rp + 256);
        __builtin_amdgcn_wave_barrier();
        dV[nt] = mf(at.x, dhb[0], dV[nt]);
        dK[nt] = mf(dt.x, qb[0], dK[nt]);
        dV[nt] = mf(at.y, dhb[1], dV[nt]);
        dK[nt] = mf(dt.y, qb[1], dK[nt]);
        dV[nt] = mf(at.z, dhb[2], dV[nt]);
        dK[nt] = mf(dt.z, qb[2], dK[nt]);
        dV[nt] = mf(at.w, dhb[3], dV[nt]);
        dK[nt] = mf(dt.w, qb[3], dK[nt]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward 2/2: glimpse.  Wave h = head h: recomputes the softmax, da = V dheads, ds = a (da - sum a da), dq~ = K^T ds,
// dV[n][e] += sum_q a[q][n] dheads[q][e], dK[n][d] += sum_q ds[q][n] q~[q][d].
//
// One workgroup per CU (the fragments take the whole register file), so nothing else on the CU hides a serial chain
// inside the tile loop; it is pipelined by hand:
//   * the rows of tile i + 1 (and the indices of tile i + 2) are fetched into registers while tile i is multiplied, and the
//     staged query / dheads tiles are double-buffered: one barrier per tile;
//   * the accumulator -> A-operand transpositions of the softmax weights and their gradients go through per-wave staging
//     rows laid out for one 16-byte read per lane ([key][query & 3][query >> 2]), written one key tile ahead of the
//     products that consume them, with no waits between (LDS executes one wave's accesses in order);
//   * dq~ is not scattered inside the kernel: it overwrites the tile's dheads rows (the scratch is read one tile ahead),
//     and k_reeval_bwd_gather adds the rows into dPa / dPb / dgctx / dCvec afterwards.  The K^T fragments of the dq~
//     product live in LDS, which leaves their 28 registers to the prefetch.
// (Round-2 history: with the gathers, two barriers and an LDS-atomic scatter inside the loop this kernel took 37.3 ms at the
//  POMO training size; 15.6 ms now, plus 2.0 ms of k_reeval_bwd_gather.)
// ---------------------------------------------------------------------------------------------------------------------
template <int RTT, bool DYN = false, bool MC = false>
__global__ __launch_bounds__(512, 2) void k_reeval_bwd_glimpse(ReevalArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* QTB = lds;                               // [2][16][TS]   q~ tiles (A layout)
    float* DHB = QTB + 2 * 16 * TS;                 // [2][16][TS]   dheads tiles
    float* STG = DHB + 2 * 16 * TS;                 // [8 waves][2 parities][a | ds][16 keys][16 queries]
    float* KTL = STG + 8 * 4 * 256;                 // [8 waves][RTT][64 lanes][4]  K^T fragments of the dq~ product
    float* CV = KTL + 8 * RTT * 256;                // [2][128] state-column vectors
    float* SRW = CV + 2 * RE;                       // DYN: [8 waves][32]  per-query sums of the tile, wave-private
    float* REMB = SRW + 8 * 32;                     // DYN: [2][16][16 RTT]  staged remaining demands
    float* DYNL = REMB + 2 * 16 * 16 * RTT;         // DYN: [2][E]  wk | wv
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, G = lane >> 4, pi = 4 * (j & 3) + (j >> 2);
    const int jq = tid >> 5, e4 = tid & 31;
    const Blk blk = decode_block(a);
    const int64_t b = blk.b;
    const int ch = blk.ch, kc = blk.c;
    chunk_view(a, b, kc);
    const int s0 = (int)((int64_t)a.S * ch / a.nchunk), s1 = (int)((int64_t)a.S * (ch + 1) / a.nchunk);
    const int ns = s1 - s0, T = a.T;
    const int ntiles = (ns * T + 15) / 16;
    const int h = wv;
    float* stg = STG + wv * 4 * 256;

    float kf[RTT][4], vaf[RTT][4];
#pragma unroll
    for (int kt = 0; kt < RTT; ++kt) {
        const int n = 16 * kt + pi;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            kf[kt][t] = n < a.M ? a.K[(b * a.M + n) * a.ld + 16 * h + 4 * t + G] : 0.0f;
            vaf[kt][t] = n < a.M ? a.V[(b * a.M + n) * a.ld + 16 * h + 4 * t + G] : 0.0f;
        }
    }
    float4* ktl = reinterpret_cast<float4*>(KTL) + wv * RTT * 64 + lane;
#pragma unroll
    for (int t4 = 0; t4 < RTT; ++t4) {              // K^T: row d = j, k index = key 4 t + G, t = 4 t4 + i
        float kk[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = 4 * (4 * t4 + i) + G;
            kk[i] = n < a.M ? a.K[(b * a.M + n) * a.ld + 16 * h + j] : 0.0f;
        }
        ktl[t4 * 64] = make_float4(kk[0], kk[1], kk[2], kk[3]);
    }
    for (int i = tid; i < 2 * RE; i += blockDim.x) CV[i] = i < a.NC * RE ? a.Cvec[i] : 0.0f;
    float4 gc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.gctx) gc = *reinterpret_cast<const float4*>(a.gctx + b * RE + 4 * e4);
    f32x4 dV[RTT], dK[RTT];
#pragma unroll
    for (int nt = 0; nt < RTT; ++nt) { dV[nt] = z4(); dK[nt] = z4(); }
    const DynLane dl = {DYNL, h, G};
    float dwk_acc = 0.0f, dwv_acc = 0.0f;
    if (DYN) fill_dyn_lds(a.dyn, DYNL);             // (visible after the prologue's barrier)

    // query index r * T + t as a 32-bit value (R * T < 2^31 is checked by the caller), -1 past the chunk's end
    auto qi_of = [&](const QWalk& w) -> int { return w.sl < ns ? ((s0 + w.sl) * (int)a.B + (int)b) * T + w.t : -1; };
    // ---- prologue: rows of tile 0, indices of tile 1, mask words of tile 0 ------------------------------------------------
    QWalk wr, wj;                       // the staging role (query jq) and the MFMA role (query j) of this thread
    wr.init(jq, T);
    wj.init(j, T);
    int qr = qi_of(wr), ia, ib;
    load_idx(a, qr, ia, ib);
    RowPre pre;
    load_rows<true, DYN>(a, b, e4, qr, wr.t, ia, ib, pre);
    wr.next(T);
    qr = qi_of(wr);
    load_idx(a, qr, ia, ib);
    int qj = qi_of(wj);
    uint4 mb = load_mask_words(a, qj);
    // key chunks: the query's statistics over all chunks (maximum, 1 / sum, rs), fetched one tile ahead like the mask words
    const int64_t RT = a.R * (int64_t)a.T;
    auto load_given = [&](int q, float (&g)[3]) {
        const int qc = max(q, 0);
        const float2 st2 = *reinterpret_cast<const float2*>(a.mc_gstat + ((int64_t)qc * RH + h) * 2);
        const float rsv = a.mc_rsq[(int64_t)qc * RH + h];
        g[0] = q >= 0 ? st2.x : 0.0f; g[1] = q >= 0 ? st2.y : 0.0f; g[2] = q >= 0 ? rsv : 0.0f;
    };
    float gv[3] = {0.f, 0.f, 0.f};
    if (MC) load_given(qj, gv);
    __syncthreads();                    // CV, KTL

    for (int tile = 0; tile < ntiles; ++tile) {
        const int cur = tile & 1;
        stage_rows<true>(pre, gc, CV, QTB + cur * 16 * TS, DHB + cur * 16 * TS, jq, e4);
        if (DYN) stage_rem<RTT>(REMB + cur * 16 * 16 * RTT, jq, e4, pre.rem, pre.fl & 1);
        // rows of tile + 1 (their indices arrived during the previous tile), indices of tile + 2, mask words of tile + 1
        load_rows<true, DYN>(a, b, e4, qr, wr.t, ia, ib, pre);
        wr.next(T);
        qr = qi_of(wr);
        load_idx(a, qr, ia, ib);
        const int qj_cur = qj;
        const uint4 mb_cur = mb;
        const float gv_cur[3] = {gv[0], gv[1], gv[2]};
        wj.next(T);
        qj = qi_of(wj);
        mb = load_mask_words(a, qj);
        if (MC) load_given(qj, gv);
        __syncthreads();
        const float* QT = QTB + cur * 16 * TS;
        const float* DHT = DHB + cur * 16 * TS;
        attention_bwd_tile<RTT, DYN>(kf, vaf, ktl, QT, DHT, stg, h, lane, mb_cur, a.M, dV, dK, [&](const f32x4& dq) {
            if (qj_cur >= 0) {      // dq~ takes the place of the tile's dheads row (read one tile ahead); key chunks: its own buffer
                float* dst = MC ? a.mc_dq + ((int64_t)kc * RT + qj_cur) * RE : a.dheads + (int64_t)qj_cur * RE;
                *reinterpret_cast<float4*>(dst + 16 * h + 4 * G) = make_float4(0.25f * dq[0], 0.25f * dq[1], 0.25f * dq[2], 0.25f * dq[3]);
            }
        }, REMB + cur * 16 * 16 * RTT, &dl, SRW + wv * 32, &dwk_acc, &dwv_acc, MC ? gv_cur : nullptr);
    }
    if (DYN) {
        dwk_acc = group_sum(dwk_acc);
        dwv_acc = group_sum(dwv_acc);
        if (G == 0) {
            atomicAdd(a.ddyn + 16 * h + j, dwk_acc);
            atomicAdd(a.ddyn + RE + 16 * h + j, dwv_acc);
        }
    }
#pragma unroll
    for (int nt = 0; nt < RTT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = 16 * nt + 4 * G + r;
            if (n < a.M) {
                atomicAdd(a.dV + (b * a.M + n) * a.ldg + 16 * h + j, dV[nt][r]);
                atomicAdd(a.dK + (b * a.M + n) * a.ldg + 16 * h + j, dK[nt][r]);
            }
        }
}

// dq~ rows -> dPa / dPb / dgctx / dCvec without floating-point LDS atomics (ds_add_f32 retires about one lane every two
// cycles on this part: the 2.6e9 lane-adds of the POMO step took 12 ms however the banks were mapped).  Per (instance,
// chunk of starts) and index set: counting sort of the chunk's queries by node in LDS (integer atomics, one per query),
// then each node's rows are summed in registers by one wavefront (bins of more than GBIG rows: by all eight) and added
// to the gradient row once.  Bin M of the first index set collects the active queries that name no node, so that the
// sets' rows also sum to dgctx and (weighted by the state scalars) dCvec.
constexpr int GU = 8;           // rows in flight per wavefront
constexpr int GBIG = 512;
constexpr int GCAP = 24576;     // queries per workgroup (LDS: 4 bytes each)
constexpr int GBINS = 1032;     // bins: up to 1024 nodes (+ "no node", + the end offset)

__global__ __launch_bounds__(512) void k_reeval_bwd_gather(ReevalArgs a, int nchunk)
{
    extern __shared__ __attribute__((aligned(16))) int ldsi[];
    int* OFF = ldsi;                // [M + 2] counts, then offsets
    int* CUR = OFF + GBINS;         // [M + 1] cursors
    int* ORD = CUR + GBINS;         // [nq] query indices grouped by node
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t b = blockIdx.x / nchunk;
    const int ch = (int)(blockIdx.x - b * nchunk);
    const int s0 = (int)((int64_t)a.S * ch / nchunk), s1 = (int)((int64_t)a.S * (ch + 1) / nchunk);
    const int T = a.T, nq = (s1 - s0) * T, M = a.M;
    float2 gs = make_float2(0.f, 0.f), cs[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};

    for (int pass = 0; pass < 2; ++pass) {
        const int32_t* idx = pass ? a.idxB : a.idxA;
        float* dP = pass ? a.dPb : a.dPa;
        if (!idx) break;
        const int nb = pass ? M : M + 1;
        __syncthreads();
        for (int i = tid; i < 2 * GBINS; i += blockDim.x) OFF[i] = 0;
        __syncthreads();
        for (int l = tid; l < nq; l += blockDim.x) {
            const int sl = l / T, t = l - sl * T;
            if (t < a.tstart) continue;
            const int n = idx[((int64_t)(s0 + sl) * a.B + b) * T + t];
            if (n >= 0 || !pass) atomicAdd(OFF + (n >= 0 ? n : M), 1);
        }
        __syncthreads();
        if (wv == 0) {              // exclusive prefix over the nb bins: 128 per pass (two per lane), the running total carried over
            int carry = 0;
            for (int base = 0; base <= nb; base += 128) {
                const int i0 = base + 2 * lane, i1 = i0 + 1;
                const int c0 = i0 < nb ? OFF[i0] : 0, c1 = i1 < nb ? OFF[i1] : 0;
                int x = c0 + c1;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int y = __shfl_up(x, d);
                    if (lane >= d) x += y;
                }
                const int ex = carry + x - c0 - c1;
                if (i0 <= nb) { OFF[i0] = ex; CUR[i0] = ex; }
                if (i1 <= nb) { OFF[i1] = ex + c0; CUR[i1] = ex + c0; }
                carry += __shfl(x, 63);
            }
        }
        __syncthreads();
        for (int l = tid; l < nq; l += blockDim.x) {
            const int sl = l / T, t = l - sl * T;
            if (t < a.tstart) continue;
            const int qi = ((s0 + sl) * (int)a.B + (int)b) * T + t;
            const int n = idx[qi];
            if (n >= 0 || !pass) ORD[atomicAdd(CUR + (n >= 0 ? n : M), 1)] = qi;
        }
        __syncthreads();
        for (int n = 0; n < nb; ++n) {
            const int beg = OFF[n], end = OFF[n + 1];
            const bool big = end - beg > GBIG;
            if (end == beg || (!big && (n & 7) != wv)) continue;
            float2 acc = make_float2(0.f, 0.f);
            for (int i = beg + (big ? wv * GU : 0); i < end; i += big ? 8 * GU : GU) {
                float2 v[GU];
                float sc[GU][2];
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    v[u] = make_float2(0.f, 0.f);
                    sc[u][0] = sc[u][1] = 0.0f;
                    if (i + u < end) {
                        const int q = __builtin_amdgcn_readfirstlane(ORD[i + u]);
                        if (a.nkc > 1) {        // key chunks: the query's gradient is the sum of the chunks' dq~ rows
                            for (int c = 0; c < a.nkc; ++c) {
                                const float2 w2 = reinterpret_cast<const float2*>(a.mc_dq + ((int64_t)c * a.R * T + q) * RE)[lane];
                                v[u].x += w2.x; v[u].y += w2.y;
                            }
                        } else {
                            v[u] = reinterpret_cast<const float2*>(a.dheads + (int64_t)q * RE)[lane];
                        }
                        if (!pass) {
#pragma unroll
                            for (int k = 0; k < 2; ++k)
                                if (k < a.NC) sc[u][k] = a.sc[(int64_t)k * a.R * T + q];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    acc.x += v[u].x; acc.y += v[u].y;
                    if (!pass) {
#pragma unroll
                        for (int k = 0; k < 2; ++k) { cs[k].x = fmaf(sc[u][k], v[u].x, cs[k].x); cs[k].y = fmaf(sc[u][k], v[u].y, cs[k].y); }
                    }
                }
            }
            if (n < M) {
                atomicAdd(dP + (b * M + n) * a.ldg + 2 * lane, acc.x);
                atomicAdd(dP + (b * M + n) * a.ldg + 2 * lane + 1, acc.y);
            }
            if (!pass) { gs.x += acc.x; gs.y += acc.y; }
        }
    }
    if (a.dgctx) {
        atomicAdd(a.dgctx + b * RE + 2 * lane, gs.x);
        atomicAdd(a.dgctx + b * RE + 2 * lane + 1, gs.y);
    }
    for (int k = 0; k < a.NC && k < 2; ++k) {
        atomicAdd(a.dCvec + k * RE + 2 * lane, cs[k].x);
        atomicAdd(a.dCvec + k * RE + 2 * lane + 1, cs[k].y);
    }
}

template <int RTT>
static int launch_bwd_t(const ReevalArgs& a, hipStream_t st)
{
    const unsigned grid = (unsigned)(a.B * a.nchunk);
    const bool dyn = a.dyn != nullptr;
    const size_t ldl = (4 * 16 * (size_t)TS + 8 * RTT * 256 + 2 * RE + 16 * RTT * DS + 16 + 32 + (dyn ? 2 * 16 * 16 * RTT + 128 + 2 * RE : 0)) * sizeof(float);
    auto kl = dyn ? k_reeval_bwd_logits<RTT, false, true> : a.heads ? k_reeval_bwd_logits<RTT, true> : k_reeval_bwd_logits<RTT, false>;
    if (ldl > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kl), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldl) != hipSuccess)
        return EAMRL_E_LAUNCH;
    hipLaunchKernelGGL(kl, dim3(grid), dim3(512), ldl, st, a);
    const size_t lds = (4 * 16 * (size_t)TS + 8 * 4 * 256 + 8 * RTT * 256 + 2 * RE + 8 * 32 + (dyn ? 2 * 16 * 16 * RTT + 2 * RE : 0)) * sizeof(float);
    auto k = dyn ? k_reeval_bwd_glimpse<RTT, true> : k_reeval_bwd_glimpse<RTT, false>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return EAMRL_E_LAUNCH;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, st, a);
    // the gather kernel sorts its chunk's queries in LDS: at most GCAP of them per workgroup
    int ng = a.nchunk;
    while ((int64_t)((a.S + ng - 1) / ng) * a.T > GCAP && ng < a.S) ++ng;
    if ((int64_t)((a.S + ng - 1) / ng) * a.T > GCAP) return EAMRL_E_LAUNCH;
    const size_t ldg = (2 * GBINS + (size_t)((a.S + ng - 1) / ng) * a.T) * sizeof(int);
    if (ldg > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(k_reeval_bwd_gather),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldg) != hipSuccess)
        return EAMRL_E_LAUNCH;
    hipLaunchKernelGGL(k_reeval_bwd_gather, dim3((unsigned)(a.B * ng)), dim3(512), ldg, st, a, ng);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

// key chunks: the forward call has left heads, the glimpse statistics and lse in the scratch / the caller's buffers
static int launch_bwd_mc(const ReevalArgs& a0, hipStream_t st)
{
    constexpr int RTT = 7;
    ReevalArgs a = a0;
    float* heads = mc_bind(a);
    const int64_t RT = a.R * (int64_t)a.T;
    const unsigned grid = (unsigned)(a.B * a.nchunk * a.nkc);
    ReevalArgs l = a;
    l.heads = heads; l.heads_T = a.T;
    const bool dyn = a.dyn != nullptr;
    const size_t ldl = (4 * 16 * (size_t)TS + 8 * RTT * 256 + 2 * RE + 16 * RTT * DS + 16 + 32 + (dyn ? 2 * 16 * 16 * RTT + 128 + 2 * RE : 0)) * sizeof(float);
    auto kl = dyn ? k_reeval_bwd_logits<RTT, true, true> : k_reeval_bwd_logits<RTT, true>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kl), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldl) != hipSuccess)
        return EAMRL_E_LAUNCH;
    hipLaunchKernelGGL(kl, dim3(grid), dim3(512), ldl, st, l);
    hipLaunchKernelGGL(k_reeval_mc_sum_dheads, dim3((unsigned)((RT * 32 + 255) / 256)), dim3(256), 0, st, a, heads);
    const size_t lds = (4 * 16 * (size_t)TS + 8 * 4 * 256 + 8 * RTT * 256 + 2 * RE + 8 * 32 + (dyn ? 2 * 16 * 16 * RTT + 2 * RE : 0)) * sizeof(float);
    auto k = dyn ? k_reeval_bwd_glimpse<RTT, true, true> : k_reeval_bwd_glimpse<RTT, false, true>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return EAMRL_E_LAUNCH;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, st, a);
    int ng = a.nchunk;
    while ((int64_t)((a.S + ng - 1) / ng) * a.T > GCAP && ng < a.S) ++ng;
    if ((int64_t)((a.S + ng - 1) / ng) * a.T > GCAP) return EAMRL_E_LAUNCH;
    const size_t ldg = (2 * GBINS + (size_t)((a.S + ng - 1) / ng) * a.T) * sizeof(int);
    if (ldg > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(k_reeval_bwd_gather),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldg) != hipSuccess)
        return EAMRL_E_LAUNCH;
    hipLaunchKernelGGL(k_reeval_bwd_gather, dim3((unsigned)(a.B * ng)), dim3(512), ldg, st, a, ng);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_reeval_bwd(const ReevalArgs& a0, hipStream_t st)
{
    if (a0.B <= 0 || a0.S <= 0 || a0.T <= 0) return 0;
    if (a0.M > KCH) return launch_bwd_mc(a0, st);
    ReevalArgs a = a0;
    a.nkc = 1; a.mc_koff = 0; a.mc_mstride = 4;
    if (a.M <= 32) return launch_bwd_t<2>(a, st);
    if (a.M <= 64) return launch_bwd_t<4>(a, st);
    return launch_bwd_t<7>(a, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// Encoder self-attention backward (MultiHeadAttention.forward, rl4co/models/nn/attention.py:112-136, under loss.backward()):
// qkv [B][N][3E] packed "b s (three h d)", dO [B][N][E] -> dqkv [B][N][3E].  The same five products as the glimpse backward
// above with the instance's own N nodes as queries (no mask, scale 1/4 folded into the staged q): wave h = head h,
//   a = softmax(K q~),  da = V dO,  ds = a (da - sum a da),  dq = 0.25 K^T ds,  dV += a^T dO,  dK += ds^T q~
// over the ceil(N / 16) query tiles of the instance; one workgroup owns an instance, so dK / dV / dq are stored, not added.
// ---------------------------------------------------------------------------------------------------------------------
template <int RTT>
__global__ __launch_bounds__(512, 2) void k_mha_encoder_bwd(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                            float* __restrict__ dqkv, int N)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* QTB = lds;                               // [2][16][TS]   q~ tiles (A layout)
    float* DHB = QTB + 2 * 16 * TS;                 // [2][16][TS]   dO tiles
    float* STG = DHB + 2 * 16 * TS;                 // [8 waves][2 parities][a | ds][16 keys][16 queries]
    float* KTL = STG + 8 * 4 * 256;                 // [8 waves][RTT][64 lanes][4]  K^T fragments of the dq product
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, G = lane >> 4, pi = 4 * (j & 3) + (j >> 2);
    const int jq = tid >> 5, e4 = tid & 31;
    const int64_t b = blockIdx.x;
    const int h = wv;
    const int ntiles = (N + 15) / 16;
    const float* qkvb = qkv + b * N * (3 * RE);
    float* stg = STG + wv * 4 * 256;

    float kf[RTT][4], vaf[RTT][4];
#pragma unroll
    for (int kt = 0; kt < RTT; ++kt) {
        const int n = 16 * kt + pi;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            kf[kt][t] = n < N ? qkvb[(int64_t)n * (3 * RE) + RE + 16 * h + 4 * t + G] : 0.0f;
            vaf[kt][t] = n < N ? qkvb[(int64_t)n * (3 * RE) + 2 * RE + 16 * h + 4 * t + G] : 0.0f;
        }
    }
    float4* ktl = reinterpret_cast<float4*>(KTL) + wv * RTT * 64 + lane;
#pragma unroll
    for (int t4 = 0; t4 < RTT; ++t4) {              // K^T: row d = j, k index = key 4 t + G, t = 4 t4 + i
        float kk[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = 4 * (4 * t4 + i) + G;
            kk[i] = n < N ? qkvb[(int64_t)n * (3 * RE) + RE + 16 * h + j] : 0.0f;
        }
        ktl[t4 * 64] = make_float4(kk[0], kk[1], kk[2], kk[3]);
    }
    uint4 mb;                                       // keys n < N
    {
        auto word = [&](int w) -> uint32_t { const int r = N - 32 * w; return r >= 32 ? 0xffffffffu : r <= 0 ? 0u : (1u << r) - 1u; };
        mb = make_uint4(word(0), word(1), word(2), word(3));
    }
    f32x4 dV[RTT], dK[RTT];
#pragma unroll
    for (int nt = 0; nt < RTT; ++nt) { dV[nt] = z4(); dK[nt] = z4(); }

    auto fetch = [&](int tile, float4& q, float4& d) {
        const int n = 16 * tile + jq;
        const int nc = n < N ? n : 0;
        const float4 vq = *reinterpret_cast<const float4*>(qkvb + (int64_t)nc * (3 * RE) + 4 * e4);
        const float4 vd = *reinterpret_cast<const float4*>(dout + (b * N + nc) * RE + 4 * e4);
        const bool ok = n < N && tile < ntiles;
        q = ok ? vq : make_float4(0.f, 0.f, 0.f, 0.f);
        d = ok ? vd : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float4 pq, pd;
    fetch(0, pq, pd);
    __syncthreads();                    // KTL
    for (int tile = 0; tile < ntiles; ++tile) {
        const int cur = tile & 1;
        {
            float* qp = QTB + cur * 16 * TS + jq * TS + e4;
            qp[0] = 0.25f * pq.x; qp[TG] = 0.25f * pq.y; qp[2 * TG] = 0.25f * pq.z; qp[3 * TG] = 0.25f * pq.w;
            float* dp = DHB + cur * 16 * TS + jq * TS + e4;
            dp[0] = pd.x; dp[TG] = pd.y; dp[2 * TG] = pd.z; dp[3 * TG] = pd.w;
        }
        fetch(tile + 1 < ntiles ? tile + 1 : tile, pq, pd);
        __syncthreads();
        const float* QT = QTB + cur * 16 * TS;
        const float* DHT = DHB + cur * 16 * TS;
        attention_bwd_tile<RTT>(kf, vaf, ktl, QT, DHT, stg, h, lane, mb, N, dV, dK, [&](const f32x4& dq) {
            if (16 * tile + j < N)      // lane (query j, G), register r -> d = 4 G + r
                *reinterpret_cast<float4*>(dqkv + (b * N + 16 * tile + j) * (3 * RE) + 16 * h + 4 * G) =
                    make_float4(0.25f * dq[0], 0.25f * dq[1], 0.25f * dq[2], 0.25f * dq[3]);
        });
    }
    // dV / dK: lane (column 16 h + j, G), register r -> key 16 nt + 4 G + r
#pragma unroll
    for (int nt = 0; nt < RTT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = 16 * nt + 4 * G + r;
            if (n < N) {
                dqkv[(b * N + n) * (3 * RE) + RE + 16 * h + j] = dK[nt][r];
                dqkv[(b * N + n) * (3 * RE) + 2 * RE + 16 * h + j] = dV[nt][r];
            }
        }
}

template <int RTT>
static int launch_mha_bwd_t(const float* qkv, const float* dout, float* dqkv, int64_t B, int N, hipStream_t st)
{
    const size_t ldsz = (4 * 16 * (size_t)TS + 8 * 4 * 256 + 8 * RTT * 256) * sizeof(float);
    auto k = k_mha_encoder_bwd<RTT>;
    if (ldsz > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz) != hipSuccess)
        return EAMRL_E_LAUNCH;
    hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(512), ldsz, st, qkv, dout, dqkv, N);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

bool mha_encoder_bwd_supports(int N, int E, int H) { return N >= 1 && N <= 112 && E == RE && H == RH; }

int launch_mha_encoder_bwd(const float* qkv, const float* dout, float* dqkv, int64_t B, int N, hipStream_t st)
{
    if (B <= 0) return 0;
    if (N <= 32) return launch_mha_bwd_t<2>(qkv, dout, dqkv, B, N, st);
    if (N <= 64) return launch_mha_bwd_t<4>(qkv, dout, dqkv, B, N, st);
    return launch_mha_bwd_t<7>(qkv, dout, dqkv, B, N, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// feasibility masks as bit sets: bits[(r * T + t) * 4 + (n >> 5)] |= mask[r][n] << (n & 31)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void k_pack_mask_bits(const uint8_t* __restrict__ mask, uint32_t* __restrict__ bits, int64_t R, int M, int T, int t)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one 32-bit word per thread
    if (idx >= R * 4) return;
    const int64_t r = idx >> 2;
    const int w = (int)(idx & 3);
    uint32_t v = 0;
    for (int i = 0; i < 32; ++i) {
        const int n = 32 * w + i;
        if (n < M && mask[r * M + n]) v |= 1u << i;
    }
    bits[(r * T + t) * 4 + w] = v;
}

int launch_pack_mask_bits(const uint8_t* mask, uint32_t* bits, int64_t R, int M, int T, int t, hipStream_t st)
{
    if (R <= 0) return 0;
    hipLaunchKernelGGL(k_pack_mask_bits, dim3((unsigned)((R * 4 + 255) / 256)), dim3(256), 0, st, mask, bits, R, M, T, t);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

// TSP: the whole [R][T] mask-bit array from the action rows (node n is feasible at step t iff it is not among a_0 .. a_{t-1})
__global__ void k_tsp_mask_bits(const int64_t* __restrict__ actions, uint32_t* __restrict__ bits, int64_t R, int M, int T)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    uint32_t m[4];
    for (int w = 0; w < 4; ++w) {
        const int left = M - 32 * w;
        m[w] = left >= 32 ? 0xffffffffu : left > 0 ? ((1u << left) - 1u) : 0u;
    }
    for (int t = 0; t < T; ++t) {
        *reinterpret_cast<uint4*>(bits + (r * T + t) * 4) = make_uint4(m[0], m[1], m[2], m[3]);
        const int a = (int)actions[r * T + t];
        if (a >= 0 && a < M) m[a >> 5] &= ~(1u << (a & 31));
    }
}

// the same two in the layout of the key-chunked kernels: bits [R][T][nkc][4], bit i of chunk c = node 112 c + i
__global__ void k_pack_mask_bits_chunked(const uint8_t* __restrict__ mask, uint32_t* __restrict__ bits, int64_t R, int M, int T, int t,
                                         int nkc)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one 32-bit word per thread
    if (idx >= R * nkc * 4) return;
    const int64_t r = idx / (nkc * 4);
    const int cw = (int)(idx - r * nkc * 4), c = cw >> 2, w = cw & 3;
    uint32_t v = 0;
    for (int i = 0; i < 32; ++i) {
        const int li = 32 * w + i, n = KCH * c + li;
        if (li < KCH && n < M && mask[r * M + n]) v |= 1u << i;
    }
    bits[((r * T + t) * nkc + c) * 4 + w] = v;
}
__global__ void k_tsp_mask_bits_chunked(const int64_t* __restrict__ actions, uint32_t* __restrict__ bits, int64_t R, int M, int T, int nkc)
{
    // one thread per (row, chunk): the chunk's 112 bits through the steps
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * nkc) return;
    const int64_t r = idx / nkc;
    const int c = (int)(idx - r * nkc);
    const int left = min(KCH, M - KCH * c);
    uint32_t m[4];
    for (int w = 0; w < 4; ++w) {
        const int l = left - 32 * w;
        m[w] = l >= 32 ? 0xffffffffu : l > 0 ? ((1u << l) - 1u) : 0u;
    }
    for (int t = 0; t < T; ++t) {
        *reinterpret_cast<uint4*>(bits + ((r * T + t) * nkc + c) * 4) = make_uint4(m[0], m[1], m[2], m[3]);
        const int a = (int)actions[r * T + t] - KCH * c;
        if (a >= 0 && a < left) m[a >> 5] &= ~(1u << (a & 31));
    }
}
int launch_pack_mask_bits_chunked(const uint8_t* mask, uint32_t* bits, int64_t R, int M, int T, int t, hipStream_t st)
{
    if (R <= 0) return 0;
    const int nkc = (M + KCH - 1) / KCH;
    hipLaunchKernelGGL(k_pack_mask_bits_chunked, dim3((unsigned)((R * nkc * 4 + 255) / 256)), dim3(256), 0, st, mask, bits, R, M, T, t, nkc);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}
int launch_tsp_mask_bits_chunked(const int64_t* actions, uint32_t* bits, int64_t R, int M, int T, hipStream_t st)
{
    if (R <= 0) return 0;
    const int nkc = (M + KCH - 1) / KCH;
    hipLaunchKernelGGL(k_tsp_mask_bits_chunked, dim3((unsigned)((R * nkc + 127) / 128)), dim3(128), 0, st, actions, bits, R, M, T, nkc);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_tsp_mask_bits(const int64_t* actions, uint32_t* bits, int64_t R, int M, int T, hipStream_t st)
{
    if (R <= 0) return 0;
    hipLaunchKernelGGL(k_tsp_mask_bits, dim3((unsigned)((R + 127) / 128)), dim3(128), 0, st, actions, bits, R, M, T);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

}  // namespace eamrl
