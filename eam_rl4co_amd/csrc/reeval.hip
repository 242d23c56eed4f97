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
template <int RTT>
__device__ __forceinline__ float head_softmax(const float (&kf)[RTT][4], const float* QT, int h, int lane, const uint4& mb, int M,
                                              f32x4 (&s)[RTT])
{
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
    m = group_max(m);
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
    return z > 0.0f ? 1.0f / z : 0.0f;
}

// heads tile of all heads -> HT (A layout): wave h computes head h
template <int RTT>
__device__ __forceinline__ void glimpse_tile(const float (&kf)[RTT][4], const float (&vtf)[4 * RTT], const float* QT, float* HT,
                                             int h, int lane, const uint4& mb, int M)
{
    const int j = lane & 15, G = lane >> 4;
    f32x4 s[RTT];
    const float iz = head_softmax<RTT>(kf, QT, h, lane, mb, M, s);
    f32x4 o = z4();
#pragma unroll
    for (int t = 0; t < 4 * RTT; ++t) o = mf(vtf[t], s[t >> 2][t & 3], o);
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
template <int RTT>
__global__ __launch_bounds__(512, 2) void k_reeval_fwd(ReevalArgs a)
{
    __shared__ __attribute__((aligned(16))) float QT[16 * TS];
    __shared__ __attribute__((aligned(16))) float HT[16 * TS];
    __shared__ float RED[8][16], RED2[8][16], RED3[8][16], ZA[16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, G = lane >> 4;
    const int64_t b = blockIdx.x / a.nchunk;
    const int ch = (int)(blockIdx.x - b * a.nchunk);
    const int s0 = (int)((int64_t)a.S * ch / a.nchunk), s1 = (int)((int64_t)a.S * (ch + 1) / a.nchunk);
    const int64_t nq = (int64_t)(s1 - s0) * a.T;
    const int64_t ntiles = (nq + 15) / 16;

    float kf[RTT][4], vtf[4 * RTT], lpf[32];
    load_head_frags<RTT>(a, b, wv, lane, kf, vtf);
    if (wv < RTT) load_lp_frags(a, b, wv, lane, lpf);
    const float inv_temp = 1.0f / a.temp;

    for (int64_t tile = 0; tile < ntiles; ++tile) {
        build_query_tile(a, b, s0, nq, tile, QT);
        const Q q = tile_query(a, b, s0, nq, tile, j);
        uint4 mb = make_uint4(0, 0, 0, 0);
        int act = -1;
        if (q.qi >= 0) {
            mb = *reinterpret_cast<const uint4*>(a.maskbits + q.qi * 4);
            act = (int)a.actions[q.qi];
        }
        __syncthreads();
        glimpse_tile<RTT>(kf, vtf, QT, HT, wv, lane, mb, a.M);
        __syncthreads();
        f32x4 z = z4();
        float mx = -INFINITY;
        if (wv < RTT) {
            const f32x4 u = logit_tile(lpf, HT, lane);
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

template <int RTT>
static int launch_fwd_t(const ReevalArgs& a, hipStream_t st)
{
    hipLaunchKernelGGL(k_reeval_fwd<RTT>, dim3((unsigned)(a.B * a.nchunk)), dim3(512), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

bool reeval_supports(int M, int E, int H) { return M >= 1 && M <= 112 && E == RE && H == RH; }

int launch_reeval_fwd(const ReevalArgs& a, hipStream_t st)
{
    if (a.B <= 0 || a.S <= 0 || a.T <= 0) return 0;
    if (a.M <= 32) return launch_fwd_t<2>(a, st);
    if (a.M <= 64) return launch_fwd_t<4>(a, st);
    return launch_fwd_t<7>(a, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward 1/2: logits.  Recomputes heads and u per tile; du -> dheads (scratch in HBM, read by the glimpse kernel) and
// dLp[n][e] += sum_q du[q][n] heads[q][e] (accumulators of wave w: embedding columns 16 w .. 16 w + 15, all key tiles)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int DS = 17;        // row stride of the du / transposition staging tiles ([n][q])

template <int RTT>
__global__ __launch_bounds__(512, 2) void k_reeval_bwd_logits(ReevalArgs a)
{
    __shared__ __attribute__((aligned(16))) float QT[16 * TS];
    __shared__ __attribute__((aligned(16))) float HT[16 * TS];
    __shared__ float DU[16 * RTT * DS];
    __shared__ float LSE[16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, G = lane >> 4;
    const int64_t b = blockIdx.x / a.nchunk;
    const int ch = (int)(blockIdx.x - b * a.nchunk);
    const int s0 = (int)((int64_t)a.S * ch / a.nchunk), s1 = (int)((int64_t)a.S * (ch + 1) / a.nchunk);
    const int64_t nq = (int64_t)(s1 - s0) * a.T;
    const int64_t ntiles = (nq + 15) / 16;
    // lse == NULL: no forward pass was run -- logp holds the ROLLOUT's log-prob of the chosen node (the same quantity in the
    // rollout kernels' arithmetic), and the normaliser is recovered as z[action] - logp, one extra LDS hand-off per tile
    const bool derive_lse = a.lse == nullptr;

    float kf[RTT][4], vtf[4 * RTT], lpf[32], lptf[4 * RTT];
    load_head_frags<RTT>(a, b, wv, lane, kf, vtf);
    if (wv < RTT) load_lp_frags(a, b, wv, lane, lpf);
#pragma unroll
    for (int t = 0; t < 4 * RTT; ++t) {                 // Lp^T: row e = 16 wv + j, k index = key 4 t + G
        const int n = 4 * t + G;
        lptf[t] = n < a.M ? a.Lp[(b * a.M + n) * a.ld + 16 * wv + j] : 0.0f;
    }
    f32x4 dLp[RTT];
#pragma unroll
    for (int nt = 0; nt < RTT; ++nt) dLp[nt] = z4();
    const float inv_temp = 1.0f / a.temp;

    for (int64_t tile = 0; tile < ntiles; ++tile) {
        build_query_tile(a, b, s0, nq, tile, QT);
        const Q q = tile_query(a, b, s0, nq, tile, j);
        uint4 mb = make_uint4(0, 0, 0, 0);
        int act = -1;
        float g = 0.0f, lse = 0.0f;
        if (q.qi >= 0) {
            mb = *reinterpret_cast<const uint4*>(a.maskbits + q.qi * 4);
            act = (int)a.actions[q.qi];
            if (q.active) { g = a.glogp[q.qi]; lse = derive_lse ? a.logp[q.qi] : a.lse[q.qi]; }
        }
        __syncthreads();
        glimpse_tile<RTT>(kf, vtf, QT, HT, wv, lane, mb, a.M);
        __syncthreads();
        f32x4 u = z4();
        if (wv < RTT) u = logit_tile(lpf, HT, lane);
        if (derive_lse) {           // the lane that owns the chosen node publishes z[action] - logp for its query
            if (wv < RTT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float dzdu;
                    const float z = process_logit(u[r], a.clip, inv_temp, dzdu);
                    if (16 * wv + 4 * r + G == act && g != 0.0f) LSE[j] = z - lse;
                }
            }
            __syncthreads();
            if (g != 0.0f) lse = LSE[j];
        }
        if (wv < RTT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n0 = 16 * wv + 4 * r;
                const uint32_t w = (n0 >> 5) == 0 ? mb.x : (n0 >> 5) == 1 ? mb.y : (n0 >> 5) == 2 ? mb.z : mb.w;
                const bool ok = (w >> ((n0 & 31) + G)) & 1u;
                float dzdu;
                const float z = process_logit(u[r], a.clip, inv_temp, dzdu);
                float du = 0.0f;
                if (ok && g != 0.0f) {
                    const float p = fexp(z - lse);
                    du = g * ((n0 + G == act ? 1.0f : 0.0f) - p) * dzdu;
                }
                DU[(n0 + G) * DS + j] = du;
            }
        }
        __syncthreads();
        {   // wave wv: embedding columns 16 wv .. 16 wv + 15
            f32x4 dh = z4();
#pragma unroll
            for (int t = 0; t < 4 * RTT; ++t) dh = mf(lptf[t], DU[(4 * t + G) * DS + j], dh);
            if (q.qi >= 0)
                *reinterpret_cast<float4*>(a.dheads + q.qi * RE + 16 * wv + 4 * G) = make_float4(dh[0], dh[1], dh[2], dh[3]);
            float hb[4];
            const int c = 16 * wv + j;
#pragma unroll
            for (int t = 0; t < 4; ++t) hb[t] = HT[(4 * t + G) * TS + (c & 3) * TG + (c >> 2)];
#pragma unroll
            for (int nt = 0; nt < RTT; ++nt)
#pragma unroll
                for (int t = 0; t < 4; ++t) dLp[nt] = mf(DU[(16 * nt + j) * DS + 4 * t + G], hb[t], dLp[nt]);
        }
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

// ---------------------------------------------------------------------------------------------------------------------
// backward 2/2: glimpse.  Wave h = head h: recomputes the softmax, da = V dheads, ds = a (da - sum a da),
// dq~ = K^T ds -> scattered (LDS atomics) into the per-instance dPa / dPb / dgctx / dCvec accumulators,
// dV[n][e] += sum_q a[q][n] dheads[q][e], dK[n][d] += sum_q ds[q][n] q~[q][d]
// ---------------------------------------------------------------------------------------------------------------------
constexpr int DQS = 132;      // row stride of the dq tile

template <int RTT>
__global__ __launch_bounds__(512, 2) void k_reeval_bwd_glimpse(ReevalArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* QT = lds;                         // [16][TS]
    float* DHT = QT + 16 * TS;               // [16][TS]   dheads tile (A layout)
    float* DQ = DHT + 16 * TS;               // [16][DQS]  dq tile
    float* ST = DQ + 16 * DQS;               // [8 waves][16][DS] transposition staging
    float* ACC_A = ST + 8 * 16 * DS;         // [M][128] dPa
    float* ACC_B = ACC_A + a.M * RE;         // [M][128] dPb
    float* ACC_G = ACC_B + a.M * RE;         // [128] dgctx
    float* ACC_C = ACC_G + RE;               // [4][128] dCvec
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, G = lane >> 4, pi = 4 * (j & 3) + (j >> 2);
    const int64_t b = blockIdx.x / a.nchunk;
    const int ch = (int)(blockIdx.x - b * a.nchunk);
    const int s0 = (int)((int64_t)a.S * ch / a.nchunk), s1 = (int)((int64_t)a.S * (ch + 1) / a.nchunk);
    const int64_t nq = (int64_t)(s1 - s0) * a.T;
    const int64_t ntiles = (nq + 15) / 16;
    const int h = wv;
    float* st = ST + wv * 16 * DS;

    for (int i = tid; i < 2 * a.M * RE + 5 * RE; i += blockDim.x) ACC_A[i] = 0.0f;

    float kf[RTT][4], vaf[RTT][4], ktf[4 * RTT];
#pragma unroll
    for (int kt = 0; kt < RTT; ++kt) {
        const int n = 16 * kt + pi;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            kf[kt][t] = n < a.M ? a.K[(b * a.M + n) * a.ld + 16 * h + 4 * t + G] : 0.0f;
            vaf[kt][t] = n < a.M ? a.V[(b * a.M + n) * a.ld + 16 * h + 4 * t + G] : 0.0f;
        }
    }
#pragma unroll
    for (int t = 0; t < 4 * RTT; ++t) {                 // K^T: row d = j, k index = key 4 t + G
        const int n = 4 * t + G;
        ktf[t] = n < a.M ? a.K[(b * a.M + n) * a.ld + 16 * h + j] : 0.0f;
    }
    f32x4 dV[RTT], dK[RTT];
#pragma unroll
    for (int nt = 0; nt < RTT; ++nt) { dV[nt] = z4(); dK[nt] = z4(); }

    for (int64_t tile = 0; tile < ntiles; ++tile) {
        build_query_tile(a, b, s0, nq, tile, QT);
        {   // dheads tile -> DHT (A layout)
            const int jq = tid >> 5, e4 = tid & 31;
            const Q qq = tile_query(a, b, s0, nq, tile, jq);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qq.qi >= 0 && qq.active) v = *reinterpret_cast<const float4*>(a.dheads + qq.qi * RE + 4 * e4);
            float* p = DHT + jq * TS + e4;
            p[0] = v.x; p[TG] = v.y; p[2 * TG] = v.z; p[3 * TG] = v.w;
        }
        const Q q = tile_query(a, b, s0, nq, tile, j);
        uint4 mb = make_uint4(0, 0, 0, 0);
        if (q.qi >= 0) mb = *reinterpret_cast<const uint4*>(a.maskbits + q.qi * 4);
        __syncthreads();
        {
            f32x4 s[RTT], da[RTT];
            const float iz = head_softmax<RTT>(kf, QT, h, lane, mb, a.M, s);
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
            float rs = 0.0f;
#pragma unroll
            for (int kt = 0; kt < RTT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[kt][r] *= iz;                          // a
                    rs = fmaf(s[kt][r], da[kt][r], rs);
                }
            rs = group_sum(rs);
#pragma unroll
            for (int kt = 0; kt < RTT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) da[kt][r] = s[kt][r] * (da[kt][r] - rs);      // ds
            // dq~^T = K^T ds: lane (query j, G), register r -> d = 4 G + r
            f32x4 dq = z4();
#pragma unroll
            for (int t = 0; t < 4 * RTT; ++t) dq = mf(ktf[t], da[t >> 2][t & 3], dq);
            *reinterpret_cast<float4*>(DQ + j * DQS + 16 * h + 4 * G) =
                make_float4(0.25f * dq[0], 0.25f * dq[1], 0.25f * dq[2], 0.25f * dq[3]);
            // B operands of the two "sum over queries" products: dheads_h [q][e] and q~_h [q][d], column j, k index q = 4 t + G
            float dhb[4], qb[4];
            const int c = 16 * h + j;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                dhb[t] = DHT[(4 * t + G) * TS + (c & 3) * TG + (c >> 2)];
                qb[t] = QT[(4 * t + G) * TS + (c & 3) * TG + (c >> 2)];
            }
#pragma unroll
            for (int nt = 0; nt < RTT; ++nt) {
                // a^T tile: accumulator layout (lane = query, register r -> key 4 r + G of the tile) -> rows = keys, k = queries
#pragma unroll
                for (int r = 0; r < 4; ++r) st[(4 * r + G) * DS + j] = s[nt][r];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                float at[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) at[t] = st[j * DS + 4 * t + G];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int r = 0; r < 4; ++r) st[(4 * r + G) * DS + j] = da[nt][r];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                float dt[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) dt[t] = st[j * DS + 4 * t + G];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    dV[nt] = mf(at[t], dhb[t], dV[nt]);
                    dK[nt] = mf(dt[t], qb[t], dK[nt]);
                }
            }
        }
        __syncthreads();
        {   // scatter the tile's dq rows into the instance's accumulators (LDS atomics)
            const int jq = tid >> 5, e4 = tid & 31;
            const Q qq = tile_query(a, b, s0, nq, tile, jq);
            if (qq.qi >= 0 && qq.active) {
                const float4 v = *reinterpret_cast<const float4*>(DQ + jq * DQS + 4 * e4);
                const float d[4] = {v.x, v.y, v.z, v.w};
                const int ia = a.idxA[qq.qi];
                const int ib = a.idxB ? a.idxB[qq.qi] : -1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (ia >= 0) __hip_atomic_fetch_add(ACC_A + ia * RE + 4 * e4 + i, d[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (ib >= 0) __hip_atomic_fetch_add(ACC_B + ib * RE + 4 * e4 + i, d[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(ACC_G + 4 * e4 + i, d[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                for (int k = 0; k < a.NC; ++k) {
                    const float sck = a.sc[(int64_t)k * a.R * a.T + qq.qi];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        __hip_atomic_fetch_add(ACC_C + k * RE + 4 * e4 + i, sck * d[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    // ---- flush --------------------------------------------------------------------------------------------------------------
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
    __syncthreads();
    for (int i = tid; i < a.M * RE; i += blockDim.x) {
        const int n = i / RE, e = i - n * RE;
        const float va = ACC_A[i];
        if (va != 0.0f) atomicAdd(a.dPa + (b * a.M + n) * a.ldg + e, va);
        if (a.dPb) {
            const float vb = ACC_B[i];
            if (vb != 0.0f) atomicAdd(a.dPb + (b * a.M + n) * a.ldg + e, vb);
        }
    }
    if (tid < RE) {
        if (a.dgctx) atomicAdd(a.dgctx + b * RE + tid, ACC_G[tid]);
        for (int k = 0; k < a.NC; ++k) atomicAdd(a.dCvec + k * RE + tid, ACC_C[k * RE + tid]);
    }
}

template <int RTT>
static int launch_bwd_t(const ReevalArgs& a, hipStream_t st)
{
    hipLaunchKernelGGL(k_reeval_bwd_logits<RTT>, dim3((unsigned)(a.B * a.nchunk)), dim3(512), 0, st, a);
    const size_t lds = (2 * 16 * (size_t)TS + 16 * DQS + 8 * 16 * DS + 2 * (size_t)a.M * RE + 5 * RE) * sizeof(float);
    auto k = k_reeval_bwd_glimpse<RTT>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return EAMRL_E_LAUNCH;
    hipLaunchKernelGGL(k, dim3((unsigned)(a.B * a.nchunk)), dim3(512), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_reeval_bwd(const ReevalArgs& a, hipStream_t st)
{
    if (a.B <= 0 || a.S <= 0 || a.T <= 0) return 0;
    if (a.M <= 32) return launch_bwd_t<2>(a, st);
    if (a.M <= 64) return launch_bwd_t<4>(a, st);
    return launch_bwd_t<7>(a, st);
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

int launch_tsp_mask_bits(const int64_t* actions, uint32_t* bits, int64_t R, int M, int T, hipStream_t st)
{
    if (R <= 0) return 0;
    hipLaunchKernelGGL(k_tsp_mask_bits, dim3((unsigned)((R + 127) / 128)), dim3(128), 0, st, actions, bits, R, M, T);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

}  // namespace eamrl
