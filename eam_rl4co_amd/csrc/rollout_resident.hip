// Register-resident whole-rollout kernel (M <= 128 nodes, E = 128, H = 8): the MI355X-native decode loop.
//
// One 256-thread workgroup owns one rollout row for its whole episode.  The row's glimpse keys, glimpse
// values and folded logit keys (3 * M * E fp32 = 153.6 KB at M = 100) are read from HBM ONCE and then live
// in the workgroup's vector registers (a CU has 512 KB of VGPRs, more than its 160 KB of LDS); the
// current-node context rows live in LDS.  Two workgroups fit a CU (<= 256 VGPRs, ~60 KB LDS each), so
// 512 rows are in flight on the chip and a 1024-row batch takes two waves of workgroups.  Per step nothing
// but the selected action and its log-prob (12 B) goes to HBM, and the Exp(1) noise row when sampling.
//
// Thread layouts (tid in [0,256), lane = tid & 63):
//   scores / logits : thread (n = tid & 127, hs = tid >> 7) keeps K[n][64hs..64hs+63] and Lp[n][64hs..64hs+63]
//   glimpse (P*V)   : thread (e = tid & 127, g = tid >> 7) keeps V[n][e] for the nodes of chunks 2g and 2g+1
//   final softmax   : wavefront 0, lane l handles nodes l and l + 64 (all reductions are DPP butterflies)
// The arithmetic follows the canonical order, so tours, log-probs and rewards are bit-identical
// to k_rollout_stream, k_decode_step and the CPU oracle.
//
// Reference loop replaced: rl4co/models/common/constructive/base.py:236-250 with
// rl4co/models/zoo/am/decoder.py:161-198, rl4co/models/nn/attention.py:282-328,
// rl4co/utils/decoding.py:140-190,346-465 and rl4co/envs/routing/{tsp,cvrp}/env.py step functions.
#include "kernels.hpp"

namespace eamrl {

namespace {

constexpr int RB = 256;      // threads
constexpr int RE = 128;      // embed dim
constexpr int RH = 8;        // heads
constexpr int RD = 16;       // head dim
constexpr int RNP = 128;     // node slots

template <int CP>
struct ResLds {
    static constexpr int WROW = 4 * CP + 4;     // floats per head row of w (chunk-padded, +4 spreads banks)
    float q[RE];
    float heads[RE];
    float w[RH * WROW];
    float partA[EAMRL_NCHUNK * RE];
    float partZ[EAMRL_NCHUNK * RH];
    float cpart[RNP * 4];
    float redmax[4 * 4];
    float dem[RNP];
    int sel;
    float sel_lp;
    int flags;
    uint8_t msk[RNP];
    uint8_t vis[RNP];
    // followed by P[M][RE] (context rows of the current node: TSP Pb, CVRP Pa)
};

template <int ENV, int CP>
__global__ __launch_bounds__(RB, 2) void k_rollout_resident(DecArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using L = ResLds<CP>;
    L& l = *reinterpret_cast<L*>(smem);
    float* Plds = reinterpret_cast<float*>(smem + ((sizeof(L) + 15) & ~size_t(15)));
    constexpr int WROW = L::WROW;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = tid & 127, hs = tid >> 7;   // (node, column half) for K / Lp; also (e = n, g = hs) for V
    const int M = a.M;
    const int64_t r = blockIdx.x;
    const int64_t bi = r % a.B;
    const int64_t ld = a.ld;
    const int C = (M + EAMRL_NCHUNK - 1) / EAMRL_NCHUNK;

    // ---- one-time loads: K, Lp rows (node-major), V columns (chunked), context rows -> LDS -----------------
    float kreg[64], lreg[64], vreg[2][CP];
    {
        const bool valid = n < M;
        const float* kp = a.K + (bi * M + (valid ? n : 0)) * ld + 64 * hs;
        const float* lp = a.Lp + (bi * M + (valid ? n : 0)) * ld + 64 * hs;
#pragma unroll
        for (int i = 0; i < 64; i += 4) {
            float4 kk = valid ? *reinterpret_cast<const float4*>(kp + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 ll = valid ? *reinterpret_cast<const float4*>(lp + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            kreg[i] = kk.x; kreg[i + 1] = kk.y; kreg[i + 2] = kk.z; kreg[i + 3] = kk.w;
            lreg[i] = ll.x; lreg[i + 1] = ll.y; lreg[i + 2] = ll.z; lreg[i + 3] = ll.w;
        }
        const float* vp = a.V + bi * M * ld + n;   // column e = n
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int n0 = (2 * hs + c) * C;
#pragma unroll
            for (int i = 0; i < CP; ++i) {
                const int nn = n0 + i;
                vreg[c][i] = (i < C && nn < M) ? vp[(int64_t)nn * ld] : 0.0f;
            }
        }
        const float* Pc = (ENV == EAMRL_ENV_TSP ? a.Pb : a.Pa) + bi * M * ld;
        for (int i = tid; i < M * (RE / 4); i += RB) {
            const int row = i / (RE / 4), c4 = i - row * (RE / 4);
            *reinterpret_cast<float4*>(Plds + row * RE + 4 * c4) =
                *reinterpret_cast<const float4*>(Pc + (int64_t)row * ld + 4 * c4);
        }
        for (int i = tid; i < RH * WROW; i += RB) l.w[i] = 0.0f;   // chunk padding stays 0 for the whole episode
    }
    // per-thread constants
    const int my_pos = (n < M) ? (n / C) * CP + (n - (n / C) * C) : 0;   // slot of node n in a w row
    const float gq = (a.gctx && tid < RE) ? a.gctx[bi * RE + tid] : 0.0f;
    const float cv = (tid < RE) ? a.cvec[tid] : 0.0f;
    const float sqrtE = __builtin_sqrtf((float)RE);

    // ---- row state (uniform) ---------------------------------------------------------------------------
    int64_t first = 0, cur = a.cur[r], istep = 1;
    float used = 0.0f, vcap = 0.0f;
    if (ENV == EAMRL_ENV_TSP) { first = a.first[r]; istep = a.istep[r]; }
    else { used = a.used[r]; vcap = a.vcap[r]; }
    bool done = a.done[r] != 0;
    if (tid < RNP) {
        l.msk[tid] = (tid < M) ? a.mask[r * M + tid] : 0;
        l.vis[tid] = (ENV == EAMRL_ENV_CVRP && tid < M) ? a.visited[r * M + tid] : 0;
    }
    __syncthreads();
    // remaining feasible (TSP) / visited (CVRP) node count, kept incrementally (== the reference's mask.sum / visited.sum)
    int count = 0;
    for (int i = 0; i < M; ++i) count += (ENV == EAMRL_ENV_TSP) ? (l.msk[i] != 0) : (l.vis[i] != 0);
    float p1f = 0.0f;   // TSP: P_first[first][e] once the first node is known
    if (ENV == EAMRL_ENV_TSP && istep > 0 && tid < RE) p1f = a.Pa[(bi * M + first) * ld + tid];
    const float my_dem = (ENV == EAMRL_ENV_CVRP && n >= 1 && n < M) ? a.demand[bi * (M - 1) + n - 1] : 0.0f;
    if (ENV == EAMRL_ENV_CVRP && hs == 0 && n >= 1) l.dem[n - 1] = my_dem;   // demand row staged in LDS
    const float* dem = l.dem;

    // query for the first step
    if (tid < RE) {
        float ctx;
        if (ENV == EAMRL_ENV_TSP) ctx = (istep == 0) ? cv : p1f + Plds[cur * RE + tid];
        else ctx = fma_(cv, vcap - used, Plds[cur * RE + tid]);
        l.q[tid] = ctx + gq;
    }
    __syncthreads();

    int t = 0;
    uint32_t st_flags = 0;
    while (!done && t < a.t_max) {
        // prefetch this step's per-row inputs (latency hidden behind the glimpse)
        float nz0 = 1.0f, nz1 = 1.0f;
        if (a.mode == EAMRL_SAMPLE && wv == 0) {
            const float* nzp = a.noise + (r * a.t_max + t) * (int64_t)M;
            if (lane < M) nz0 = nzp[lane];
            if (lane + 64 < M) nz1 = nzp[lane + 64];
        }
        int64_t given = 0;
        if (a.mode == EAMRL_EVALUATE) given = (t < a.t_given) ? a.given[r * a.t_given + t] : 0;

        // ---- S1: scores of 4 heads for node n, per-head max ------------------------------------------------
        const bool feas = (n < M) && l.msk[n] != 0;
        float sc[4];
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
            const float* qp = l.q + (4 * hs + hh) * RD;
            float acc = 0.0f;
#pragma unroll
            for (int d = 0; d < RD; d += 4) {
                const float4 qq = *reinterpret_cast<const float4*>(qp + d);
                acc = fma_(qq.x, kreg[hh * RD + d], acc);
                acc = fma_(qq.y, kreg[hh * RD + d + 1], acc);
                acc = fma_(qq.z, kreg[hh * RD + d + 2], acc);
                acc = fma_(qq.w, kreg[hh * RD + d + 3], acc);
            }
            sc[hh] = feas ? acc * 0.25f : -INFINITY;      // 1/sqrt(16)
            const float m = wave_max(sc[hh]);
            if (lane == 0) l.redmax[wv * 4 + hh] = m;
        }
        __syncthreads();
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
            const float m = __builtin_fmaxf(l.redmax[(2 * hs) * 4 + hh], l.redmax[(2 * hs + 1) * 4 + hh]);
            if (n < M) l.w[(4 * hs + hh) * WROW + my_pos] = feas ? d_expf(sc[hh] - m) : 0.0f;
        }
        __syncthreads();

        // ---- S2: glimpse partials, column e = n, chunks 2hs and 2hs+1 -------------------------------------------
        {
            const int h = n >> 4;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float* wp = l.w + h * WROW + (2 * hs + c) * CP;
                float zg = 0.0f, ag = 0.0f;
#pragma unroll
                for (int i = 0; i < CP; i += 4) {
                    const float4 ww = *reinterpret_cast<const float4*>(wp + i);
                    zg = zg + ww.x; ag = fma_(ww.x, vreg[c][i], ag);
                    zg = zg + ww.y; ag = fma_(ww.y, vreg[c][i + 1], ag);
                    zg = zg + ww.z; ag = fma_(ww.z, vreg[c][i + 2], ag);
                    zg = zg + ww.w; ag = fma_(ww.w, vreg[c][i + 3], ag);
                }
                l.partA[(2 * hs + c) * RE + n] = ag;
                if ((n & 15) == 0) l.partZ[(2 * hs + c) * RH + h] = zg;
            }
        }
        __syncthreads();
        if (tid < RE) {
            const int h = tid >> 4;
            float A = l.partA[tid], Z = l.partZ[h];
#pragma unroll
            for (int g = 1; g < EAMRL_NCHUNK; ++g) { A = A + l.partA[g * RE + tid]; Z = Z + l.partZ[g * RH + h]; }
            l.heads[tid] = A / Z;
        }
        __syncthreads();

        // ---- S4: logit partials of column chunks 2hs, 2hs+1 for node n -------------------------------------------
        {
            float cp2[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float* hp = l.heads + (2 * hs + c) * 32;
                float cg = 0.0f;
#pragma unroll
                for (int e = 0; e < 32; e += 4) {
                    const float4 hh4 = *reinterpret_cast<const float4*>(hp + e);
                    cg = fma_(hh4.x, lreg[c * 32 + e], cg);
                    cg = fma_(hh4.y, lreg[c * 32 + e + 1], cg);
                    cg = fma_(hh4.z, lreg[c * 32 + e + 2], cg);
                    cg = fma_(hh4.w, lreg[c * 32 + e + 3], cg);
                }
                cp2[c] = cg;
            }
            *reinterpret_cast<float2*>(l.cpart + n * 4 + 2 * hs) = make_float2(cp2[0], cp2[1]);
        }
        __syncthreads();

        // ---- S5: wavefront 0 finishes the step: clip, mask, log-softmax, selection -----------------------------------
        if (wv == 0) {
            float x[2], lpv[2];
            bool fe[2];
            bool nan_seen = false;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int nn = lane + 64 * k;
                fe[k] = (nn < M) && l.msk[nn] != 0;
                const float4 cp4 = *reinterpret_cast<const float4*>(l.cpart + nn * 4);
                const float u = ((cp4.x + cp4.y) + cp4.z) + cp4.w;
                const float logit = u / sqrtE;
                if (fe[k] && logit != logit) nan_seen = true;
                float v = (a.clip > 0.0f) ? d_tanhf(logit) * a.clip : logit;
                if (!fe[k]) v = -INFINITY;
                x[k] = v / a.temp;
            }
            const float mx = wave_max(__builtin_fmaxf(x[0], x[1]));
            const float e0 = fe[0] ? d_expf(x[0] - mx) : 0.0f;
            const float e1 = fe[1] ? d_expf(x[1] - mx) : 0.0f;
            float Zl = wave_tree_sum(e0);
            if (M > 64) Zl = Zl + wave_tree_sum(e1);
            const float lse = d_logf(Zl);
            lpv[0] = fe[0] ? (x[0] - mx) - lse : -INFINITY;
            lpv[1] = fe[1] ? (x[1] - mx) - lse : -INFINITY;
            float key0 = lpv[0], key1 = lpv[1];
            if (a.mode == EAMRL_SAMPLE) { key0 = d_expf(lpv[0]) / nz0; key1 = d_expf(lpv[1]) / nz1; }
            float best = (lane < M) ? key0 : -INFINITY;
            int besti = (lane < M) ? lane : 0x7fffffff;
            if (lane + 64 < M && key1 > best) { best = key1; besti = lane + 64; }
            wave_argmax(best, besti);
            int sel = besti;
            if (a.mode == EAMRL_EVALUATE) sel = (int)given;
            sel = __builtin_amdgcn_readfirstlane(sel);
            uint32_t fl = 0;
            if (__ballot(nan_seen) != 0ull) fl |= EAMRL_ST_NAN_LOGITS;
            if (sel < 0 || sel >= M) { fl |= EAMRL_ST_INFEASIBLE; sel = 0; }
            else if (!l.msk[sel]) fl |= EAMRL_ST_INFEASIBLE;
            const float lp_sel = __int_as_float(
                (sel < 64) ? __builtin_amdgcn_readlane(__float_as_int(lpv[0]), sel)
                           : __builtin_amdgcn_readlane(__float_as_int(lpv[1]), sel - 64));
            if (lane == 0) {
                l.sel = sel;
                l.flags = (int)fl;
                a.action[r * a.t_max + t] = sel;
                a.logp[r * a.t_max + t] = lp_sel;
            }
        }
        __syncthreads();

        // ---- S6: env transition + next query -----------------------------------------------------------------------------
        const int act = l.sel;
        st_flags |= (uint32_t)l.flags;
        if (ENV == EAMRL_ENV_TSP) {
            if (istep == 0) {
                first = act;
                if (tid < RE) p1f = a.Pa[(bi * M + first) * ld + tid];
            }
            cur = act;
            istep += 1;
            count -= (l.msk[act] != 0);
            done = (count == 0);
            __syncthreads();                       // everyone has read msk[act]
            if (tid == 0) l.msk[act] = 0;
            if (tid < RE) l.q[tid] = (p1f + Plds[cur * RE + tid]) + gq;
            __syncthreads();
        } else {
            const int N = M - 1;
            int di = act - 1;
            di = di < 0 ? 0 : (di > N - 1 ? N - 1 : di);
            used = (used + dem[di]) * (act != 0 ? 1.0f : 0.0f);
            cur = act;
            count += (l.vis[act] == 0);
            done = (count == M);
            __syncthreads();                       // everyone has read vis[act]
            if (tid == 0) l.vis[act] = 1;
            if (tid < RE) l.q[tid] = fma_(cv, vcap - used, Plds[cur * RE + tid]) + gq;
            __syncthreads();
            const float lim = vcap + 1e-5f;
            int free_n = 0;
            if (hs == 0 && n >= 1 && n < M) {
                const int blocked = (l.vis[n] != 0) | ((my_dem + used) > lim);
                l.msk[n] = !blocked;
                free_n = !blocked;
            }
            const int any_free = __syncthreads_or(free_n);
            if (tid == 0) l.msk[0] = !((cur == 0) && any_free);
            __syncthreads();
        }
        ++t;
    }

    // ---- write back the final state -----------------------------------------------------------------------------------
    __syncthreads();
    if (tid < M) {
        a.mask[r * M + tid] = l.msk[tid];
        if (ENV == EAMRL_ENV_CVRP) a.visited[r * M + tid] = l.vis[tid];
    }
    if (tid == 0) {
        a.cur[r] = cur;
        a.done[r] = done ? 1 : 0;
        if (ENV == EAMRL_ENV_TSP) { a.first[r] = first; a.istep[r] = istep; }
        else a.used[r] = used;
        atomicMax(a.steps_out, t);
        if (!done) st_flags |= EAMRL_ST_STEP_OVERRUN;
        if (st_flags) atomicOr(a.status, st_flags);
    }
}

__global__ void k_rollout_pad_cvrp_res(DecArgs a)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.R) return;
    const int T = *a.steps_out;
    if (T <= 0 || !a.done[r]) return;
    if (a.cur[r] != 0 && a.action[r * a.t_max + (T - 1)] == 0) {   // see k_rollout_pad_cvrp
        a.cur[r] = 0;
        a.used[r] = 0.0f;
    }
}

template <int ENV, int CP>
int launch_cp(const DecArgs& a, hipStream_t st)
{
    const size_t lds = ((sizeof(ResLds<CP>) + 15) & ~size_t(15)) + (size_t)a.M * RE * sizeof(float);
    auto k = k_rollout_resident<ENV, CP>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return EAMRL_E_LAUNCH;
    hipLaunchKernelGGL(k, dim3((unsigned)a.R), dim3(RB), lds, st, a);
    if (ENV == EAMRL_ENV_CVRP)
        hipLaunchKernelGGL(k_rollout_pad_cvrp_res, dim3((unsigned)((a.R + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

template <int ENV>
int launch_env(const DecArgs& a, hipStream_t st)
{
    const int C = (a.M + EAMRL_NCHUNK - 1) / EAMRL_NCHUNK;
    if (C <= 8) return launch_cp<ENV, 8>(a, st);
    if (C <= 16) return launch_cp<ENV, 16>(a, st);
    if (C <= 28) return launch_cp<ENV, 28>(a, st);
    return launch_cp<ENV, 32>(a, st);
}

}  // namespace

bool rollout_resident_supports(int env, const DecArgs& a)
{
    return a.E == RE && a.H == RH && a.M <= RNP && a.M >= 2 && a.ld % 4 == 0;
}

int launch_rollout_resident(int env, const DecArgs& a, hipStream_t st)
{
    return env == EAMRL_ENV_TSP ? launch_env<EAMRL_ENV_TSP>(a, st) : launch_env<EAMRL_ENV_CVRP>(a, st);
}

}  // namespace eamrl
