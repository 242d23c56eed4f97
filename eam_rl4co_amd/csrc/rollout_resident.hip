// Register-resident whole-rollout kernel (M <= 128 nodes, E = 128, H = 8): the MI355X-native decode loop.
//
// One 256-thread workgroup owns one rollout row for its whole episode.  The row's glimpse keys, glimpse
// values and folded logit keys (3 * M * E fp32 = 153.6 KB at M = 100) are read from HBM ONCE and then live
// in the workgroup's vector registers (a CU has 512 KB of VGPRs, more than its 160 KB of LDS); the
// current-node context rows live in LDS.  Two workgroups fit a CU (<= 256 VGPRs, ~60 KB LDS each), so
// 512 rows are in flight on the chip and a 1024-row batch takes two waves of workgroups.  Per step nothing
// but the selected action and its log-prob (12 B) goes to HBM, and the Exp(1) noise row when sampling.
//
// Thread layouts (tid in [0,256), lane = tid & 63, wave w = tid >> 6).  A step is two phases and two barriers (round 3; it was
// four stages and four barriers: the glimpse of a head needs only that head's weights and value columns, so a wavefront that
// owns two heads can carry them from the scores to the logit partials of its own 32 columns without meeting the others):
//   wave w, heads 2w and 2w+1, columns 32w .. 32w+31, no workgroup barrier inside:
//     scores   : lane owns the node pair (lane, lane + 64): K[n][32w .. 32w+31] of both nodes in registers -> per-head max
//                and softmax weights by in-wave DPP butterflies; the weights go to two wave-private LDS rows
//     glimpse  : lane (j = lane & 31, half = lane >> 5) keeps V[n][32w + j] for the nodes of chunks 2 half and 2 half + 1;
//                the two halves' chunk sums meet through v_permlane32_swap
//     logits   : Lp[n][32w .. 32w+31] of the lane's node pair against the wave's own 32 glimpse values (re-read through a
//                wave-private LDS row) -> the column-chunk partial of both nodes
//   finish   : wavefront 0: partial sums, clip, mask, log-softmax, selection, env transition, next context query
// The arithmetic follows the canonical order, so tours, log-probs and rewards are bit-identical
// to k_rollout_stream, k_decode_step and the CPU oracle.
//
// Reference loop replaced: rl4co/models/common/constructive/base.py:236-250 with
// rl4co/models/zoo/am/decoder.py:161-198, rl4co/models/nn/attention.py:282-328,
// rl4co/utils/decoding.py:140-190,346-465 and rl4co/envs/routing/{tsp,cvrp,sdvrp}/env.py step functions.
#include "kernels.hpp"

namespace eamrl {

namespace {

constexpr int RB = 256;      // threads
constexpr int RE = 128;      // embed dim
constexpr int RH = 8;        // heads
constexpr int RD = 16;       // head dim
constexpr int RNP = 128;     // node slots

#ifdef EAMRL_STAMPS   // development build only (tools/stamps.sh): per-stage cycle sums of wavefront 0
__device__ unsigned long long g_stamps[8];
#define STAMP(i) do { if (wv == fw) { const unsigned long long now_ = __builtin_readcyclecounter(); \
                                     stamp_acc[i] += now_ - stamp_t; stamp_t = now_; } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// TM: episode length bound (2M+1 steps for TSP / CVRP, 3M+1 for SDVRP); SDF: floats of the SDVRP-only arrays;
// XYF: node slots of the coordinate array (OP, CVRPTW); TWF: node slots of the CVRPTW time-window arrays
template <int CP, int TM, int SDF, int XYF, int TWF>
struct ResLds {
    static constexpr int WROW = 4 * CP + 4;     // floats per head row of w (chunk-padded, +4 spreads banks)
    float q[RE];
    float headsw[RE];                           // 4 wave-private rows of 32 glimpse values
    float w[RH * WROW];
    float cpart[RNP * 4];
    float dem[RNP];                             // CVRP: demand; SDVRP: remaining demand (demand_with_depot)
    float lp_out[TM];                           // selected log-probs of the episode, written out once
    int16_t act_out[TM];                        // selected actions of the episode
    float dynv[3 * SDF];                        // SDVRP: dynamic-embedding vectors wk | wv | lw
    float xy[2 * XYF];                          // OP, CVRPTW: node coordinates
    float twv[3 * TWF];                         // CVRPTW: window start | window end | service time per node
    int done;
    uint8_t msk[RNP];
    uint8_t vis[RNP];
    // followed by P[M][RE] (context rows of the current node: TSP Pb, CVRP Pa)
};

// CP: chunk stride of a w row in LDS (multiple of 4, >= chunk length C); CR: V registers per chunk (C <= CR <= CP)
// MS (multistart batches, R = S*B rows in "(s b)" order): workgroup (b, g) = blockIdx b + g*B keeps instance b's
// operands in registers and rolls out its starts g, g+G, g+2G, ... one after the other.
//
// SDVRP: the dynamic embedding (rem[n] * vector added to row n of K / V / logit key every step) is a rank-1 update and
// is folded (decode_step.hip): one fma per score / glimpse column / logit partial on top of the chains
// over the register-resident, never modified K / V / Lp.
constexpr int res_tmax(int env) { return (env == EAMRL_ENV_SDVRP ? 3 : 2) * RNP + 2; }
constexpr int res_sdf(int env) { return env == EAMRL_ENV_SDVRP ? RE : 4; }
constexpr int res_xyf(int env) { return (env == EAMRL_ENV_OP || env == EAMRL_ENV_CVRPTW) ? RNP : 2; }
constexpr int res_twf(int env) { return env == EAMRL_ENV_CVRPTW ? RNP : 2; }

template <int ENV, int CP, int CR, bool MS>
__global__ __launch_bounds__(RB, 2) void k_rollout_resident(DecArgs a, int S, int G)
{
    constexpr bool SD = ENV == EAMRL_ENV_SDVRP;
    constexpr bool PC = ENV == EAMRL_ENV_PCTSP;     // prize collecting: dem = real_prize, used = collected prize
    constexpr bool OP = ENV == EAMRL_ENV_OP;        // orienteering: dem = arrival limit per node, used = tour length
    constexpr bool TW = ENV == EAMRL_ENV_CVRPTW;    // CVRP + clock: coordinates and windows in LDS
    constexpr bool CV = ENV == EAMRL_ENV_CVRP || TW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using L = ResLds<CP, res_tmax(ENV), res_sdf(ENV), res_xyf(ENV), res_twf(ENV)>;
    L& l = *reinterpret_cast<L*>(smem);
    float* Plds = reinterpret_cast<float*>(smem + ((sizeof(L) + 15) & ~size_t(15)));
    constexpr int WROW = L::WROW;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ve = 32 * wv + (lane & 31), vg = lane >> 5;   // glimpse layout: column (one of the wave's 32), chunk pair
    const int M = a.M;
    const int64_t bi = blockIdx.x % a.B;
    const int64_t ld = a.ld;
    const int C = (M + EAMRL_NCHUNK - 1) / EAMRL_NCHUNK;
    const int n0 = lane, n1 = lane + 64;
    const bool in0 = n0 < M, in1 = n1 < M;

    // ---- one-time loads: K, Lp column slices of both nodes, V columns (chunked), context rows -> LDS ------------
    // operands as PAIRS (v_pk_fma_f32: two chains per instruction): k2[i] / l2[i] = (node n0, node n1) of column 32w + i,
    // v2[i] = (chunk 2vg, chunk 2vg + 1) at chunk slot i
    f32x2 k2[32], l2[32], v2[CR];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int nn = k ? n1 : n0;
        const bool valid = nn < M;
        const float* kp = a.K + (bi * M + (valid ? nn : 0)) * ld + 32 * wv;
        const float* lp = a.Lp + (bi * M + (valid ? nn : 0)) * ld + 32 * wv;
#pragma unroll
        for (int i = 0; i < 32; i += 4) {
            const float4 kk = valid ? *reinterpret_cast<const float4*>(kp + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 ll = valid ? *reinterpret_cast<const float4*>(lp + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            k2[i][k] = kk.x; k2[i + 1][k] = kk.y; k2[i + 2][k] = kk.z; k2[i + 3][k] = kk.w;
            l2[i][k] = ll.x; l2[i + 1][k] = ll.y; l2[i + 2][k] = ll.z; l2[i + 3][k] = ll.w;
        }
    }
    {
        const float* vp = a.V + bi * M * ld + ve;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int nb = (2 * vg + c) * C;
#pragma unroll
            for (int i = 0; i < CR; ++i) {
                const int nn = nb + i;
                v2[i][c] = (i < C && nn < M) ? vp[(int64_t)nn * ld] : 0.0f;
            }
        }
        const float* Pc = (ENV == EAMRL_ENV_TSP ? a.Pb : a.Pa) + bi * M * ld;
        // context rows -> LDS, eight loads in flight per thread (a load-store loop would pay the latency 13 times)
        for (int i0 = tid; i0 < M * (RE / 4); i0 += 8 * RB) {
            float4 st[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * RB;
                const int ic = i < M * (RE / 4) ? i : M * (RE / 4) - 1;
                const int row = ic / (RE / 4), c4 = ic - row * (RE / 4);
                st[u] = *reinterpret_cast<const float4*>(Pc + (int64_t)row * ld + 4 * c4);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * RB;
                if (i < M * (RE / 4)) {
                    const int row = i / (RE / 4), c4 = i - row * (RE / 4);
                    *reinterpret_cast<float4*>(Plds + row * RE + 4 * c4) = st[u];
                }
            }
        }
        for (int i = tid; i < RH * WROW; i += RB) l.w[i] = 0.0f;   // chunk padding stays 0 for the whole episode
        if (SD)
            for (int i = tid; i < 3 * RE; i += RB) l.dynv[i] = a.dyn[i];
    }
    // slots of the two nodes in a w row: chunk c, slot i -> ((c >> 1) * CP + i) * 2 + (c & 1), i.e. the two chunks a glimpse
    // lane sums side by side (one float4 = slots i, i + 1 of both); absent nodes (n >= M) point at the row's never-read spare slot
    const int ch0 = n0 / C, ch1 = n1 / C;
    const int pos0 = in0 ? ((ch0 >> 1) * CP + (n0 - ch0 * C)) * 2 + (ch0 & 1) : 4 * CP;
    const int pos1 = in1 ? ((ch1 >> 1) * CP + (n1 - ch1 * C)) * 2 + (ch1 & 1) : 4 * CP;
    const float inv_sqrtE = 1.0f / __builtin_sqrtf((float)RE);     // one rounded constant (canonical: logit = u * inv_sqrtE)

    for (int s = MS ? (int)(blockIdx.x / a.B) : 0; s < (MS ? S : 1); s += (MS ? G : 1)) {
    const int64_t r = MS ? (int64_t)s * a.B + bi : (int64_t)blockIdx.x;

    // ---- state: mask / visited in LDS; the scalar row state lives in wavefront 0 -----------------------------------------
    if (tid < RNP) {
        l.msk[tid] = (tid < M) ? a.mask[r * M + tid] : 0;
        l.vis[tid] = ((CV || PC || OP) && tid < M) ? a.visited[r * M + tid] : 0;
        if (PC || OP) l.dem[tid] = (tid < M) ? a.demand[bi * M + tid] : 0.0f;
        if (OP || TW) {
            l.xy[2 * tid] = (tid < M) ? a.locs[(bi * M + tid) * 2] : 0.0f;
            l.xy[2 * tid + 1] = (tid < M) ? a.locs[(bi * M + tid) * 2 + 1] : 0.0f;
        }
        if (TW) {
            l.twv[tid] = (tid < M) ? a.tw[(bi * M + tid) * 2] : 0.0f;
            l.twv[RNP + tid] = (tid < M) ? a.tw[(bi * M + tid) * 2 + 1] : 0.0f;
            l.twv[2 * RNP + tid] = (tid < M) ? a.dur[bi * M + tid] : 0.0f;
        }
        if (CV) l.dem[tid] = (tid < M - 1) ? a.demand[bi * (M - 1) + tid] : 0.0f;
        if (SD) l.dem[tid] = (tid < M) ? a.rem[r * M + tid] : 0.0f;
    }
    if (tid == 0) l.done = a.done[r] != 0;
    __syncthreads();

    constexpr int fw = 0;      // the wavefront that runs the serial "finish" section of a step

    int64_t first = 0, cur = 0, istep = 1, i0 = 0;
    float used = 0.0f, vcap = 0.0f, now = 0.0f;
    int count = 0;
    float gq[2] = {0.f, 0.f}, cv[2] = {0.f, 0.f}, cv2[2] = {0.f, 0.f}, p1f[2] = {0.f, 0.f}, mydem[2] = {0.f, 0.f};
    bool done = l.done != 0;
    if (wv == fw) {
        cur = a.cur[r];
        if (ENV == EAMRL_ENV_TSP) { first = a.first[r]; istep = a.istep[r]; }
        else { used = a.used[r]; vcap = a.vcap[r]; }
        if (PC || OP) { istep = a.istep[r]; i0 = istep; }
        if (TW) now = a.time[r];
        // remaining feasible (TSP) / visited (CVRP; PCTSP: visited customers) node count, kept incrementally
        // (== the reference's mask.sum / visited.sum)
        const int c0 = (ENV == EAMRL_ENV_TSP) ? (in0 && l.msk[n0] != 0) : (in0 && l.vis[n0] != 0 && !(PC && n0 == 0));
        const int c1 = (ENV == EAMRL_ENV_TSP) ? (in1 && l.msk[n1] != 0) : (in1 && l.vis[n1] != 0);
        count = __builtin_popcountll(__ballot(c0)) + __builtin_popcountll(__ballot(c1));
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = lane + 64 * k;
            gq[k] = a.gctx ? a.gctx[bi * RE + e] : 0.0f;
            cv[k] = a.cvec[e];
            if (ENV == EAMRL_ENV_TSP && istep > 0) p1f[k] = a.Pa[(bi * M + first) * ld + e];
            const int nn = lane + 64 * k;
            if (CV && nn >= 1 && nn < M) mydem[k] = l.dem[nn - 1];
            float ctx;
            if (ENV == EAMRL_ENV_TSP) ctx = (istep == 0) ? cv[k] : p1f[k] + Plds[cur * RE + e];
            else if (PC) ctx = fma_(cv[k], (vcap - used) < 0.0f ? 0.0f : (vcap - used), Plds[cur * RE + e]);
            else ctx = fma_(cv[k], vcap - used, Plds[cur * RE + e]);
            if (TW) { cv2[k] = a.cvec[RE + e]; ctx = fma_(cv2[k], now, ctx); }
            l.q[e] = ctx + gq[k];
        }
    }
    __syncthreads();

    int t = 0;
    uint32_t st_flags = 0;
#ifdef EAMRL_STAMPS
    unsigned long long stamp_acc[5] = {0, 0, 0, 0, 0}, stamp_t = __builtin_readcyclecounter();
#endif
    for (;;) {
        // the LDS reads that open a step are issued together: episode flag and this lane's two mask bytes
        // (slots >= M hold 0 for the whole episode, so no range check is needed)
        // (+ the wave's 32 query columns, lane e = column e: broadcast by v_readlane in S1 -- one LDS read instead of eight
        //  wave-uniform float4 reads with three exposed latencies and 32 registers of buffers.  Measured and dropped: forcing the
        //  flag's read into the same wait as the other three with an empty asm -- the four pinned registers cost 14 more spilled
        //  dwords at the limit of 256, 0.485 ms instead of 0.460.)
        done = l.done != 0;
        const bool f0 = l.msk[n0] != 0, f1 = l.msk[n1] != 0;
        const float qv = l.q[32 * wv + (lane & 31)];
        if (done || t >= a.t_max) break;
        // prefetch this step's per-row inputs (latency hidden behind the glimpse)
        float nz0 = 1.0f, nz1 = 1.0f;
        int64_t given = 0;
        if (wv == fw) {
            if (a.mode == EAMRL_SAMPLE) {
                const float* nzp = a.noise + (r * a.t_max + t) * (int64_t)M;
                if (in0) nz0 = nzp[n0];
                if (in1) nz1 = nzp[n1];
            }
            if (a.mode == EAMRL_EVALUATE) given = (t < a.t_given) ? a.given[r * a.t_given + t] : 0;
        }

        // ---- S1: scores, per-head max and softmax weights of heads 2w, 2w+1 for both nodes (all inside the wave) ----
        // Branch-free and written so that the two heads' chains, butterflies and exponentials interleave.
        float Rh[2] = {0.0f, 0.0f};         // SDVRP: R_h = lane tree of w * rem, wavefront-uniform
        {
            f32x2 s2[2] = {splat2(0.0f), splat2(0.0f)};     // (node n0, node n1) per head
            float qw[2] = {0.0f, 0.0f};
#pragma unroll
            for (int d = 0; d < RD; d += 4) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    float4 qq;
                    qq.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qv), hh * RD + d));
                    qq.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qv), hh * RD + d + 1));
                    qq.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qv), hh * RD + d + 2));
                    qq.w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qv), hh * RD + d + 3));
                    if (SD) {       // q_h . wk_h (wavefront-uniform, every lane keeps its own copy)
                        const float4 dk = *reinterpret_cast<const float4*>(l.dynv + (2 * wv + hh) * RD + d);
                        qw[hh] = fma_(qq.x, dk.x, qw[hh]); qw[hh] = fma_(qq.y, dk.y, qw[hh]);
                        qw[hh] = fma_(qq.z, dk.z, qw[hh]); qw[hh] = fma_(qq.w, dk.w, qw[hh]);
                    }
                    s2[hh] = pk_fma(splat2(qq.x), k2[hh * RD + d], s2[hh]);
                    s2[hh] = pk_fma(splat2(qq.y), k2[hh * RD + d + 1], s2[hh]);
                    s2[hh] = pk_fma(splat2(qq.z), k2[hh * RD + d + 2], s2[hh]);
                    s2[hh] = pk_fma(splat2(qq.w), k2[hh * RD + d + 3], s2[hh]);
                }
            }
            f32x2 sc[2];
            float m[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if (SD)         // remaining demand of the lane's nodes (slots >= M hold 0); re-read where needed: no live range
                    s2[hh] = pk_fma((f32x2){l.dem[n0], l.dem[n1]}, splat2(qw[hh]), s2[hh]);
                const f32x2 sq = s2[hh] * splat2(0.25f);                 // 1/sqrt(16)
                sc[hh].x = f0 ? sq.x : -INFINITY;
                sc[hh].y = f1 ? sq.y : -INFINITY;
                m[hh] = vmax_raw(sc[hh].x, sc[hh].y);
            }
            m[0] = wave_max(m[0]);
            m[1] = wave_max(m[1]);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const f32x2 e2 = d_expf2_nonpos(sc[hh] - splat2(m[hh]));        // masked / absent nodes: value discarded below
                float* wrow = l.w + (2 * wv + hh) * WROW;
                const float w0 = f0 ? e2.x : 0.0f, w1 = f1 ? e2.y : 0.0f;
                wrow[pos0] = w0;                                         // absent nodes write 0 into the row's spare slot
                wrow[pos1] = w1;
                if (SD)         // R_h = lane tree of w * rem over the two 64-blocks (the second is all zeros when M <= 64)
                    Rh[hh] = wave_tree_sum(w0 * l.dem[n0]) + wave_tree_sum(w1 * l.dem[n1]);
            }
        }
        __builtin_amdgcn_wave_barrier();        // the rows are this wavefront's own: LDS executes its accesses in order
        STAMP(0);

        // ---- S2: glimpse of the wave's 32 columns: lane (column ve, half vg) sums chunks 2vg and 2vg+1, the halves meet
        //      through v_permlane32_swap, every lane ends with heads[ve] ------------------------------------------------
        float head_e;
        {
            const int h = ve >> 4;              // = 2 wv + ((lane & 31) >> 4)
            // both chunks of the lane at once: one float4 of the (pair-interleaved) weight row = slots i, i + 1 of chunk 2vg (.x, .z)
            // and chunk 2vg + 1 (.y, .w).  Z_g in the canonical order: four partial sums by slot class (i mod 4), combined as
            // (P0 + P1) + (P2 + P3); padded slots hold w = 0, so whole float4 can be added.
            static_assert(CR <= CP && CP % 2 == 0, "a chunk's CR slots must lie inside its padded row");
            const float* wp = l.w + h * WROW + vg * (2 * CP);
            f32x2 acc2 = splat2(0.0f), zc[4] = {splat2(0.0f), splat2(0.0f), splat2(0.0f), splat2(0.0f)};
            // The reads of the row run ahead of the chain that consumes them: left to itself the compiler keeps two in flight and the
            // chain below waits out an LDS latency per float4 (the glimpse was nine exposed latencies long).  TSP / CVRP / SDVRP / PCTSP
            // at <= 104 nodes (CR <= 26): eight reads up front, the remaining ones once their buffers are free (0.491 -> 0.460 ms at
            // TSP-100 x 1024).  Larger chunks and the envs with more state of their own have less register room -- spills cost more
            // than the prefetch hides (CVRPTW-100: 1.17 -> 1.12 ms with the shallower form) --: a rolling window of 4 .. 6.
            // (Measured and dropped for CR <= 26: the rolling form, 0.47 ms; the episode flag riding with the query as float2, 0.47 ms.)
            constexpr int NWQ = (CR + 1) / 2;
            constexpr bool ROLL = CR > 26 || TW || OP;
            constexpr int WD0 = ROLL ? ((CR <= 26 ? 7 : CR <= 28 ? 6 : 4) - ((TW || OP) ? 2 : 0)) : 8, WDEEP = NWQ < WD0 ? NWQ : WD0;
            float4 wq[NWQ];
#pragma unroll
            for (int i = 0; i < WDEEP; ++i) wq[i] = *reinterpret_cast<const float4*>(wp + 4 * i);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < CR; i += 2) {          // slots >= C hold w = 0 (and are skipped beyond CR)
                if (ROLL) {
                    if ((i >> 1) + WDEEP < NWQ) {      // (the buffer of the float4 consumed one round ago is free)
                        wq[(i >> 1) + WDEEP] = *reinterpret_cast<const float4*>(wp + 4 * ((i >> 1) + WDEEP));
                        asm volatile("" ::: "memory");
                    }
                } else if ((i >> 1) == NWQ - WDEEP && NWQ > WDEEP) {      // the first NWQ - WDEEP buffers are free again: the rest of the row
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int k = WDEEP; k < NWQ; ++k) wq[k] = *reinterpret_cast<const float4*>(wp + 4 * k);
                    asm volatile("" ::: "memory");
                }
                const float4 ww = wq[i >> 1];
                const f32x2 wa = (f32x2){ww.x, ww.y}, wb = (f32x2){ww.z, ww.w};
                zc[i & 3] = zc[i & 3] + wa;
                acc2 = pk_fma(wa, v2[i], acc2);
                if (i + 1 < CR) {
                    zc[(i + 1) & 3] = zc[(i + 1) & 3] + wb;
                    acc2 = pk_fma(wb, v2[i + 1], acc2);
                }
            }
            const f32x2 zz = (zc[0] + zc[1]) + (zc[2] + zc[3]);
            const float ag[2] = {acc2.x, acc2.y}, zg[2] = {zz.x, zz.y};
            // chunks 0, 1 live in lanes 0-31, chunks 2, 3 in lanes 32-63 of the same column: one swap per value leaves both in
            // every lane ([0] = the low half's, [1] = the high half's); then the canonical ((A0 + A1) + A2) + A3 over the chunks
            const auto a02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(ag[0]), __float_as_uint(ag[0]), false, false);
            const auto a13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(ag[1]), __float_as_uint(ag[1]), false, false);
            const auto z02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(zg[0]), __float_as_uint(zg[0]), false, false);
            const auto z13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(zg[1]), __float_as_uint(zg[1]), false, false);
            float A = ((__uint_as_float(a02[0]) + __uint_as_float(a13[0])) + __uint_as_float(a02[1])) + __uint_as_float(a13[1]);
            const float Z = ((__uint_as_float(z02[0]) + __uint_as_float(z13[0])) + __uint_as_float(z02[1])) + __uint_as_float(z13[1]);
            if (SD) A = fma_((lane & 16) ? Rh[1] : Rh[0], l.dynv[RE + ve], A);
            head_e = A / Z;
        }
        STAMP(1);

        // ---- S4: the wave's 32 glimpse values (lane e holds column e) broadcast by v_readlane -- scalar operands of the chains, no
        //      LDS round trip (round 3: the wave-private row cost a write -> read latency and eight reads the register-starved
        //      loop could only issue two at a time); then the logit partials of both nodes ---------------------------------------
        {
            f32x2 c2 = splat2(0.0f);            // (node n0, node n1)
            float hl = 0.0f;
#pragma unroll
            for (int e = 0; e < 32; e += 4) {
                float4 h4;
                h4.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(head_e), e));
                h4.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(head_e), e + 1));
                h4.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(head_e), e + 2));
                h4.w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(head_e), e + 3));
                if (SD) {       // heads_c . lw_c of this wavefront's column chunk
                    const float4 dl = *reinterpret_cast<const float4*>(l.dynv + 2 * RE + 32 * wv + e);
                    hl = fma_(h4.x, dl.x, hl); hl = fma_(h4.y, dl.y, hl);
                    hl = fma_(h4.z, dl.z, hl); hl = fma_(h4.w, dl.w, hl);
                }
                c2 = pk_fma(splat2(h4.x), l2[e], c2);
                c2 = pk_fma(splat2(h4.y), l2[e + 1], c2);
                c2 = pk_fma(splat2(h4.z), l2[e + 2], c2);
                c2 = pk_fma(splat2(h4.w), l2[e + 3], c2);
            }
            if (SD) c2 = pk_fma((f32x2){l.dem[n0], l.dem[n1]}, splat2(hl), c2);
            *reinterpret_cast<float2*>(l.cpart + wv * (2 * 64) + 2 * lane) = make_float2(c2.x, c2.y);   // [chunk][lane][node of the pair]
        }
        __syncthreads();
        STAMP(2);

        // ---- S5: one wavefront finishes the step: clip, mask, log-softmax, selection, env transition, next query -----
        if (wv == fw) {
            __builtin_amdgcn_s_setprio(3);   // the serial section of the step: let it win issue arbitration
            // both nodes of the lane go through the elementwise math side by side (packed fp32 instructions:
            // half the issue slots and two independent dependency chains for this single wavefront)
            float x[2], lpv[2];
            bool fe[2];
            fe[0] = f0;                          // the mask changes only at the end of this section
            fe[1] = f1;
            // the four column-chunk partials of the lane's node pair arrive as pairs (one 8-byte read each, no repacking)
            const float2 cp0 = *reinterpret_cast<const float2*>(l.cpart + 2 * lane);
            const float2 cp1 = *reinterpret_cast<const float2*>(l.cpart + 128 + 2 * lane);
            const float2 cp2 = *reinterpret_cast<const float2*>(l.cpart + 256 + 2 * lane);
            const float2 cp3 = *reinterpret_cast<const float2*>(l.cpart + 384 + 2 * lane);
            const unsigned long long fb0 = __ballot(f0), fb1 = __ballot(f1);      // feasibility bits (the mask row, in SGPRs)
            f32x2 u2 = (f32x2){cp0.x, cp0.y} + (f32x2){cp1.x, cp1.y};
            u2 = u2 + (f32x2){cp2.x, cp2.y};
            u2 = u2 + (f32x2){cp3.x, cp3.y};
            const f32x2 logit2 = u2 * splat2(inv_sqrtE);
            const bool nan_seen = (fe[0] && logit2.x != logit2.x) || (fe[1] && logit2.y != logit2.y);
            f32x2 v2 = (a.clip > 0.0f) ? d_tanhf2(logit2) * splat2(a.clip) : logit2;
            v2.x = fe[0] ? v2.x : -INFINITY;
            v2.y = fe[1] ? v2.y : -INFINITY;
            if (a.temp != 1.0f) {                               // v / 1 == v exactly
                asm volatile("" ::: "memory");                  // keep this a (uniform) branch: 2 IEEE divisions
                v2 = v2 / splat2(a.temp);
            }
            x[0] = v2.x; x[1] = v2.y;
            const float mx = wave_max(vmax_raw(x[0], x[1]));
            const f32x2 ex2 = d_expf2_nonpos(v2 - splat2(mx));
            const float e0 = fe[0] ? ex2.x : 0.0f;
            const float e1 = fe[1] ? ex2.y : 0.0f;
            // second 64-block of the lane tree: all zeros when M <= 64, and Z + 0 == Z
            const float Zl = wave_tree_sum(e0) + wave_tree_sum(e1);
            const float lse = d_logf(Zl);
            const f32x2 lp2 = (v2 - splat2(mx)) - splat2(lse);
            lpv[0] = fe[0] ? lp2.x : -INFINITY;
            lpv[1] = fe[1] ? lp2.y : -INFINITY;
            int sel;
            if (a.mode == EAMRL_SAMPLE) {
                // argmax of p / noise, lowest index on ties: the maximum by a value-only butterfly, then the first lane
                // (node) that equals it -- a NaN key (0 / 0) never equals anything and is skipped by v_max, as before
                const f32x2 k2 = d_expf2_nonpos((f32x2){lpv[0], lpv[1]}) / (f32x2){nz0, nz1};
                const float k0 = in0 ? k2.x : -INFINITY, k1 = in1 ? k2.y : -INFINITY;
                const float top = wave_max(vmax_raw(k0, k1));
                const unsigned long long b0 = __ballot(k0 == top), b1 = __ballot(k1 == top);
                sel = b0 ? __builtin_ctzll(b0) : (b1 ? 64 + __builtin_ctzll(b1) : 0);
            } else {
                // argmax of the log-probs, lowest index on ties: the maximum is known -- the lanes holding the
                // largest logit have (mx - mx) - lse = 0 - lse, and rounding is monotonic -- so the winner is the
                // first lane that equals it
                const float top = 0.0f - lse;
                const unsigned long long b0 = __ballot(in0 && lpv[0] == top), b1 = __ballot(in1 && lpv[1] == top);
                sel = b0 ? __builtin_ctzll(b0) : 64 + (b1 ? __builtin_ctzll(b1) : 64);
            }
            if (a.mode == EAMRL_EVALUATE) sel = (int)given;
            sel = __builtin_amdgcn_readfirstlane(sel);
            if (__ballot(nan_seen) != 0ull) st_flags |= EAMRL_ST_NAN_LOGITS;
            if (sel < 0 || sel >= M) { st_flags |= EAMRL_ST_INFEASIBLE; sel = 0; }
            // l.msk[sel], from the ballots instead of an LDS round trip on the serial path
            const bool sel_ok = (((sel < 64) ? (fb0 >> sel) : (fb1 >> (sel - 64))) & 1ull) != 0ull;
            if (!sel_ok) st_flags |= EAMRL_ST_INFEASIBLE;
            const float lp_sel = __int_as_float(
                (sel < 64) ? __builtin_amdgcn_readlane(__float_as_int(lpv[0]), sel)
                           : __builtin_amdgcn_readlane(__float_as_int(lpv[1]), sel - 64));
            if (lane == 0) {      // kept in LDS: a global store here would sit on every later vmcnt wait of the loop
                l.act_out[t] = (int16_t)sel;
                l.lp_out[t] = lp_sel;
            }
            // ---- env transition (TSPEnv._step / CVRPEnv._step + get_action_mask) and the next context query ----
            if (ENV == EAMRL_ENV_TSP) {
                if (istep == 0) {
                    first = sel;
#pragma unroll
                    for (int k = 0; k < 2; ++k) p1f[k] = a.Pa[(bi * M + first) * ld + lane + 64 * k];
                }
                cur = sel;
                istep += 1;
                count -= sel_ok ? 1 : 0;
                done = (count == 0);
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) { l.msk[sel] = 0; l.done = done; }
#pragma unroll
                for (int k = 0; k < 2; ++k) l.q[lane + 64 * k] = (p1f[k] + Plds[cur * RE + lane + 64 * k]) + gq[k];
            } else if (OP) {
                // OPEnv._step + get_action_mask (op/env.py:69-102,149-165)
                const float cx = l.xy[2 * sel], cy = l.xy[2 * sel + 1];
                {
                    const float dx = cx - l.xy[2 * cur], dy = cy - l.xy[2 * cur + 1];
                    used = used + __builtin_sqrtf(fma_(dy, dy, dx * dx));
                }
                done = (sel == 0) && (istep > 0);
                cur = sel;
                istep += 1;
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) { l.vis[sel] = 1; l.done = done; }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 2; ++k) l.q[lane + 64 * k] = fma_(cv[k], vcap - used, Plds[cur * RE + lane + 64 * k]) + gq[k];
                const int v0 = l.vis[0] != 0;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int nn = lane + 64 * k;
                    if (nn >= 1 && nn < M) {
                        const float dx = l.xy[2 * nn] - cx, dy = l.xy[2 * nn + 1] - cy;
                        const int exceeds = (used + __builtin_sqrtf(fma_(dy, dy, dx * dx))) > l.dem[nn];
                        l.msk[nn] = !((l.vis[nn] != 0) | v0 | exceeds);
                    }
                }
                if (lane == 0) l.msk[0] = 1;
            } else if (PC) {
                // PCTSPEnv._step + get_action_mask (pctsp/env.py:64-97,156-163)
                used = used + l.dem[sel];
                done = (istep > 0) && (sel == 0);
                cur = sel;
                istep += 1;
                count += (sel != 0 && l.vis[sel] == 0);
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) { l.vis[sel] = 1; l.done = done; }
                __builtin_amdgcn_wave_barrier();
                const float state = (vcap - used) < 0.0f ? 0.0f : (vcap - used);
#pragma unroll
                for (int k = 0; k < 2; ++k) l.q[lane + 64 * k] = fma_(cv[k], state, Plds[cur * RE + lane + 64 * k]) + gq[k];
                const int v0 = l.vis[0] != 0;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int nn = lane + 64 * k;
                    if (nn >= 1 && nn < M) l.msk[nn] = !((l.vis[nn] != 0) | v0);
                }
                if (lane == 0) l.msk[0] = !((used < 1.0f) && (count < M - 1));
            } else if (SD) {
                // SDVRPEnv._step + get_action_mask (sdvrp/env.py:58-92,137-146): deliver min(remaining demand, free capacity)
                const float selrem = l.dem[sel];
                const float free_cap = vcap - used;
                const float delivered = selrem < free_cap ? selrem : free_cap;
                used = (used + delivered) * (sel != 0 ? 1.0f : 0.0f);
                cur = sel;
                const float left = selrem + (-delivered);
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) l.dem[sel] = left;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 2; ++k) l.q[lane + 64 * k] = fma_(cv[k], vcap - used, Plds[cur * RE + lane + 64 * k]) + gq[k];
                const bool full = used >= vcap;
                int free_n = 0, rem_n = 0;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int nn = lane + 64 * k;
                    if (nn < M) {
                        const float rv = l.dem[nn];
                        rem_n |= rv > 0.0f;
                        if (nn >= 1) {
                            const int blocked = (rv == 0.0f) | full;
                            l.msk[nn] = !blocked;
                            free_n |= !blocked;
                        }
                    }
                }
                const bool any_free = __ballot(free_n) != 0ull;
                done = __ballot(rem_n) == 0ull;
                if (lane == 0) { l.msk[0] = !((cur == 0) && any_free); l.done = done; }
            } else {
                const int N = M - 1;
                float cx = 0.0f, cy = 0.0f;
                if (TW) {       // clock (cvrptw/env.py:118-138)
                    cx = l.xy[2 * sel]; cy = l.xy[2 * sel + 1];
                    const float dx = l.xy[2 * cur] - cx, dy = l.xy[2 * cur + 1] - cy;
                    const float arrive = now + __builtin_sqrtf(fma_(dy, dy, dx * dx));
                    const float ws = l.twv[sel];
                    const float start = arrive > ws ? arrive : ws;
                    now = (sel != 0 ? 1.0f : 0.0f) * (start + l.twv[2 * RNP + sel]);
                }
                int di = sel - 1;
                di = di < 0 ? 0 : (di > N - 1 ? N - 1 : di);
                used = (used + l.dem[di]) * (sel != 0 ? 1.0f : 0.0f);
                cur = sel;
                count += (l.vis[sel] == 0);
                done = (count == M);
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) { l.vis[sel] = 1; l.done = done; }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    float qv = fma_(cv[k], vcap - used, Plds[cur * RE + lane + 64 * k]);
                    if (TW) qv = fma_(cv2[k], now, qv);
                    l.q[lane + 64 * k] = qv + gq[k];
                }
                const float lim = vcap + 1e-5f;
                int free_n = 0;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int nn = lane + 64 * k;
                    if (nn >= 1 && nn < M) {
                        const int blocked = (l.vis[nn] != 0) | ((mydem[k] + used) > lim);
                        int ok = !blocked;
                        if (TW) {
                            const float dx = cx - l.xy[2 * nn], dy = cy - l.xy[2 * nn + 1];
                            ok &= (now + __builtin_sqrtf(fma_(dy, dy, dx * dx))) <= l.twv[RNP + nn];
                        }
                        l.msk[nn] = ok;
                        free_n |= !blocked;
                    }
                }
                const bool any_free = __ballot(free_n) != 0ull;
                if (lane == 0) {
                    int ok0 = !((cur == 0) && any_free);
                    if (TW) {
                        const float dx = cx - l.xy[0], dy = cy - l.xy[1];
                        ok0 &= (now + __builtin_sqrtf(fma_(dy, dy, dx * dx))) <= l.twv[RNP];
                    }
                    l.msk[0] = ok0;
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }
        STAMP(3);
        __syncthreads();
        STAMP(4);
        ++t;
    }
#ifdef EAMRL_STAMPS
    if (wv == fw && lane == 0) {
        for (int i = 0; i < 5; ++i) atomicAdd(&g_stamps[i], stamp_acc[i]);
        atomicAdd(&g_stamps[5], (unsigned long long)t);
    }
#endif

    // ---- write back the episode and the final state ---------------------------------------------------------------
    __syncthreads();
    for (int i = tid; i < t; i += RB) {
        a.action[r * a.t_max + i] = l.act_out[i];
        a.logp[r * a.t_max + i] = l.lp_out[i];
    }
    if (tid < M) {
        a.mask[r * M + tid] = l.msk[tid];
        if (CV || PC || OP) a.visited[r * M + tid] = l.vis[tid];
        if (SD) a.rem[r * M + tid] = l.dem[tid];
    }
    if (wv == fw && lane == 0) {
        a.cur[r] = cur;
        a.done[r] = done ? 1 : 0;
        if (ENV == EAMRL_ENV_TSP) { a.first[r] = first; a.istep[r] = istep; }
        else a.used[r] = used;
        if (PC || OP) a.istep[r] = i0;    // launch_rollout_pad adds the batch's step count
        if (TW) a.time[r] = now;
        atomicMax(a.steps_out, t);
        if (!done) st_flags |= EAMRL_ST_STEP_OVERRUN;
        if (st_flags) atomicOr(a.status, st_flags);
    }
    if (MS) __syncthreads();      // the next start re-initialises the row state in LDS
    }
}

template <int ENV, int CP, int CR, bool MS>
int launch_ms(const DecArgs& a, int S, int G, hipStream_t st)
{
    const size_t lds = ((sizeof(ResLds<CP, res_tmax(ENV), res_sdf(ENV), res_xyf(ENV), res_twf(ENV)>) + 15) & ~size_t(15)) +
                       (size_t)a.M * RE * sizeof(float);
    auto k = k_rollout_resident<ENV, CP, CR, MS>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return EAMRL_E_LAUNCH;
    hipLaunchKernelGGL(k, dim3((unsigned)(MS ? a.B * G : a.R)), dim3(RB), lds, st, a, S, G);
    return 0;
}

template <int ENV, int CP, int CR>
int launch_cp(const DecArgs& a, hipStream_t st)
{
    // multistart batch: enough workgroups per instance to fill the chip four times over, the rest of the starts
    // are looped inside the workgroup on the register-resident operands
    const int64_t S = a.R / a.B;
    int rc;
    if (S > 1 && a.R % a.B == 0 && !g_debug[6]) {
        int64_t G = (4 * 512 + a.B - 1) / a.B;
        G = G < 1 ? 1 : (G > S ? S : G);
        rc = launch_ms<ENV, CP, CR, true>(a, (int)S, (int)G, st);
    } else {
        rc = launch_ms<ENV, CP, CR, false>(a, 1, 1, st);
    }
    if (rc) return rc;
    launch_rollout_pad(ENV, a, st);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

template <int ENV>
int launch_env(const DecArgs& a, hipStream_t st)
{
    const int C = (a.M + EAMRL_NCHUNK - 1) / EAMRL_NCHUNK;
    if (a.t_max > res_tmax(ENV)) return EAMRL_E_ARG;
    if (C <= 8) return launch_cp<ENV, 8, 8>(a, st);
    if (C <= 16) return launch_cp<ENV, 16, 16>(a, st);
    if (C <= 26) return launch_cp<ENV, 28, 26>(a, st);      // M <= 104 (TSP-100; CVRP-100: M = 101).  (An odd CR leaves the
                                                            // pair-register array of the values unaligned: the 25-slot variant spilled.)
    if (C <= 28) return launch_cp<ENV, 28, 28>(a, st);
    return launch_cp<ENV, 32, 32>(a, st);
}

}  // namespace

#ifdef EAMRL_STAMPS
extern "C" __attribute__((visibility("default"))) int eamrl_debug_read_stamps(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

bool rollout_resident_supports(int env, const DecArgs& a)
{
    return a.E == RE && a.H == RH && a.M <= RNP && a.M >= 2 && a.ld % 4 == 0 && a.t_max <= res_tmax(env);
}

int launch_rollout_resident(int env, const DecArgs& a, hipStream_t st)
{
    return env == EAMRL_ENV_TSP ? launch_env<EAMRL_ENV_TSP>(a, st)
         : env == EAMRL_ENV_CVRP ? launch_env<EAMRL_ENV_CVRP>(a, st)
         : env == EAMRL_ENV_SDVRP ? launch_env<EAMRL_ENV_SDVRP>(a, st)
         : env == EAMRL_ENV_PCTSP ? launch_env<EAMRL_ENV_PCTSP>(a, st)
         : env == EAMRL_ENV_OP ? launch_env<EAMRL_ENV_OP>(a, st) : launch_env<EAMRL_ENV_CVRPTW>(a, st);
}

}  // namespace eamrl
