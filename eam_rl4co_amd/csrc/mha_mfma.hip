// Encoder self-attention on the fp32 matrix cores (head dim 16, N <= 128 nodes): scores and the weighted sum of
// values run as v_mfma_f32_32x32x2_f32, the softmax stays on the VALU of the same wavefront.
//
// Reference: MultiHeadAttention.forward, rl4co/models/nn/attention.py:66-136 (scaled_dot_product_attention of the
// packed "(three h d)" projection).
//
// One 128-thread workgroup = (instance, pair of heads); its Q, K, V slices are staged once in LDS (row stride
// 33 floats: the MFMA operand reads -- 32 different rows, one column -- hit 32 different banks; 39 KB at N = 100, so
// four workgroups = two wavefronts per SIMD share a CU and one's softmax overlaps the other's MFMAs).  Wavefront w
// owns head w of the pair and walks the query tiles of 32:
//   S^T tile   [32 keys x 32 queries] = K (A operand, rows = keys) x Q^T (B operand, columns = queries), 8 MFMAs per
//              key tile (k = 2 of the 16 head dims each).  An accumulator lane then holds ONE query column and 16 of the
//              tile's keys, so the row max is a per-lane maximum plus one 32-lane swap.
//   softmax    w = d_expf(0.25 s - max) per element (packed fp32 math), padded keys masked to 0.
//   O^T tile   [32 rows x 32 queries] += A x W^T: the accumulator registers of S^T ARE the B operand of this product
//              (no data movement) because the keys were assigned to A rows in the order the accumulator layout
//              produces: register i of key tile t holds key 32t + 2i + (lane / 32), which is exactly the pair of k
//              indices MFMA number (t, i) consumes.  A rows 0-15 = V^T (head dims), rows 16 and 20 = 1.0, so the same
//              instruction stream also yields Z = sum of w in both lane halves.
// MEASURED (MI355X, B = 1024, N = 100, profiles/README.md): 194 us against 186 us for the VALU kernel
// k_mha_encoder_x2, so this kernel is NOT the default (eamrl_debug_set key 7 selects it).  Switching phases off one
// at a time shows that their costs simply add -- staging + operand reads 54 us, score MFMAs 33, softmax 54, value
// MFMAs 49, divide + store 8 -- i.e. the fp32 MFMAs do not overlap with the VALU work of the other wavefront on the
// SIMD: on gfx950 the f32 MFMA rate equals the packed-fp32 VALU rate and the two compete for the same FMA lanes.
// MFMA buys issue slots for fp32, not extra FLOPs; with the 28 % padding of N = 100 to 128 and the extra
// row of ones for Z it loses what it gains.  Kept as a tested reference for a future 16x16x4 variant.
// Canonical arithmetic: an f32 MFMA accumulates along k as an ordered fmaf chain, k = head dim ascending
// for the scores and key index ascending for the values, both starting from 0 -- the defined order of the oracle --
// and fma(1, w, Z) == Z + w exactly.  Padded keys contribute w = +0.  Results are bit-identical to k_mha_encoder.
#include "kernels.hpp"

namespace eamrl {

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

#ifdef EAMRL_STAMPS   // development build only (tools/build_stamps.sh): per-phase cycle sums of every wavefront
__device__ unsigned long long g_mha_stamps[8];
#define MSTAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); st_acc[i] += now_ - st_t; st_t = now_; } while (0)
#else
#define MSTAMP(i) do { } while (0)
#endif
constexpr int HG = 2;                 // heads per workgroup (one wavefront each)
constexpr int MW = HG * 16;           // floats per LDS row
constexpr int MS = MW + 1;            // LDS row stride (floats)
constexpr int MT = HG * 64;           // threads

// key held by A row m of key tile t  <->  accumulator register i = 4*(m/8) + m%4 of lane half (m/4)%2
__device__ __forceinline__ int key_of_row(int t, int m) { return 32 * t + 2 * (4 * (m >> 3) + (m & 3)) + ((m >> 2) & 1); }

template <int KT>      // key (= query) tiles of 32: N <= 32*KT
__global__ __launch_bounds__(MT, 2) void k_mha_mfma(const float* __restrict__ qkv, float* __restrict__ out, int N, int E, int H)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qs = reinterpret_cast<float*>(smem);        // [N][MS]
    float* ks = qs + (size_t)N * MS;
    float* vs = ks + (size_t)N * MS;
    float* cs = vs + (size_t)N * MS;                   // [32] constants: 24 zeros, then 1.0 (targets of padded reads)
    if (threadIdx.x < 32) cs[threadIdx.x] = threadIdx.x == 24 ? 1.0f : 0.0f;
    const int groups = H / HG;
    const int64_t b = blockIdx.x / groups;
    const int h0 = (int)(blockIdx.x - b * groups) * HG;
    const float* base = qkv + b * (int64_t)N * 3 * E;
#ifdef EAMRL_STAMPS
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_t = __builtin_readcyclecounter();
#endif
    // all global reads of the slice are issued before the first LDS write waits for one (a load-store loop would
    // pay the HBM/L2 latency once per iteration)
    {
        constexpr int NU = (32 * KT * 3 * (MW / 4) + MT - 1) / MT;
        const int total = N * 3 * (MW / 4);
        float4 stage[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int i = threadIdx.x + u * MT;
            const int ic = i < total ? i : total - 1;
            const int n = ic / (3 * (MW / 4)), rem = ic - n * (3 * (MW / 4));
            const int seg = rem / (MW / 4), c4 = rem - seg * (MW / 4);
            stage[u] = *reinterpret_cast<const float4*>(base + (int64_t)n * 3 * E + seg * E + h0 * 16 + 4 * c4);
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int i = threadIdx.x + u * MT;
            if (i < total) {
                const int n = i / (3 * (MW / 4)), rem = i - n * (3 * (MW / 4));
                const int seg = rem / (MW / 4), c4 = rem - seg * (MW / 4);
                float* dst = qs + (size_t)seg * N * MS + n * MS + 4 * c4;
                dst[0] = stage[u].x; dst[1] = stage[u].y; dst[2] = stage[u].z; dst[3] = stage[u].w;
            }
        }
    }
    __syncthreads();
    MSTAMP(0);

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int hc = wv * 16;                           // this head's columns inside the LDS rows

    // ---- operands that stay in registers for all query tiles ------------------------------------------------------------
    float kop[KT][8];      // A operand of S^T: K[key_of_row(t, col)][2s + half]
    float vop[KT][16];     // A operand of O^T: V[32t + 2i + half][col] (col < 16), 1.0 (col 16, 20), else 0
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        // branch-free: padded rows / constant rows read from the constants area, so every operand is ONE unguarded
        // ds_read whose address does the selecting (guarded reads would serialise ~100 LDS round trips)
        const int key = key_of_row(t, col);
        const float* kp = key < N ? ks + key * MS + hc + half : cs;
#pragma unroll
        for (int s = 0; s < 8; ++s) kop[t][s] = kp[2 * s];
        const float* fillp = cs + ((col == 16 || col == 20) ? 24 : 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kv = 32 * t + 2 * i + half;
            const float* vp = col < 16 ? (kv < N ? vs + kv * MS + hc + col : cs) : fillp;
            vop[t][i] = *vp;
        }
    }
    const int nk_last = N - 32 * (KT - 1);            // valid keys of the last key tile (1..32)


    MSTAMP(1);
    for (int qt = 0; qt < KT; ++qt) {
        const int query = 32 * qt + col;
        float qop[8];
        const float* qp = query < N ? qs + query * MS + hc + half : cs;
#pragma unroll
        for (int s = 0; s < 8; ++s) qop[s] = qp[2 * s] * 0.25f;     // 1/sqrt(16) folded into q: a power of two, so
                                                                    // chain(0.25 q, k) == 0.25 chain(q, k) bit for bit

        // ---- scores (transposed): acc[t] lane = query column, 16 keys ----------------------------------------------------
        floatx16 acc[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
#pragma unroll
            for (int s = 0; s < 8; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kop[t][s], qop[s], acc[t], 0, 0, 0);
        }
        MSTAMP(2);
        // ---- softmax weights in place ---------------------------------------------------------------------------------------
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            if (t == KT - 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (2 * i + half >= nk_last) acc[t][i] = -INFINITY;          // padded key
            }
#pragma unroll
            for (int i = 0; i < 16; i += 2) m = vmax3_raw(m, acc[t][i], acc[t][i + 1]);
        }
        {
            auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            m = vmax_raw(__uint_as_float(r[0]), __uint_as_float(r[1]));
        }
        const f32x2 m2 = splat2(m);
#pragma unroll
        for (int t = 0; t < KT; ++t) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                if (t == KT - 1 && 2 * i >= nk_last) {        // both lane halves padded: uniform skip
                    acc[t][i] = 0.0f; acc[t][i + 1] = 0.0f;
                    continue;
                }
                const f32x2 e2 = d_expf2_nonpos((f32x2){acc[t][i], acc[t][i + 1]} - m2);
                acc[t][i] = (t == KT - 1 && 2 * i + half >= nk_last) ? 0.0f : e2.x;
                acc[t][i + 1] = (t == KT - 1 && 2 * (i + 1) + half >= nk_last) ? 0.0f : e2.y;
            }
        }
        MSTAMP(3);
        // ---- weighted values and Z ------------------------------------------------------------------------------------------
        floatx16 o;
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = 0.0f;
#pragma unroll
        for (int t = 0; t < KT; ++t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (t == KT - 1 && 2 * i >= nk_last) continue;                   // only zeros left: skipping them is exact
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(vop[t][i], acc[t][i], o, 0, 0, 0);
            }
        }
        MSTAMP(4);
        // rows of o in this lane: i = 0..3 -> head dim 4*half + i; i = 4..7 -> 8 + 4*half + (i-4); i = 8 -> Z (rows 16 / 20)
        if (query < N) {
            const float Z = o[8];
            float* dst = out + (b * N + query) * (int64_t)E + (h0 + wv) * 16 + 4 * half;
            *reinterpret_cast<float4*>(dst) = make_float4(o[0] / Z, o[1] / Z, o[2] / Z, o[3] / Z);
            *reinterpret_cast<float4*>(dst + 8) = make_float4(o[4] / Z, o[5] / Z, o[6] / Z, o[7] / Z);
        }
        MSTAMP(5);
    }
#ifdef EAMRL_STAMPS
    if ((threadIdx.x & 63) == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(&g_mha_stamps[i], st_acc[i]);
        atomicAdd(&g_mha_stamps[6], 1ull);
    }
#endif
}

template <int KT>
int launch_kt(const float* qkv, float* out, int64_t B, int N, int E, int H, hipStream_t st)
{
    const size_t lds = ((size_t)3 * N * MS + 32) * sizeof(float);
    auto k = k_mha_mfma<KT>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return EAMRL_E_LAUNCH;
    hipLaunchKernelGGL(k, dim3((unsigned)(B * (H / HG))), dim3(MT), lds, st, qkv, out, N, E, H);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

}  // namespace

#ifdef EAMRL_STAMPS
extern "C" __attribute__((visibility("default"))) int eamrl_debug_read_mha_stamps(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mha_stamps), sizeof(g_mha_stamps)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_mha_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

bool mha_mfma_supports(int64_t B, int N, int E, int H, const float* qkv, const float* out)
{
    return H > 0 && E == 16 * H && H % HG == 0 && N >= 1 && N <= 128 && B * (H / HG) <= 0x7fffffffLL &&
           ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0;
}

int launch_mha_mfma(const float* qkv, float* out, int64_t B, int N, int E, int H, hipStream_t st)
{
    const int kt = (N + 31) / 32;
    if (kt == 1) return launch_kt<1>(qkv, out, B, N, E, H, st);
    if (kt == 2) return launch_kt<2>(qkv, out, B, N, E, H, st);
    if (kt == 3) return launch_kt<3>(qkv, out, B, N, E, H, st);
    return launch_kt<4>(qkv, out, B, N, E, H, st);
}

}  // namespace eamrl
