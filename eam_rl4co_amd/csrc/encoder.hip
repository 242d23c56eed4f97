// One-shot encoder + cache kernels (fp32, defined k-ordered accumulation).
//
//   k_linear_mfma     y = [res +] act(bias + x W^T) on v_mfma_f32_32x32x2_f32: the f32 MFMA accumulates
//                     along K as an ordered fma chain (k0 then k1, instruction after instruction), i.e.
//                     bit-for-bit the canonical chain(x, w, K, bias) of DESIGN.md
//   k_linear_valu     same result on the VALU (cross-check kernel, eamrl_debug_set(0, 1))
//   k_small_linear    in_dim <= 4 (init embeddings): plain fma chain per output
//   k_mha_encoder     per (instance, head) self-attention: two passes over the keys (max, then exp/sum/PV)
//   k_norm_*          BatchNorm1d(eval) / InstanceNorm1d(affine)
//   k_mean_nodes      embeddings.mean(1)
//
// Reference: rl4co/models/zoo/am/encoder.py:70-91, rl4co/models/nn/graph/attnnet.py:16-103,
// rl4co/models/nn/attention.py:66-136, rl4co/models/nn/ops.py:32-56, rl4co/models/nn/mlp.py:52-61,
// rl4co/models/nn/env_embeddings/init.py:55-68,115-138, rl4co/models/zoo/am/decoder.py:206-235.
#include <cstdlib>

#include "kernels.hpp"

namespace eamrl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------
// GEMM  y[r][j] = bias[j] + sum_k x[r][k] * w(j,k),  w(j,k) = WT ? W[k*ldw + j] : W[j*ldw + k]
// block = 256 threads (4 waves, 2x2), tile 128 rows x 128 cols, each wave 64x64 = 2x2 MFMA 32x32 tiles
// ---------------------------------------------------------------------------------------------------
constexpr int GT = 128;       // block tile (rows and cols)
constexpr int GK = 32;        // k tile
constexpr int GS = GK + 4;    // LDS row stride in floats: 144 B = 9 x 16 B -> conflict-free ds_read_b128 / ds_write_b128

// Per-column epilogue constants of a fused BatchNorm1d(eval): y = fma(v, scale, shift)
__device__ __forceinline__ void bn_consts(const GemmArgs& g, int j, float& scale, float& shift)
{
    scale = g.bn_gamma[j] / __builtin_sqrtf(g.bn_var[j] + g.bn_eps);
    const float ms = g.bn_mean[j] * scale;
    shift = g.bn_beta[j] - ms;
}

// One 128 x 32 operand tile: 1024 float4, 4 per thread; lanes 0-7 cover one 128-B row segment.
// Addresses are a block-uniform base (SGPRs) plus four fixed 32-bit per-thread element offsets, computed once.
// Out-of-range rows / columns are clamped to the last valid one (their products land in accumulator rows /
// columns that are never stored); in_dim is a multiple of the k-tile on this path (other shapes take the VALU kernel).
// WT tiles (W stored [k][j]) are read as float4 along j and transposed on the way into LDS.
struct TileOffs { int o[4]; };

template <bool WT>
__device__ __forceinline__ TileOffs gemm_offsets(int64_t ldm, int nvalid)
{
    // nvalid: rows (or columns) of this block's tile that exist, 1..128
    TileOffs t;
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int f = tid + 256 * p;
        if (!WT) {
            int rr = f >> 3;
            rr = rr < nvalid ? rr : nvalid - 1;
            t.o[p] = rr * (int)ldm + 4 * (f & 7);
        } else {
            int jj = 4 * (f & 31);
            jj = jj + 3 < nvalid ? jj : nvalid - 4;          // out_dim % 4 == 0 on this path
            t.o[p] = (f >> 5) * (int)ldm + jj;
        }
    }
    return t;
}

#define GEMM_FETCH(WT_, base_, ldm_, off_, k0_, r0_, r1_, r2_, r3_)                                   \
    do {                                                                                              \
        const float* kb_ = (WT_) ? (base_) + (int64_t)(k0_) * (ldm_) : (base_) + (k0_);              \
        r0_ = *reinterpret_cast<const float4*>(kb_ + (off_).o[0]);                                    \
        r1_ = *reinterpret_cast<const float4*>(kb_ + (off_).o[1]);                                    \
        r2_ = *reinterpret_cast<const float4*>(kb_ + (off_).o[2]);                                    \
        r3_ = *reinterpret_cast<const float4*>(kb_ + (off_).o[3]);                                    \
    } while (0)

// LDS image: plain row-major [128][GS] (float4 stores of the fetched registers as they are -- any register
// shuffle here makes hipcc wait for the global loads before the MFMA loop instead of after it).
template <bool WT>
__device__ __forceinline__ void gemm_stash1(float* S, int p, const float4 v)
{
    const int f = threadIdx.x + 256 * p;
    if (!WT) {
        *reinterpret_cast<float4*>(S + (f >> 3) * GS + 4 * (f & 7)) = v;
    } else {
        const int kk = f >> 5, j4 = f & 31;
        S[(4 * j4 + 0) * GS + kk] = v.x;
        S[(4 * j4 + 1) * GS + kk] = v.y;
        S[(4 * j4 + 2) * GS + kk] = v.z;
        S[(4 * j4 + 3) * GS + kk] = v.w;
    }
}
#define GEMM_STASH(WT_, S_, r0_, r1_, r2_, r3_)  \
    do { gemm_stash1<WT_>(S_, 0, r0_); gemm_stash1<WT_>(S_, 1, r1_); gemm_stash1<WT_>(S_, 2, r2_); gemm_stash1<WT_>(S_, 3, r3_); } while (0)

// MFMA operands of 4 consecutive k for one 32-row strip: lane half lk reads the aligned pair (k+2lk, k+2lk+1);
// one v_permlane32_swap turns ([k | k+2], [k+1 | k+3]) into ([k | k+1], [k+2 | k+3]) = the operands of the two
// k-steps (lanes 0-31 carry the lower k of a step, lanes 32-63 the higher one).
__device__ __forceinline__ void pair_swap(const float2 v, float& first, float& second)
{
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v.x), __float_as_uint(v.y), false, false);
    first = __uint_as_float(r[0]);
    second = __uint_as_float(r[1]);
}

// Tile = BM rows x 128 columns, 4 waves as 2 x 2, each wave (BM/2) x 64 = (BM/64) x 2 MFMA tiles of 32 x 32.
// BM = 64 gives twice as many (half-size) tiles: better balance over the 256 CUs and more blocks per CU.
// EPI: -1 = epilogue configured at run time (any shape / alignment); 0..3 = compile-time epilogue for aligned, whole-tile
// outputs (0 plain, 1 ReLU, 2 residual, 3 residual + BatchNorm): no option selects and one 64-bit address per tile --
// on gfx950 every VALU instruction of a GEMM is paid out of its fp32-MFMA time.
template <bool WT, int BM, int EPI>
__global__ __launch_bounds__(256, 2) void k_linear_mfma(GemmArgs g)
{
    constexpr int TA = BM / 64;                     // MFMA row tiles per wave; also A-tile float4 per thread / 2
    constexpr int NA = BM / 32;                     // A-tile float4 per thread (BM * 8 float4 / 256 threads)
    constexpr int SMEM_FLOATS = (BM + GT) * GS > 4 * 32 * 68 ? (BM + GT) * GS : 4 * 32 * 68;
    __shared__ __attribute__((aligned(16))) float smem_ab[SMEM_FLOATS];   // A tile | B tile, reused by the epilogue slabs
    float* As = smem_ab;
    float* Bs = smem_ab + BM * GS;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 1, wc = wv & 1;           // wave position in the 2 x 2 grid
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int col0 = blockIdx.y * GT;
    const int li = lane & 31, lk = lane >> 5;

    f32x16 acc[TA][2];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int j = col0 + wc * 64 + b * 32 + li;           // C column of this lane
            const float bj = (g.bias && j < g.out_dim) ? g.bias[j] : 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = bj;
        }

    float4 ta0, ta1, ta2, ta3, tb0, tb1, tb2, tb3;    // register-staged next operand tiles
    ta2 = ta3 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t rows_left = g.rows - row0;
    const TileOffs oa = gemm_offsets<false>(g.ldx, rows_left < BM ? (int)rows_left : BM);
    const TileOffs ob = gemm_offsets<WT>(g.ldw, g.out_dim - col0 < GT ? g.out_dim - col0 : GT);
    const float* xbase = g.x + row0 * g.ldx;                                        // block-uniform bases
    const float* wbase = WT ? g.W + col0 : g.W + (int64_t)col0 * g.ldw;
#define FETCH_A(k0_)                                                                                   \
    do {                                                                                               \
        const float* kb_ = xbase + (k0_);                                                              \
        ta0 = *reinterpret_cast<const float4*>(kb_ + oa.o[0]);                                         \
        ta1 = *reinterpret_cast<const float4*>(kb_ + oa.o[1]);                                         \
        if (NA == 4) {                                                                                 \
            ta2 = *reinterpret_cast<const float4*>(kb_ + oa.o[2]);                                     \
            ta3 = *reinterpret_cast<const float4*>(kb_ + oa.o[3]);                                     \
        }                                                                                              \
    } while (0)
#define STASH_A()                                                                                      \
    do {                                                                                               \
        gemm_stash1<false>(As, 0, ta0);                                                                \
        gemm_stash1<false>(As, 1, ta1);                                                                \
        if (NA == 4) { gemm_stash1<false>(As, 2, ta2); gemm_stash1<false>(As, 3, ta3); }               \
    } while (0)
    FETCH_A(0);
    GEMM_FETCH(WT, wbase, g.ldw, ob, 0, tb0, tb1, tb2, tb3);
    STASH_A();
    GEMM_STASH(WT, Bs, tb0, tb1, tb2, tb3);
    __syncthreads();
    for (int k0 = 0; k0 < g.in_dim; k0 += GK) {
        const bool more = k0 + GK < g.in_dim;
        if (more) {   // next tile's global loads fly while this tile is multiplied
            FETCH_A(k0 + GK);
            GEMM_FETCH(WT, wbase, g.ldw, ob, k0 + GK, tb0, tb1, tb2, tb3);
        }
        // 8 groups of 4 k values (= 2 MFMA k-steps each); operand pairs come from LDS as aligned 8-byte reads
        const float* arow = As + (wr * (BM / 2) + li) * GS + 2 * lk;
        const float* brow = Bs + (wc * 64 + li) * GS + 2 * lk;
#pragma unroll
        for (int kq = 0; kq < GK; kq += 4) {
            float a0[TA], a1[TA], b0[2], b1[2];
#pragma unroll
            for (int a = 0; a < TA; ++a) pair_swap(*reinterpret_cast<const float2*>(arow + a * 32 * GS + kq), a0[a], a1[a]);
#pragma unroll
            for (int b = 0; b < 2; ++b) pair_swap(*reinterpret_cast<const float2*>(brow + b * 32 * GS + kq), b0[b], b1[b]);
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[a], b0[b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[a], b1[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            STASH_A();
            GEMM_STASH(WT, Bs, tb0, tb1, tb2, tb3);
            __syncthreads();
        }
    }
#undef FETCH_A
#undef STASH_A
    // Epilogue through LDS: each wave parks a 32 x 64 slab of its accumulators (C/D layout: col = lane&31,
    // row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) and reads it back row-major, so residual loads and output
    // stores are float4 per lane and 256 contiguous bytes per 16 lanes.
    constexpr int CS = 68;                                  // slab row stride (floats): 272 B keeps float4 alignment
    float* slab = smem_ab + wv * (32 * CS);                 // 4 waves x 8.5 KB inside the operand tile storage
    static_assert(4 * 32 * CS <= SMEM_FLOATS, "epilogue slabs must fit the shared buffer");
    const int er = lane >> 4, ec = (lane & 15) * 4;         // this lane's (row within 4, first column) when reading
    const int jbase = col0 + wc * 64 + ec;
    float sc4[4] = {1.f, 1.f, 1.f, 1.f}, sh4[4] = {0.f, 0.f, 0.f, 0.f};
    if (EPI == 3 || (EPI < 0 && g.bn_gamma)) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (jbase + q < g.out_dim) bn_consts(g, jbase + q, sc4[q], sh4[q]);
    }
    if (EPI >= 0) {
        // fast epilogue: whole column tiles, 16-byte aligned rows (checked by the launcher)
        const int64_t rfirst = row0 + wr * (BM / 2) + er;
#pragma unroll
        for (int a = 0; a < TA; ++a) {
            __syncthreads();
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    slab[((i & 3) + 8 * (i >> 2) + 4 * lk) * CS + b * 32 + li] = acc[a][b][i];
            __syncthreads();
            float* yp = g.y + (rfirst + a * 32) * g.ldy + jbase;
            const float* rp = EPI >= 2 ? g.res + (rfirst + a * 32) * g.ldres + jbase : nullptr;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                if (rfirst + a * 32 + 4 * p < g.rows) {
                    float4 v = *reinterpret_cast<const float4*>(slab + (er + 4 * p) * CS + ec);
                    if (EPI == 1) {
                        v.x = !(v.x > 0.0f) ? 0.0f : v.x; v.y = !(v.y > 0.0f) ? 0.0f : v.y;
                        v.z = !(v.z > 0.0f) ? 0.0f : v.z; v.w = !(v.w > 0.0f) ? 0.0f : v.w;
                    }
                    if (EPI >= 2) {
                        const float4 r4 = *reinterpret_cast<const float4*>(rp);
                        v.x = r4.x + v.x; v.y = r4.y + v.y; v.z = r4.z + v.z; v.w = r4.w + v.w;
                    }
                    if (EPI == 3) {
                        v.x = fma_(v.x, sc4[0], sh4[0]); v.y = fma_(v.y, sc4[1], sh4[1]);
                        v.z = fma_(v.z, sc4[2], sh4[2]); v.w = fma_(v.w, sc4[3], sh4[3]);
                    }
                    *reinterpret_cast<float4*>(yp) = v;
                }
                yp += 4 * g.ldy;
                if (EPI >= 2) rp += 4 * g.ldres;
            }
        }
        return;
    }
    const bool vec_ok = ((g.ldy & 3) == 0) && (((uintptr_t)g.y & 15) == 0) && (jbase + 3 < g.out_dim) &&
                        (!g.res || (((g.ldres & 3) == 0) && (((uintptr_t)g.res & 15) == 0)));
#pragma unroll
    for (int a = 0; a < TA; ++a) {
        __syncthreads();                                    // slab (and, first time, the operand tiles) are free
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                slab[((i & 3) + 8 * (i >> 2) + 4 * lk) * CS + b * 32 + li] = acc[a][b][i];
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int rr = er + 4 * p;
            const int64_t r = row0 + wr * (BM / 2) + a * 32 + rr;
            if (r >= g.rows) continue;
            float4 v4 = *reinterpret_cast<const float4*>(slab + rr * CS + ec);
            float v[4] = {v4.x, v4.y, v4.z, v4.w};
            if (vec_ok) {
                float rs[4] = {0.f, 0.f, 0.f, 0.f};
                if (g.res) {
                    const float4 r4 = *reinterpret_cast<const float4*>(g.res + r * g.ldres + jbase);
                    rs[0] = r4.x; rs[1] = r4.y; rs[2] = r4.z; rs[3] = r4.w;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (g.relu && !(v[q] > 0.0f)) v[q] = 0.0f;
                    if (g.res) v[q] = rs[q] + v[q];
                    if (g.bn_gamma) v[q] = fma_(v[q], sc4[q], sh4[q]);
                }
                *reinterpret_cast<float4*>(g.y + r * g.ldy + jbase) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = jbase + q;
                    if (j >= g.out_dim) continue;
                    float x = v[q];
                    if (g.relu && !(x > 0.0f)) x = 0.0f;
                    if (g.res) x = g.res[r * g.ldres + j] + x;
                    if (g.bn_gamma) x = fma_(x, sc4[q], sh4[q]);
                    g.y[r * g.ldy + j] = x;
                }
            }
        }
    }
}

// Thin problems (rows * out_dim small, e.g. the placeholder query or the graph context): one thread per output,
// plain k-ordered chain; W rows stay in L1/L2.
__global__ void k_thin_linear(GemmArgs g)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.rows * g.out_dim) return;
    const int64_t r = idx / g.out_dim;
    const int j = (int)(idx - r * g.out_dim);
    const float* xr = g.x + r * g.ldx;
    float acc = g.bias ? g.bias[j] : 0.0f;
    if (!g.wt) {
        const float* w = g.W + (int64_t)j * g.ldw;
        for (int k = 0; k < g.in_dim; ++k) acc = fma_(xr[k], w[k], acc);
    } else {
        for (int k = 0; k < g.in_dim; ++k) acc = fma_(xr[k], g.W[(int64_t)k * g.ldw + j], acc);
    }
    if (g.relu && !(acc > 0.0f)) acc = 0.0f;
    if (g.res) acc = g.res[r * g.ldres + j] + acc;
    if (g.bn_gamma) { float sc, sh; bn_consts(g, j, sc, sh); acc = fma_(acc, sc, sh); }
    g.y[r * g.ldy + j] = acc;
}

// VALU cross-check: 64x64 tile, 4x4 outputs per thread, sequential k
__global__ __launch_bounds__(256) void k_linear_valu(GemmArgs g)
{
    constexpr int T = 64, KT = 32, KP = KT + 1;
    __shared__ float As[T * KP];
    __shared__ float Bs[T * KP];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * T;
    const int col0 = blockIdx.y * T;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = col0 + tx * 4 + j;
            acc[i][j] = (g.bias && c < g.out_dim) ? g.bias[c] : 0.0f;
        }
    for (int k0 = 0; k0 < g.in_dim; k0 += KT) {
        __syncthreads();
        for (int i = tid; i < T * KT; i += 256) {
            const int rr = i / KT, kk = i - rr * KT;
            const int64_t r = row0 + rr;
            const int k = k0 + kk, j = col0 + rr;
            As[rr * KP + kk] = (r < g.rows && k < g.in_dim) ? g.x[r * g.ldx + k] : 0.0f;
            float wv = 0.0f;
            if (j < g.out_dim && k < g.in_dim) wv = g.wt ? g.W[(int64_t)k * g.ldw + j] : g.W[(int64_t)j * g.ldw + k];
            Bs[rr * KP + kk] = wv;
        }
        __syncthreads();
        const int kmax = min(KT, g.in_dim - k0);
        for (int kk = 0; kk < kmax; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[(ty * 4 + i) * KP + kk];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[(tx * 4 + j) * KP + kk];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fma_(a[i], b[j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = row0 + ty * 4 + i;
            const int c = col0 + tx * 4 + j;
            if (r >= g.rows || c >= g.out_dim) continue;
            float v = acc[i][j];
            if (g.relu && !(v > 0.0f)) v = 0.0f;
            if (g.res) v = g.res[r * g.ldres + c] + v;
            if (g.bn_gamma) { float sc, sh; bn_consts(g, c, sc, sh); v = fma_(v, sc, sh); }
            g.y[r * g.ldy + c] = v;
        }
}

// in_dim <= 4: one thread per output element
__global__ void k_small_linear(GemmArgs g)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.rows * g.out_dim) return;
    const int64_t r = idx / g.out_dim;
    const int j = (int)(idx - r * g.out_dim);
    float acc = g.bias ? g.bias[j] : 0.0f;
    for (int k = 0; k < g.in_dim; ++k) {
        const float w = g.wt ? g.W[(int64_t)k * g.ldw + j] : g.W[(int64_t)j * g.ldw + k];
        acc = fma_(g.x[r * g.ldx + k], w, acc);
    }
    if (g.relu && !(acc > 0.0f)) acc = 0.0f;
    if (g.res) acc = g.res[r * g.ldres + j] + acc;
    if (g.bn_gamma) { float sc, sh; bn_consts(g, j, sc, sh); acc = fma_(acc, sc, sh); }
    g.y[r * g.ldy + j] = acc;
}

// the same, four adjacent outputs per thread (one 16-byte store): the init embeddings write B*M*E floats from 2-4 inputs,
// i.e. the launch is a pure store stream.  in_dim is a template parameter and the epilogue is bias only, so that all loads of
// a thread are issued before the first is consumed (with run-time trip counts they serialise behind s_waitcnt).
template <int K>
__global__ __launch_bounds__(256) void k_small_linear4(GemmArgs g)
{
    const int oq = g.out_dim >> 2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.rows * oq) return;
    const int64_t r = idx / oq;
    const int j = 4 * (int)(idx - r * oq);
    float x[K], w[4][K];
#pragma unroll
    for (int k = 0; k < K; ++k) x[k] = g.x[r * g.ldx + k];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < K; ++k) w[c][k] = g.W[(int64_t)(j + c) * g.ldw + k];
    float4 acc = g.bias ? *reinterpret_cast<const float4*>(g.bias + j) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        acc.x = fma_(x[k], w[0][k], acc.x);
        acc.y = fma_(x[k], w[1][k], acc.y);
        acc.z = fma_(x[k], w[2][k], acc.z);
        acc.w = fma_(x[k], w[3][k], acc.w);
    }
    *reinterpret_cast<float4*>(g.y + r * g.ldy + j) = acc;
}

int launch_linear(const GemmArgs& g0, hipStream_t st)
{
    GemmArgs g = g0;
    const bool force_valu = g_debug[0] != 0;
    if (g.rows <= 0) return 0;
    if (g.in_dim >= 2 && g.in_dim <= 4 && !g.wt && !g.relu && !g.res && !g.bn_gamma && (g.out_dim & 3) == 0 && (g.ldy & 3) == 0 &&
        ((uintptr_t)g.y & 15) == 0 && (!g.bias || ((uintptr_t)g.bias & 15) == 0) && !force_valu) {
        const int64_t n = g.rows * (g.out_dim >> 2);
        const dim3 grid((unsigned)((n + 255) / 256));
        if (g.in_dim == 2) hipLaunchKernelGGL(k_small_linear4<2>, grid, dim3(256), 0, st, g);
        else if (g.in_dim == 3) hipLaunchKernelGGL(k_small_linear4<3>, grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL(k_small_linear4<4>, grid, dim3(256), 0, st, g);
    } else if (g.in_dim <= 4) {
        const int64_t n = g.rows * g.out_dim;
        hipLaunchKernelGGL(k_small_linear, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g);
    } else if (!force_valu && g.rows * g.out_dim <= 16384) {
        const int64_t n = g.rows * g.out_dim;
        hipLaunchKernelGGL(k_thin_linear, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g);
    } else if (force_valu || (g.in_dim % GK) || (g.ldx & 3) || ((uintptr_t)g.x & 15) || (g.ldw & 3) ||
               ((uintptr_t)g.W & 15) || (g.wt && (g.out_dim & 3))) {
        dim3 grid((unsigned)((g.rows + 63) / 64), (unsigned)((g.out_dim + 63) / 64));
        hipLaunchKernelGGL(k_linear_valu, grid, dim3(256), 0, st, g);
    } else {
        const int bm = g_debug[4] ? 128 : 64;
        dim3 grid((unsigned)((g.rows + bm - 1) / bm), (unsigned)((g.out_dim + GT - 1) / GT));
        // compile-time epilogue where the output is whole 128-column tiles of 16-byte aligned rows
        const bool aligned = (g.out_dim % GT == 0) && ((g.ldy & 3) == 0) && (((uintptr_t)g.y & 15) == 0) &&
                             (!g.res || (((g.ldres & 3) == 0) && (((uintptr_t)g.res & 15) == 0)));
        int epi = -1;
        if (aligned && !g_debug[10]) {
            if (g.res && g.bn_gamma && !g.relu) epi = 3;
            else if (g.res && !g.bn_gamma && !g.relu) epi = 2;
            else if (!g.res && !g.bn_gamma) epi = g.relu ? 1 : 0;
        }
        if (bm == 64 && !g.wt) {
            switch (epi) {
            case 0: hipLaunchKernelGGL((k_linear_mfma<false, 64, 0>), grid, dim3(256), 0, st, g); break;
            case 1: hipLaunchKernelGGL((k_linear_mfma<false, 64, 1>), grid, dim3(256), 0, st, g); break;
            case 2: hipLaunchKernelGGL((k_linear_mfma<false, 64, 2>), grid, dim3(256), 0, st, g); break;
            case 3: hipLaunchKernelGGL((k_linear_mfma<false, 64, 3>), grid, dim3(256), 0, st, g); break;
            default: hipLaunchKernelGGL((k_linear_mfma<false, 64, -1>), grid, dim3(256), 0, st, g); break;
            }
        } else if (bm == 64) {
            if (epi == 0) hipLaunchKernelGGL((k_linear_mfma<true, 64, 0>), grid, dim3(256), 0, st, g);
            else hipLaunchKernelGGL((k_linear_mfma<true, 64, -1>), grid, dim3(256), 0, st, g);
        } else {
            if (g.wt) hipLaunchKernelGGL((k_linear_mfma<true, 128, -1>), grid, dim3(256), 0, st, g);
            else hipLaunchKernelGGL((k_linear_mfma<false, 128, -1>), grid, dim3(256), 0, st, g);
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------
// encoder self-attention: block = (instance, head); thread = query row
// ---------------------------------------------------------------------------------------------------
// Z of the attention rows: ZRot (dmath.hpp), the canonical (P0 + P1) + (P2 + P3) order.
template <int D>
__global__ __launch_bounds__(128) void k_mha_encoder(const float* qkv, float* out, int N, int E, int H)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ks = reinterpret_cast<float*>(smem);   // [N][D]
    float* vs = ks + (size_t)N * D;               // [N][D]
    const int64_t b = blockIdx.x / H;
    const int h = blockIdx.x - b * H;
    const float* base = qkv + b * (int64_t)N * 3 * E;
    for (int i = threadIdx.x; i < N * D; i += blockDim.x) {
        const int n = i / D, d = i - n * D;
        ks[i] = base[(int64_t)n * 3 * E + E + h * D + d];
        vs[i] = base[(int64_t)n * 3 * E + 2 * E + h * D + d];
    }
    __syncthreads();
    const float scale = 1.0f / __builtin_sqrtf((float)D);
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        float q[D];
#pragma unroll
        for (int d = 0; d < D; ++d) q[d] = base[(int64_t)i * 3 * E + h * D + d];
        float m = -INFINITY;
        for (int j = 0; j < N; ++j) {
            float acc = 0.0f;
#pragma unroll
            for (int d = 0; d < D; ++d) acc = fma_(q[d], ks[j * D + d], acc);
            m = __builtin_fmaxf(m, acc * scale);
        }
        ZRot<float> zr{0.0f, 0.0f, 0.0f, 0.0f};
        float o[D];
#pragma unroll
        for (int d = 0; d < D; ++d) o[d] = 0.0f;
        for (int j = 0; j < N; ++j) {
            float acc = 0.0f;
#pragma unroll
            for (int d = 0; d < D; ++d) acc = fma_(q[d], ks[j * D + d], acc);
            const float w = d_expf(acc * scale - m);
            zr.add(w);
#pragma unroll
            for (int d = 0; d < D; ++d) o[d] = fma_(w, vs[j * D + d], o[d]);
        }
        const float Z = zr.total(N);
        float* op = out + (b * N + i) * (int64_t)E + h * D;
#pragma unroll
        for (int d = 0; d < D; ++d) op[d] = o[d] / Z;
    }
}

// Register-blocked variant for D = 16: one block = (instance, group of 4 heads), one thread = TWO query rows of one
// head, so every K / V row fetched from LDS feeds two fma chains and the kernel is VALU-bound instead of
// LDS-bound (a broadcast ds_read_b128 still costs 4 LDS cycles).  Same per-row arithmetic as k_mha_encoder.
__global__ __launch_bounds__(256) void k_mha_encoder_x2(const float* __restrict__ qkv, float* __restrict__ out, int N, int E,
                                                        int H)
{
    constexpr int D = 16, HG = 4, W = HG * D;          // 64 floats of K (and of V) per node in LDS
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ks = reinterpret_cast<float*>(smem);        // [N][W]
    float* vs = ks + (size_t)N * W;                    // [N][W]
    const int groups = H / HG;
    const int64_t b = blockIdx.x / groups;
    const int h0 = (blockIdx.x - b * groups) * HG;
    const float* base = qkv + b * (int64_t)N * 3 * E;
    for (int i = threadIdx.x; i < N * (W / 4); i += blockDim.x) {
        const int n = i / (W / 4), c4 = i - n * (W / 4);
        const float* src = base + (int64_t)n * 3 * E + h0 * D + 4 * c4;
        *reinterpret_cast<float4*>(ks + n * W + 4 * c4) = *reinterpret_cast<const float4*>(src + E);
        *reinterpret_cast<float4*>(vs + n * W + 4 * c4) = *reinterpret_cast<const float4*>(src + 2 * E);
    }
    __syncthreads();
    const int P = (N + 1) / 2;                          // row pairs per head
    const int hh = threadIdx.x / P, pr = threadIdx.x - hh * P;
    if (hh >= HG) return;
    const int i0 = pr, i1 = pr + P;                     // rows i0 < P <= i1
    const bool has1 = i1 < N;
    const int i1c = has1 ? i1 : N - 1;
    f32x2 q2[D];                                        // (row i0, row i1) side by side: packed fp32 math
    {
        const float* p0 = base + (int64_t)i0 * 3 * E + (h0 + hh) * D;
        const float* p1 = base + (int64_t)i1c * 3 * E + (h0 + hh) * D;
#pragma unroll
        for (int d = 0; d < D; d += 4) {
            const float4 a = *reinterpret_cast<const float4*>(p0 + d);
            const float4 c = *reinterpret_cast<const float4*>(p1 + d);
            // 1/sqrt(16) folded into q: a power of two, so chain(0.25 q, k) == 0.25 chain(q, k) bit for bit
            q2[d] = (f32x2){a.x, c.x} * 0.25f; q2[d + 1] = (f32x2){a.y, c.y} * 0.25f;
            q2[d + 2] = (f32x2){a.z, c.z} * 0.25f; q2[d + 3] = (f32x2){a.w, c.w} * 0.25f;
        }
    }
    const float* kh = ks + hh * D;
    const float* vh = vs + hh * D;
    f32x2 m2 = splat2(-INFINITY);
    for (int j = 0; j < N; ++j) {
        f32x2 a2 = splat2(0.0f);
#pragma unroll
        for (int d = 0; d < D; d += 4) {
            const float4 kk = *reinterpret_cast<const float4*>(kh + j * W + d);
            a2 = pk_fma(q2[d], splat2(kk.x), a2);
            a2 = pk_fma(q2[d + 1], splat2(kk.y), a2);
            a2 = pk_fma(q2[d + 2], splat2(kk.z), a2);
            a2 = pk_fma(q2[d + 3], splat2(kk.w), a2);
        }
        m2.x = __builtin_fmaxf(m2.x, a2.x);
        m2.y = __builtin_fmaxf(m2.y, a2.y);
    }
    ZRot<f32x2> zr{splat2(0.0f), splat2(0.0f), splat2(0.0f), splat2(0.0f)};
    f32x2 o2[D];
#pragma unroll
    for (int d = 0; d < D; ++d) o2[d] = splat2(0.0f);
    for (int j = 0; j < N; ++j) {
        f32x2 a2 = splat2(0.0f);
#pragma unroll
        for (int d = 0; d < D; d += 4) {
            const float4 kk = *reinterpret_cast<const float4*>(kh + j * W + d);
            a2 = pk_fma(q2[d], splat2(kk.x), a2);
            a2 = pk_fma(q2[d + 1], splat2(kk.y), a2);
            a2 = pk_fma(q2[d + 2], splat2(kk.z), a2);
            a2 = pk_fma(q2[d + 3], splat2(kk.w), a2);
        }
        const f32x2 w2 = d_expf2_nonpos(a2 - m2);
        zr.add(w2);
#pragma unroll
        for (int d = 0; d < D; d += 4) {
            const float4 vv = *reinterpret_cast<const float4*>(vh + j * W + d);
            o2[d] = pk_fma(w2, splat2(vv.x), o2[d]);
            o2[d + 1] = pk_fma(w2, splat2(vv.y), o2[d + 1]);
            o2[d + 2] = pk_fma(w2, splat2(vv.z), o2[d + 2]);
            o2[d + 3] = pk_fma(w2, splat2(vv.w), o2[d + 3]);
        }
    }
    float o0[D], o1[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { o0[d] = o2[d].x; o1[d] = o2[d].y; }
    const f32x2 Z2 = zr.total(N);
    const float Z0 = Z2.x, Z1 = Z2.y;
    float* op0 = out + (b * N + i0) * (int64_t)E + (h0 + hh) * D;
#pragma unroll
    for (int d = 0; d < D; d += 4)
        *reinterpret_cast<float4*>(op0 + d) = make_float4(o0[d] / Z0, o0[d + 1] / Z0, o0[d + 2] / Z0, o0[d + 3] / Z0);
    if (has1) {
        float* op1 = out + (b * N + i1) * (int64_t)E + (h0 + hh) * D;
#pragma unroll
        for (int d = 0; d < D; d += 4)
            *reinterpret_cast<float4*>(op1 + d) = make_float4(o1[d] / Z1, o1[d + 1] / Z1, o1[d + 2] / Z1, o1[d + 3] / Z1);
    }
}

// Large graphs (N > 127): the blocked kernel with the keys tiled through LDS.  One block = (instance, group of 4
// heads, block of 128 query rows); a thread owns the query rows (i0, i0 + 64) of one head.  Pass 1 walks all key tiles
// for the row maxima, pass 2 walks them again in ascending order for exp / Z / PV, so every sum keeps the canonical
// ascending-key order of k_mha_encoder (accumulators simply carry across tiles).
constexpr int MHA_TK = 128;     // keys per LDS tile
__global__ __launch_bounds__(256) void k_mha_encoder_tiled(const float* __restrict__ qkv, float* __restrict__ out, int N, int E,
                                                           int H)
{
    constexpr int D = 16, HG = 4, W = HG * D;
    __shared__ __attribute__((aligned(16))) float ks[MHA_TK * W];
    __shared__ __attribute__((aligned(16))) float vs[MHA_TK * W];
    const int groups = H / HG;
    const int qblocks = (N + 127) / 128;
    const int qb = blockIdx.x % qblocks;
    const int64_t bg = blockIdx.x / qblocks;
    const int64_t b = bg / groups;
    const int h0 = (int)(bg - b * groups) * HG;
    const float* base = qkv + b * (int64_t)N * 3 * E;
    const int hh = threadIdx.x >> 6, pr = threadIdx.x & 63;
    const int i0 = qb * 128 + pr, i1 = i0 + 64;
    const bool has0 = i0 < N, has1 = i1 < N;
    const int i0c = has0 ? i0 : N - 1, i1c = has1 ? i1 : N - 1;
    f32x2 q2[D];
    {
        const float* p0 = base + (int64_t)i0c * 3 * E + (h0 + hh) * D;
        const float* p1 = base + (int64_t)i1c * 3 * E + (h0 + hh) * D;
#pragma unroll
        for (int d = 0; d < D; d += 4) {
            const float4 a = *reinterpret_cast<const float4*>(p0 + d);
            const float4 c = *reinterpret_cast<const float4*>(p1 + d);
            q2[d] = (f32x2){a.x, c.x} * 0.25f; q2[d + 1] = (f32x2){a.y, c.y} * 0.25f;
            q2[d + 2] = (f32x2){a.z, c.z} * 0.25f; q2[d + 3] = (f32x2){a.w, c.w} * 0.25f;
        }
    }
    const float* kh = ks + hh * D;
    const float* vh = vs + hh * D;
    f32x2 m2 = splat2(-INFINITY);
    for (int j0 = 0; j0 < N; j0 += MHA_TK) {
        const int nj = N - j0 < MHA_TK ? N - j0 : MHA_TK;
        __syncthreads();
        for (int i = threadIdx.x; i < nj * (W / 4); i += 256) {
            const int n = i / (W / 4), c4 = i - n * (W / 4);
            *reinterpret_cast<float4*>(ks + n * W + 4 * c4) =
                *reinterpret_cast<const float4*>(base + (int64_t)(j0 + n) * 3 * E + E + h0 * D + 4 * c4);
        }
        __syncthreads();
        for (int j = 0; j < nj; ++j) {
            f32x2 a2 = splat2(0.0f);
#pragma unroll
            for (int d = 0; d < D; d += 4) {
                const float4 kk = *reinterpret_cast<const float4*>(kh + j * W + d);
                a2 = pk_fma(q2[d], splat2(kk.x), a2);
                a2 = pk_fma(q2[d + 1], splat2(kk.y), a2);
                a2 = pk_fma(q2[d + 2], splat2(kk.z), a2);
                a2 = pk_fma(q2[d + 3], splat2(kk.w), a2);
            }
            m2.x = __builtin_fmaxf(m2.x, a2.x);
            m2.y = __builtin_fmaxf(m2.y, a2.y);
        }
    }
    ZRot<f32x2> zr{splat2(0.0f), splat2(0.0f), splat2(0.0f), splat2(0.0f)};
    f32x2 o2[D];
#pragma unroll
    for (int d = 0; d < D; ++d) o2[d] = splat2(0.0f);
    for (int j0 = 0; j0 < N; j0 += MHA_TK) {
        const int nj = N - j0 < MHA_TK ? N - j0 : MHA_TK;
        __syncthreads();
        for (int i = threadIdx.x; i < nj * (W / 4); i += 256) {
            const int n = i / (W / 4), c4 = i - n * (W / 4);
            const float* src = base + (int64_t)(j0 + n) * 3 * E + E + h0 * D + 4 * c4;
            *reinterpret_cast<float4*>(ks + n * W + 4 * c4) = *reinterpret_cast<const float4*>(src);
            *reinterpret_cast<float4*>(vs + n * W + 4 * c4) = *reinterpret_cast<const float4*>(src + E);
        }
        __syncthreads();
        for (int j = 0; j < nj; ++j) {
            f32x2 a2 = splat2(0.0f);
#pragma unroll
            for (int d = 0; d < D; d += 4) {
                const float4 kk = *reinterpret_cast<const float4*>(kh + j * W + d);
                a2 = pk_fma(q2[d], splat2(kk.x), a2);
                a2 = pk_fma(q2[d + 1], splat2(kk.y), a2);
                a2 = pk_fma(q2[d + 2], splat2(kk.z), a2);
                a2 = pk_fma(q2[d + 3], splat2(kk.w), a2);
            }
            const f32x2 w2 = d_expf2_nonpos(a2 - m2);
            zr.add(w2);
#pragma unroll
            for (int d = 0; d < D; d += 4) {
                const float4 vv = *reinterpret_cast<const float4*>(vh + j * W + d);
                o2[d] = pk_fma(w2, splat2(vv.x), o2[d]);
                o2[d + 1] = pk_fma(w2, splat2(vv.y), o2[d + 1]);
                o2[d + 2] = pk_fma(w2, splat2(vv.z), o2[d + 2]);
                o2[d + 3] = pk_fma(w2, splat2(vv.w), o2[d + 3]);
            }
        }
    }
    const f32x2 Z2 = zr.total(N);
    if (has0) {
        float* op0 = out + (b * N + i0) * (int64_t)E + (h0 + hh) * D;
#pragma unroll
        for (int d = 0; d < D; d += 4)
            *reinterpret_cast<float4*>(op0 + d) =
                make_float4(o2[d].x / Z2.x, o2[d + 1].x / Z2.x, o2[d + 2].x / Z2.x, o2[d + 3].x / Z2.x);
    }
    if (has1) {
        float* op1 = out + (b * N + i1) * (int64_t)E + (h0 + hh) * D;
#pragma unroll
        for (int d = 0; d < D; d += 4)
            *reinterpret_cast<float4*>(op1 + d) =
                make_float4(o2[d].y / Z2.y, o2[d + 1].y / Z2.y, o2[d + 2].y / Z2.y, o2[d + 3].y / Z2.y);
    }
}

int launch_mha_encoder(const float* qkv, float* out, int64_t B, int N, int E, int H, hipStream_t st)
{
    const int D = E / H;
    const size_t lds = (size_t)2 * N * D * sizeof(float);
    if (D * H != E || lds > 160 * 1024 || B * H > 0x7fffffffLL) return EAMRL_E_ARG;
    dim3 grid((unsigned)(B * H)), block(128);
    // blocked kernel: 4 heads per block, 2 query rows per thread (needs 4 * ceil(N/2) <= 256 threads, aligned rows)
    if (D == 16 && H % 4 == 0 && 2 * (N + 1) <= 256 && !g_debug[3] && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0) {
        const size_t lds2 = (size_t)2 * N * 64 * sizeof(float);
        hipLaunchKernelGGL(k_mha_encoder_x2, dim3((unsigned)(B * (H / 4))), dim3(256), lds2, st, qkv, out, N, E, H);
        return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
    }
    // large graphs: the key-tiled fp32-MFMA kernel (encoder_attn_mfma.hip); debug key 15 keeps the VALU one for A/B runs
    if (!g_debug[15] && !g_debug[3] && mha_encoder_mfma_supports(N, E, H) && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0)
        return launch_mha_encoder_mfma(qkv, out, B, N, E, H, st);
    if (D == 16 && H % 4 == 0 && !g_debug[3] && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0 &&
        B * (H / 4) * ((N + 127) / 128) <= 0x7fffffffLL) {              // large graphs: keys tiled through LDS
        const unsigned nb = (unsigned)(B * (H / 4) * ((N + 127) / 128));
        hipLaunchKernelGGL(k_mha_encoder_tiled, dim3(nb), dim3(256), 0, st, qkv, out, N, E, H);
        return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
    }
    if (D == 16) {
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(k_mha_encoder<16>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return EAMRL_E_LAUNCH;
        hipLaunchKernelGGL(k_mha_encoder<16>, grid, block, lds, st, qkv, out, N, E, H);
    } else if (D == 32) {
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(k_mha_encoder<32>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return EAMRL_E_LAUNCH;
        hipLaunchKernelGGL(k_mha_encoder<32>, grid, block, lds, st, qkv, out, N, E, H);
    } else {
        return EAMRL_E_ARG;
    }
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------
// normalisation + mean
// ---------------------------------------------------------------------------------------------------
// BatchNorm1d eval: scale = gamma / sqrt(var + eps); shift = beta - mean * scale; y = fma(x, scale, shift)
__global__ __launch_bounds__(256) void k_norm_batch_eval(float* x, int64_t rows, int E, const float* gamma,
                                                         const float* beta, const float* mean, const float* var, float eps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* scale = reinterpret_cast<float*>(smem);
    float* shift = scale + E;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const float s = gamma[e] / __builtin_sqrtf(var[e] + eps);
        const float ms = mean[e] * s;
        scale[e] = s;
        shift[e] = beta[e] - ms;
    }
    __syncthreads();
    const int64_t total = rows * E;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i % E);
        x[i] = fma_(x[i], scale[e], shift[e]);
    }
}

// InstanceNorm1d(affine): block = instance, thread = channel; sequential over nodes (coalesced across channels)
__global__ void k_norm_instance(float* x, int N, int E, const float* gamma, const float* beta, float eps)
{
    float* xb = x + (int64_t)blockIdx.x * N * E;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        float s = 0.0f;
        for (int n = 0; n < N; ++n) s = s + xb[(int64_t)n * E + e];
        const float mean = s / (float)N;
        float v = 0.0f;
        for (int n = 0; n < N; ++n) { const float d = xb[(int64_t)n * E + e] - mean; v = fma_(d, d, v); }
        const float inv = 1.0f / __builtin_sqrtf(v / (float)N + eps);
        const float g = gamma[e], bt = beta[e];
        for (int n = 0; n < N; ++n) {
            const float d = xb[(int64_t)n * E + e] - mean;
            xb[(int64_t)n * E + e] = fma_(d * inv, g, bt);
        }
    }
}

__global__ void k_mean_nodes(const float* emb, float* out, int M, int E)
{
    const float* xb = emb + (int64_t)blockIdx.x * M * E;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        float s = 0.0f;
        int n = 0;
        for (; n + 8 <= M; n += 8) {            // 8 loads in flight, added in node order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = xb[(int64_t)(n + u) * E + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s = s + v[u];
        }
        for (; n < M; ++n) s = s + xb[(int64_t)n * E + e];
        out[(int64_t)blockIdx.x * E + e] = s / (float)M;
    }
}

// BatchNorm1d with BATCH statistics (policy.train(), nn/ops.py:45-47): per-channel mean and biased variance over all
// rows in a defined order -- chunks of BN_CHUNK rows summed sequentially (thread = channel, coalesced across channels),
// chunk sums added in ascending order; variance as the mean of fma(d, d, .) with d = x - mean (second pass).
constexpr int BN_CHUNK = 128;

__global__ void k_bn_partial(const float* x, int64_t rows, int E, const float* mean, float* ws)
{
    const int64_t r0 = (int64_t)blockIdx.x * BN_CHUNK;
    const int64_t r1 = r0 + BN_CHUNK < rows ? r0 + BN_CHUNK : rows;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        float s = 0.0f;
        if (!mean) {
            for (int64_t r = r0; r < r1; ++r) s = s + x[r * E + e];
        } else {
            const float m = mean[e];
            for (int64_t r = r0; r < r1; ++r) { const float d = x[r * E + e] - m; s = fma_(d, d, s); }
        }
        ws[(int64_t)blockIdx.x * E + e] = s;
    }
}

// pass 0: mean -> save_mean; pass 1: biased variance -> save_var, and the running statistics as torch updates them
// (running = (1 - momentum) * running + momentum * stat, the variance one unbiased: sum / (n - 1)).
__global__ void k_bn_final(const float* ws, int nchunks, int E, int64_t rows, int pass, float* save_mean, float* save_var,
                           float* running_mean, float* running_var, float momentum)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    float s = 0.0f;
    for (int c = 0; c < nchunks; ++c) s = s + ws[(int64_t)c * E + e];
    const float n = (float)rows;
    if (pass == 0) {
        save_mean[e] = s / n;
    } else {
        save_var[e] = s / n;
        if (running_mean) {
            const float keep = 1.0f - momentum;
            const float a = keep * running_mean[e], b = momentum * save_mean[e];
            running_mean[e] = a + b;
            const float unb = rows > 1 ? s / (n - 1.0f) : s / n;
            const float c = keep * running_var[e], d = momentum * unb;
            running_var[e] = c + d;
        }
    }
}

int launch_batchnorm_train(float* x, int64_t rows, int E, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float momentum, float eps, float* save_mean, float* save_var, float* ws,
                           hipStream_t st)
{
    if (rows <= 0) return 0;
    const int nchunks = (int)((rows + BN_CHUNK - 1) / BN_CHUNK);
    const int thr = E <= 1024 ? ((E + 63) / 64) * 64 : 1024;
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(k_bn_partial, dim3((unsigned)nchunks), dim3(thr), 0, st, x, rows, E,
                           pass ? (const float*)save_mean : (const float*)nullptr, ws);
        hipLaunchKernelGGL(k_bn_final, dim3((unsigned)((E + 127) / 128)), dim3(128), 0, st, ws, nchunks, E, rows, pass,
                           save_mean, save_var, running_mean, running_var, momentum);
    }
    const int64_t total = rows * E;
    const int64_t want = (total + 255) / 256;
    const unsigned blocks = (unsigned)(want < 2048 ? want : 2048);
    hipLaunchKernelGGL(k_norm_batch_eval, dim3(blocks), dim3(256), 2 * E * sizeof(float), st, x, rows, E, gamma, beta,
                       save_mean, save_var, eps);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------------
// InstanceNorm1d(affine) for the TRAINING graph (the differentiable encoder of train.py): out-of-place forward that keeps
// mean / 1/std per (instance, channel), and the backward.  Thread = channel, sequential over the instance's nodes, rows read
// as 4 E-byte coalesced runs -- on the [B][N][E] layout as it is (torch's instance_norm wants [B][E][N]: two transposed copies
// per call, and MIOpen's spatial batch-norm backward behind it takes 0.49 ms per call at 1024 x 100 x 128).
// ---------------------------------------------------------------------------------------------------------------------
// Forward: the SAME arithmetic as k_norm_instance, bit for bit (sum and variance sequentially over the nodes in node order,
// variance as the fma chain of (x - mean)^2, y = fma((x - mean) * rstd, gamma, beta)): with it the training graph's encoder
// reproduces the rollout's embeddings exactly, so that one encoder pass can serve both (policy.forward, phase "train").
// The instance's [N][E] tile is staged in LDS by all threads (coalesced float4 loads), then thread = channel walks its column
// (LDS reads pipeline; the global-memory version of the same walk waits a full memory latency per 8 rows).
__global__ __launch_bounds__(256) void k_instnorm_train_fwd(const float* __restrict__ x, float* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out, int N, int E,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, int in_lds)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];      // [N][E] when in_lds, then mean [E] | rstd [E]
    const int64_t b = blockIdx.x;
    const float* xb = x + b * (int64_t)N * E;
    float* yb = y + b * (int64_t)N * E;
    const int total = N * E;
    if (in_lds) {
        if ((E & 3) == 0) {
            for (int i = threadIdx.x; i < total / 4; i += blockDim.x)
                reinterpret_cast<float4*>(tile)[i] = reinterpret_cast<const float4*>(xb)[i];
        } else {
            for (int i = threadIdx.x; i < total; i += blockDim.x) tile[i] = xb[i];
        }
        __syncthreads();
    }
    const float* src = in_lds ? tile : xb;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        float s = 0.0f;
        for (int n = 0; n < N; ++n) s = s + src[n * E + e];
        const float mean = s / (float)N;
        float v = 0.0f;
        for (int n = 0; n < N; ++n) { const float d = src[n * E + e] - mean; v = fma_(d, d, v); }
        const float rstd = 1.0f / __builtin_sqrtf(v / (float)N + eps);
        mean_out[b * E + e] = mean;
        rstd_out[b * E + e] = rstd;
        if (in_lds) {
            tile[total + e] = mean;
            tile[total + E + e] = rstd;
        } else {
            const float g = gamma ? gamma[e] : 1.0f, bt = beta ? beta[e] : 0.0f;
            for (int n = 0; n < N; ++n) yb[(int64_t)n * E + e] = fma_((src[n * E + e] - mean) * rstd, g, bt);
        }
    }
    if (in_lds) {                   // the elementwise pass with all threads (coalesced stores)
        __syncthreads();
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            const int e = i % E;
            const float g = gamma ? gamma[e] : 1.0f, bt = beta ? beta[e] : 0.0f;
            yb[i] = fma_((tile[i] - tile[total + e]) * tile[total + E + e], g, bt);
        }
    }
}

// Backward: block = 4 row groups x E channels (E <= 128 here; larger E loops): thread (q, e) takes the rows n = q (mod 4); the
// four partial sums meet in LDS.  4x the loads in flight of a thread-per-channel loop.  (Not parity-critical.)
constexpr int IN_Q = 4;

// dx = gamma rstd (dy - mean_n(dy) - xhat mean_n(dy xhat));  dgamma += sum_n dy xhat;  dbeta += sum_n dy
__global__ void k_instnorm_train_bwd(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean_in,
                                     const float* __restrict__ rstd_in, const float* __restrict__ gamma, float* __restrict__ dx,
                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int E)
{
    extern __shared__ float red[];                    // [2][IN_Q][Ep]
    const int Ep = blockDim.x / IN_Q;
    const int q = threadIdx.x / Ep, el = threadIdx.x - q * Ep;
    const int64_t b = blockIdx.x;
    const float* xb = x + b * (int64_t)N * E;
    const float* gb = dy + b * (int64_t)N * E;
    float* db = dx + b * (int64_t)N * E;
    float* red2 = red + IN_Q * Ep;
    for (int e0 = 0; e0 < E; e0 += Ep) {
        const int e = e0 + el;
        const bool on = e < E;
        const float mean = on ? mean_in[b * E + e] : 0.0f, rstd = on ? rstd_in[b * E + e] : 0.0f;
        float s1 = 0.0f, s2 = 0.0f;
        if (on)
            for (int n = q; n < N; n += IN_Q) {
                const float g = gb[(int64_t)n * E + e];
                const float xh = (xb[(int64_t)n * E + e] - mean) * rstd;
                s1 += g;
                s2 = fmaf(g, xh, s2);
            }
        red[q * Ep + el] = s1;
        red2[q * Ep + el] = s2;
        __syncthreads();
        s1 = ((red[el] + red[Ep + el]) + red[2 * Ep + el]) + red[3 * Ep + el];
        s2 = ((red2[el] + red2[Ep + el]) + red2[2 * Ep + el]) + red2[3 * Ep + el];
        __syncthreads();
        if (on) {
            const float m1 = s1 / (float)N, m2 = s2 / (float)N;
            const float gr = (gamma ? gamma[e] : 1.0f) * rstd;
            for (int n = q; n < N; n += IN_Q) {
                const float xh = (xb[(int64_t)n * E + e] - mean) * rstd;
                db[(int64_t)n * E + e] = gr * ((gb[(int64_t)n * E + e] - m1) - xh * m2);
            }
            if (q == 0) {
                if (dgamma) atomicAdd(dgamma + e, s2);
                if (dbeta) atomicAdd(dbeta + e, s1);
            }
        }
    }
}

int launch_instnorm_train_fwd(const float* x, float* y, float* mean, float* rstd, int64_t B, int N, int E, const float* gamma,
                              const float* beta, float eps, hipStream_t st)
{
    const size_t bytes = ((size_t)N * E + 2 * (size_t)E) * sizeof(float);
    const int in_lds = bytes <= 96 * 1024;
    auto k = k_instnorm_train_fwd;
    if (in_lds && bytes > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        return EAMRL_E_LAUNCH;
    hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(256), in_lds ? bytes : 0, st, x, y, mean, rstd, N, E, gamma, beta, eps, in_lds);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_instnorm_train_bwd(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma, float* dx,
                              float* dgamma, float* dbeta, int64_t B, int N, int E, hipStream_t st)
{
    const int Ep = E <= 128 ? ((E + 63) / 64) * 64 : 128;
    hipLaunchKernelGGL(k_instnorm_train_bwd, dim3((unsigned)B), dim3(IN_Q * Ep), 2 * IN_Q * Ep * sizeof(float), st, x, dy, mean, rstd,
                       gamma, dx, dgamma, dbeta, N, E);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_normalize(float* x, int64_t B, int N, int E, int kind, const float* gamma, const float* beta,
                     const float* mean, const float* var, float eps, hipStream_t st)
{
    if (B <= 0) return 0;
    if (kind == EAMRL_NORM_BATCH_EVAL) {
        if (!mean || !var) return EAMRL_E_ARG;
        const int64_t total = B * N * E;
        const int64_t want = (total + 255) / 256;
        const unsigned blocks = (unsigned)(want < 2048 ? want : 2048);
        hipLaunchKernelGGL(k_norm_batch_eval, dim3(blocks), dim3(256), 2 * E * sizeof(float), st, x, B * N, E, gamma,
                           beta, mean, var, eps);
    } else if (kind == EAMRL_NORM_INSTANCE) {
        hipLaunchKernelGGL(k_norm_instance, dim3((unsigned)B), dim3(E <= 1024 ? ((E + 63) / 64) * 64 : 1024), 0, st, x,
                           N, E, gamma, beta, eps);
    } else {
        return EAMRL_E_ARG;
    }
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_mean_nodes(const float* emb, float* out, int64_t B, int M, int E, hipStream_t st)
{
    if (B <= 0) return 0;
    hipLaunchKernelGGL(k_mean_nodes, dim3((unsigned)B), dim3(E <= 1024 ? ((E + 63) / 64) * 64 : 1024), 0, st, emb, out,
                       M, E);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

}  // namespace eamrl
