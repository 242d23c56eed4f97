// Weight gradient of torch.nn.Linear for the training graph:  dW[o][i] = sum_r dY[r][o] X[r][i],  db[o] = sum_r dY[r][o].
//
// Reference: the Linears of the encoder and of the decoder's cache projections (rl4co/models/nn/attention.py:112-136,
// nn/mlp.py:52-61, zoo/am/decoder.py:206-235) as differentiated by the REINFORCE / POMO / EAM trainers' loss.backward()
// (models/rl/reinforce/reinforce.py:62-64,103-106, zoo/pomo/model.py:103-112, zoo/earl/model.py:179-195).  Not part of the
// bit-exact rollout path: summation order is the tile / chunk order, results are held to 1e-5 of torch's (tests/test_gpu_train.py).
//
// The contraction runs over the ROWS (B * N = 102,400 at the POMO training size) and both operands are row-major, which is
// exactly the operand layout of v_mfma_f32_32x32x2_f32: lane l of the A operand holds dY[r + (l >> 5)][o0 + (l & 31)], of the
// B operand X[r + (l >> 5)][i0 + (l & 31)] -- two rows per MFMA, no transposition anywhere.  A workgroup of four wavefronts
// owns a 128 x 128 block of dW over a chunk of rows: 16-row slabs of both operands go through LDS (double-buffered, one
// barrier per slab, the next slab's global loads in flight behind the 32 MFMAs of the current one), each wavefront
// accumulates a 64 x 64 quarter (four 32 x 32 tiles, 64 registers).  Chunks write partial blocks to a scratch that
// k_wgrad_reduce sums (fixed order: the result does not depend on scheduling).
#include "kernels.hpp"

namespace eamrl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int WT = 128;      // block of dW per workgroup (WT x WT)
constexpr int WK = 16;       // rows per slab

__global__ __launch_bounds__(256, 2) void k_linear_wgrad(const float* __restrict__ dy, int64_t ldy, const float* __restrict__ x,
                                                         int64_t ldx, int64_t rows, int out_dim, int in_dim,
                                                         int64_t rows_per_chunk, float* __restrict__ part)
{
    __shared__ __attribute__((aligned(16))) float As[2][WK][WT];
    __shared__ __attribute__((aligned(16))) float Bs[2][WK][WT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int oh = wv & 1, ih = wv >> 1;
    const int o0 = blockIdx.x * WT, i0 = blockIdx.y * WT;
    const int64_t r_begin = (int64_t)blockIdx.z * rows_per_chunk;
    const int64_t r_end = r_begin + rows_per_chunk < rows ? r_begin + rows_per_chunk : rows;
    const int nslab = (int)((r_end - r_begin + WK - 1) / WK);
    // staging role: rows lr and lr + 8 of the slab, columns 4 c4 .. 4 c4 + 3 of both operands
    const int lr = tid >> 5, c4 = tid & 31;
    const float* dyp = dy + o0 + 4 * c4;
    const float* xp = x + i0 + 4 * c4;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float bs0 = 0.0f, bs1 = 0.0f;

    float4 pa[2], pb[2];
    auto fetch = [&](int slab) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t r = r_begin + (int64_t)slab * WK + lr + 8 * h;
            const int64_t rc = r < r_end ? r : r_begin;          // (a valid row; masked below)
            const float4 va = *reinterpret_cast<const float4*>(dyp + rc * ldy);
            const float4 vb = *reinterpret_cast<const float4*>(xp + rc * ldx);
            const bool ok = r < r_end;
            pa[h] = ok ? va : make_float4(0.f, 0.f, 0.f, 0.f);
            pb[h] = ok ? vb : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (nslab > 0) fetch(0);
    for (int s = 0; s < nslab; ++s) {
        const int cur = s & 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<float4*>(&As[cur][lr + 8 * h][4 * c4]) = pa[h];
            *reinterpret_cast<float4*>(&Bs[cur][lr + 8 * h][4 * c4]) = pb[h];
        }
        __syncthreads();
        if (s + 1 < nslab) fetch(s + 1);
        const float* ap = &As[cur][lane >> 5][oh * 64 + (lane & 31)];
        const float* bp = &Bs[cur][lane >> 5][ih * 64 + (lane & 31)];
#pragma unroll
        for (int k = 0; k < WK / 2; ++k) {
            const float a0 = ap[2 * k * WT], a1 = ap[2 * k * WT + 32];
            const float b0 = bp[2 * k * WT], b1 = bp[2 * k * WT + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            bs0 += a0;
            bs1 += a1;
        }
    }
    // partial block of this chunk: tile (a, b) register r of lane l -> row 8 (r >> 2) + 4 (l >> 5) + (r & 3), column l & 31
    const int64_t pstride = (int64_t)out_dim * in_dim + out_dim;         // per chunk: the block of dW, then db
    float* pp = part + blockIdx.z * pstride + (int64_t)(o0 + oh * 64) * in_dim + i0 + ih * 64;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * a + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                pp[(int64_t)row * in_dim + 32 * b + (lane & 31)] = acc[a][b][r];
            }
    if (blockIdx.y == 0 && ih == 0) {
        bs0 += __shfl_xor(bs0, 32);
        bs1 += __shfl_xor(bs1, 32);
        if (lane < 32) {
            float* bq = part + blockIdx.z * pstride + (int64_t)out_dim * in_dim + o0 + oh * 64 + lane;
            bq[0] = bs0;
            bq[32] = bs1;
        }
    }
}

// out[e] = sum_c part[c][e]: 64 float4 elements per workgroup, the chunks dealt to four thread groups (four loads in
// flight each), partial sums combined through LDS in a fixed order.  Elements >= nw4 belong to the bias gradient.
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dW, float* __restrict__ db,
                                                      int64_t nw4, int64_t n4, int nchunks)
{
    __shared__ float4 red[4][64];
    const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + e;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
        const float4* p = reinterpret_cast<const float4*>(part) + i;
        int c = g;
        for (; c + 12 < nchunks; c += 16) {
            const float4 v0 = p[(int64_t)c * n4], v1 = p[(int64_t)(c + 4) * n4], v2 = p[(int64_t)(c + 8) * n4],
                         v3 = p[(int64_t)(c + 12) * n4];
            s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
            s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; c < nchunks; c += 4) {
            const float4 v = p[(int64_t)c * n4];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[g][e] = s;
    __syncthreads();
    if (g == 0 && i < n4) {
        const float4 a = red[0][e], b = red[1][e], c = red[2][e], d = red[3][e];
        const float4 r = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z),
                                     (a.w + b.w) + (c.w + d.w));
        if (i < nw4) reinterpret_cast<float4*>(dW)[i] = r;
        else if (db) reinterpret_cast<float4*>(db)[i - nw4] = r;
    }
}

}  // namespace

bool linear_wgrad_supports(int out_dim, int in_dim) { return out_dim > 0 && in_dim > 0 && out_dim % WT == 0 && in_dim % WT == 0; }

// chunks over the rows: enough workgroups for two per CU, slabs of 16 rows
int linear_wgrad_chunks(int64_t rows, int out_dim, int in_dim)
{
    const int blocks = (out_dim / WT) * (in_dim / WT);
    int64_t n = (512 + blocks - 1) / blocks;
    const int64_t most = (rows + 4 * WK - 1) / (4 * WK);        // at least four slabs per chunk
    if (n > most) n = most;
    return (int)(n < 1 ? 1 : n);
}

int64_t linear_wgrad_scratch(int64_t rows, int out_dim, int in_dim)
{
    return (int64_t)linear_wgrad_chunks(rows, out_dim, in_dim) * ((int64_t)out_dim * in_dim + out_dim);
}

int launch_linear_wgrad(const float* dy, int64_t ldy, const float* x, int64_t ldx, int64_t rows, int out_dim, int in_dim,
                        float* dW, float* db, float* scratch, hipStream_t st)
{
    const int nch = linear_wgrad_chunks(rows, out_dim, in_dim);
    int64_t rpc = (rows + nch - 1) / nch;
    rpc = (rpc + WK - 1) / WK * WK;
    hipLaunchKernelGGL(k_linear_wgrad, dim3(out_dim / WT, in_dim / WT, nch), dim3(256), 0, st, dy, ldy, x, ldx, rows, out_dim,
                       in_dim, rpc, scratch);
    const int64_t nw4 = (int64_t)out_dim * in_dim / 4, n4 = nw4 + out_dim / 4;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, st, scratch, dW, db, nw4, n4, nch);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

}  // namespace eamrl
