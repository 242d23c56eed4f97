// PointerAttention.forward as ONE launch behind the reference's constructor-injection point
// `AttentionModelDecoder(pointer=<nn.Module>)`  [rl4co/models/zoo/am/decoder.py:82,109-124; rl4co/models/nn/attention.py:224-328]:
//   heads = MHA(query, key, value, mask) without projections; glimpse = project_out(heads);
//   logits = glimpse . logit_key^T / sqrt(E)
// for query [B][L][E] against key / value / logit_key [B][M][E].  This is the compatibility path (the reference's own
// decoder loop calling into HIP once per step); the rollout kernels fuse the same stage with folded weights.
//
// Defined order (mirrored by orc_pointer_attention):
//   score[h][n] = chain_d(q, K[n]) * (1/sqrtf(D)), -inf where masked (mask_inner)
//   per head: m = max_n; w[n] = d_expf(score - m) (0 where masked);
//             lane l of a wavefront owns the nodes n = l, l+64, ... ascending: z_l = sum w[n], a_l[d] = chain_n(w[n], V[n][d]);
//             Z = lane_tree(z), heads[d] = lane_tree(a[d]) / Z
//   glimpse[o] = chain_i(heads[i], Wout[o][i], init bias[o] or 0);  logits[n] = chain_o(glimpse[o], Lk[n][o]) / sqrtf(E)
#include "kernels.hpp"

namespace eamrl {

__global__ __launch_bounds__(256) void k_pointer_attention(const float* __restrict__ q, const float* __restrict__ K,
                                                           const float* __restrict__ V, const float* __restrict__ Lk,
                                                           int64_t ld, const uint8_t* __restrict__ mask, int mask_per_query,
                                                           const float* __restrict__ Wout, const float* __restrict__ bout,
                                                           float* __restrict__ logits, int L, int M, int E, int H,
                                                           int mask_inner)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sq = reinterpret_cast<float*>(smem);   // [E]
    float* sh = sq + E;                            // [E] heads
    float* sg = sh + E;                            // [E] glimpse
    float* sw = sg + E;                            // [H][M]
    const int64_t row = blockIdx.x;                // b * L + l
    const int64_t b = row / L;
    const int D = E / H;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* Kb = K + b * (int64_t)M * ld;
    const float* Vb = V + b * (int64_t)M * ld;
    const float* Lb = Lk + b * (int64_t)M * ld;
    const uint8_t* mrow = mask ? mask + (mask_per_query ? row : b) * (int64_t)M : nullptr;
    for (int e = tid; e < E; e += blockDim.x) sq[e] = q[row * E + e];
    __syncthreads();
    const float scale = 1.0f / __builtin_sqrtf((float)D);
    for (int idx = tid; idx < H * M; idx += blockDim.x) {
        const int h = idx % H, n = idx / H;
        const float* kr = Kb + (int64_t)n * ld + h * D;
        float acc = 0.0f;
        for (int d = 0; d < D; ++d) acc = fma_(sq[h * D + d], kr[d], acc);
        acc = acc * scale;
        if (mask_inner && mrow && !mrow[n]) acc = -INFINITY;
        sw[h * M + n] = acc;
    }
    __syncthreads();
    for (int h = wv; h < H; h += 4) {              // one wavefront per head
        float m = -INFINITY;
        for (int n = lane; n < M; n += 64) m = vmax_raw(m, sw[h * M + n]);
        m = wave_max(m);
        float z = 0.0f;
        float a[32];
#pragma unroll
        for (int d = 0; d < 32; ++d) a[d] = 0.0f;
        for (int n = lane; n < M; n += 64) {
            const float s = sw[h * M + n];
            const float w = (s == -INFINITY) ? 0.0f : d_expf(s - m);
            z = z + w;
            const float* vr = Vb + (int64_t)n * ld + h * D;
#pragma unroll
            for (int d = 0; d < 32; ++d)
                if (d < D) a[d] = fma_(w, vr[d], a[d]);
        }
        const float Z = wave_tree_sum(z);
#pragma unroll
        for (int d = 0; d < 32; ++d)
            if (d < D) {
                const float t = wave_tree_sum(a[d]);
                if (lane == 0) sh[h * D + d] = t / Z;
            }
    }
    __syncthreads();
    for (int o = tid; o < E; o += blockDim.x) {
        const float* wr = Wout + (int64_t)o * E;
        float acc = bout ? bout[o] : 0.0f;
        for (int i = 0; i < E; ++i) acc = fma_(sh[i], wr[i], acc);
        sg[o] = acc;
    }
    __syncthreads();
    const float inv = __builtin_sqrtf((float)E);
    for (int n = tid; n < M; n += blockDim.x) {
        const float* lr = Lb + (int64_t)n * ld;
        float acc = 0.0f;
        for (int o = 0; o < E; ++o) acc = fma_(sg[o], lr[o], acc);
        logits[row * M + n] = acc / inv;
    }
}

int launch_pointer_attention(const float* q, const float* K, const float* V, const float* Lk, int64_t ld, const uint8_t* mask,
                             int mask_per_query, const float* Wout, const float* bout, float* logits, int64_t B, int L, int M,
                             int E, int H, int mask_inner, hipStream_t st)
{
    if (B <= 0 || L <= 0) return 0;
    const size_t lds = (3 * (size_t)E + (size_t)H * M) * sizeof(float);
    if (lds > 64 * 1024 || E / H > 32) return EAMRL_E_ARG;
    hipLaunchKernelGGL(k_pointer_attention, dim3((unsigned)(B * L)), dim3(256), lds, st, q, K, V, Lk, ld, mask, mask_per_query,
                       Wout, bout, logits, L, M, E, H, mask_inner);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

}  // namespace eamrl
