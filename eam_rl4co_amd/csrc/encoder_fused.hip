// The whole GraphAttentionNetwork of one instance in ONE workgroup: node embeddings h [M][128] stay in LDS through all
// layers; every Linear runs on v_mfma_f32_16x16x4_f32 with its A operand read from LDS and its B operand (the weights,
// pre-packed on the host into MFMA fragment order) loaded straight from L2 into registers; the self-attention, the
// residuals and the normalisations happen between the GEMM passes without touching HBM.
//
// Reference: rl4co/models/nn/graph/attnnet.py:16-103 (MultiHeadAttentionLayer / GraphAttentionNetwork),
// rl4co/models/nn/attention.py:66-136 (MultiHeadAttention), rl4co/models/nn/ops.py:32-56 (Normalization),
// rl4co/models/nn/mlp.py:52-61 (MLP 128 -> 512 -> 128, ReLU).
//
// Arithmetic = the canonical order of DESIGN.md 2, bit for bit what the unfused kernels of encoder.hip produce:
//   * every Linear output element is chain_k(x[k], W[j][k], K, bias[j]): the f32 MFMA accumulates k in ascending order
//     (k0..k3 inside an instruction, instructions in order), whatever the tile shape;
//   * attention: s = chain_d(q * 0.25, k) (the power-of-two scale folded into q), w = d_expf(s - max),
//     Z = sequential sum of w over the keys (an MFMA against a row of ones: fma(w, 1, Z)), o = chain_j(w, v) / Z;
//   * h = res + y; BatchNorm(eval): fma(h, gamma / sqrt(var + eps), beta - mean * scale); InstanceNorm: sequential sums
//     over the nodes per channel.
//
// LDS layout ("A layout"): element (row, c) of an activation matrix with K columns lives at
//   row * S + (c & 3) * G + (c >> 2)         (G = 34 for K = 128 buffers, 16 for the 64-column head-group buffers)
// so that lane (i = lane & 15, g = lane >> 4) of a 16x16x4 MFMA reads its A values A[i][4t + g], t = 0, 1, ... as
// consecutive floats (ds_read_b64 = two k-steps), and the C layout of a producing MFMA (column = lane & 15,
// row = 4 * (lane >> 4) + reg) stores with plain ds_write_b32.  (S, G) = (140, 34) makes the K = 128 reads bank-conflict
// free and the stores 2-way.
//
// Phases per layer (8 wavefronts: cw = wave & 3 owns a 32-column strip or a head, rw = wave >> 2 a half of the row tiles):
//   for head group hg in {0, 1} (4 heads = 64 columns):
//     P1  q | k | v of the group's heads (wave cw: head 4 hg + cw): q * 0.25 -> QA, k -> KB (A layout), v -> VT (transposed)
//     P2  attention of (head, query tile) units; scores S^T = K Q^T with the keys placed on MFMA rows in the order that
//         makes the accumulators (= softmax weights) the B operand of the value product; o overwrites q in QA
//     P3  out_proj accumulators += att[:, 64 hg .. 64 hg + 63] x Wo^T   (k ascending across the groups)
//   h1 = norm1(h + out_proj) -> HB
//   for chunk c in 0..3 (128 hidden units):  P4 hidden = relu(h1 W1_c^T + b1) -> HID;  P5 ffn2 accumulators += hidden x W2_c^T
//   h2 = norm2(h1 + ffn2) -> HB
#include "kernels.hpp"

namespace eamrl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FE = 128;          // embed dim
constexpr int FH = 8;            // heads
constexpr int FF = 512;          // feed-forward hidden
constexpr int SA = 140;          // row stride of the K = 128 A-layout buffers (HB, HID)
constexpr int GA = 34;           // their g-stride: (SA, GA) = (140, 34) gives conflict-free ds_read_b64 A fragments AND 2-way (free)
                                 // ds_write_b32 from the C layout; (130, 32) reads as well but stores 4-way (measured: +6 % time)
constexpr int SQ = 66;           // row stride of the 64-column head-group buffers (QA, KB)
constexpr int MAX_FUSED_LAYERS = 8;
constexpr int NCST = 9 * FE + FF;     // floats of per-layer constants staged in LDS

struct FusedLayer {
    const float* Wqkv; const float* bqkv; const float* Wo; const float* bo;      // packed weights (pack_mfma_b), biases
    const float* W1; const float* b1; const float* W2; const float* b2;
    const float* n1g; const float* n1b; const float* n1m; const float* n1v;       // norm 1: gamma, beta, running mean / var
    const float* n2g; const float* n2b; const float* n2m; const float* n2v;
};
struct FusedArgs {
    const float* h_in; float* h_out; int M; int nlayers; int norm; float eps;
    // optional decoder cache (AttentionModelDecoder._precompute_cache) computed from the final embeddings while they are
    // still in LDS: slots 0 .. nproj-1 = h Wc_s^T (K | V | L | Pa (| Pb)), slot nproj = Lp = L Wout
    const float* Wc; const float* WoT; float* cache; int64_t ld; int nproj;
    // optional graph context: gctx[inst] = mean_n(h) Wg^T (embeddings.mean(1) -> project_fixed_context, no bias)
    const float* Wg; float* gctx;
    // optional init embedding (h_in == nullptr): h[n] = Linear(feat[n]) computed straight into LDS (eamrl_encoder_init)
    const float* feat; int F; const float* Wi; const float* bi; const float* depot; int64_t depot_ld; const float* Wd; const float* bd;
    float* init_out;
    FusedLayer L[MAX_FUSED_LAYERS];
};

#ifdef EAMRL_STAMPS   // development build only (tools/build_stamps.sh): per-phase cycle sums over all wavefronts
__device__ unsigned long long g_enc_stamps[24];
#define ESTAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); st_acc[i] += now_ - st_t; st_t = now_; } while (0)
#else
#define ESTAMP(i) do { } while (0)
#endif

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// One ds_read_b64.  The A fragments of two k-steps are 8 bytes at an 8-byte-aligned address; read as plain float2 pairs the
// compiler fuses neighbouring ones into ds_read2_b64, which the LDS services in 16-lane groups on 32 banks (8 array cycles,
// and with the (SA, GA) layout 2-way conflicts on top: 16 cycles per pair -- the 46 % conflict cycles of profiles/r02g_pmc_lds_*),
// while two ds_read_b64 take 2 + 2 cycles on 64 banks, conflict-free (MI355X_MICROARCH.md, LDS table).  A volatile access is
// not merged.
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 lds_read_b64(const float* p)
{
    typedef const volatile __attribute__((address_space(3))) f32x2v* lds_ptr;      // (address-space inference skips volatile accesses)
    const f32x2v v = *(lds_ptr)(p);
    return make_float2(v.x, v.y);
}

// First group of B fragments of a pass: issued early (before the previous phase's epilogue / barrier) so that the L2 latency
// of a pass's first weights is not paid behind the barrier.
template <int CT>
__device__ __forceinline__ void load_b0(float4 (&b0)[CT], const float* const (&wp)[CT])
{
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) b0[ct] = *reinterpret_cast<const float4*>(wp[ct]);
}

// acc[rt][ct] += A[rows of tile rt][k] * B[k][cols of tile ct] over NU groups of 4 k-steps (16 k values each).
//   arow: LDS address of this lane's A row of tile 0 (+ g * G), tiles 16 * S floats apart; NU groups start at float offset 0
//   wp[ct]: this lane's float4 of the first group of column tile ct; consecutive groups are 256 floats apart
//   b0: the first group, already loaded (load_b0)
// Software pipeline: the A fragments (LDS) and B fragments (L2) of group u + 1 are requested before the MFMAs of group u.
// SWAP: the MFMA takes the weight fragment as A and the activation fragment as B, so that acc[rt][ct] holds the TRANSPOSED
// tile (lane = node row, registers = 4 consecutive output columns): the same products in the same k order, laid out for
// float4 stores to row-major memory.
// NRT: this wave's row tiles as a compile-time value (round 3: as a run-time count every k-step group carried a uniform branch
// around the last tile's loads and MFMAs plus the register copies that merge the two paths)
template <int RTW, int CT, int S, bool SWAP, int NRT>
__device__ __forceinline__ void gemm_pass_n(f32x4 (&acc)[RTW][CT], const float* arow, const float* const (&wp)[CT],
                                            int NU, const float4 (&b0)[CT])
{
    constexpr int nrt = NRT;
    float4 bn[CT];
    float2 an0[RTW], an1[RTW];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) bn[ct] = b0[ct];
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt)
        if (rt < nrt) {
            an0[rt] = lds_read_b64(arow + rt * 16 * S);
            an1[rt] = lds_read_b64(arow + rt * 16 * S + 2);
        }
    for (int u = 0; u < NU; ++u) {
        float4 b[CT];
        float2 a0[RTW], a1[RTW];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) b[ct] = bn[ct];
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt) { a0[rt] = an0[rt]; a1[rt] = an1[rt]; }
        if (u + 1 < NU) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) bn[ct] = *reinterpret_cast<const float4*>(wp[ct] + (u + 1) * 256);
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt)
                if (rt < nrt) {
                    an0[rt] = lds_read_b64(arow + rt * 16 * S + 4 * (u + 1));
                    an1[rt] = lds_read_b64(arow + rt * 16 * S + 4 * (u + 1) + 2);
                }
        }
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt)
            if (rt < nrt) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[rt][ct] = SWAP ? mfma4(b[ct].x, a0[rt].x, acc[rt][ct]) : mfma4(a0[rt].x, b[ct].x, acc[rt][ct]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[rt][ct] = SWAP ? mfma4(b[ct].y, a0[rt].y, acc[rt][ct]) : mfma4(a0[rt].y, b[ct].y, acc[rt][ct]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[rt][ct] = SWAP ? mfma4(b[ct].z, a1[rt].x, acc[rt][ct]) : mfma4(a1[rt].x, b[ct].z, acc[rt][ct]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[rt][ct] = SWAP ? mfma4(b[ct].w, a1[rt].y, acc[rt][ct]) : mfma4(a1[rt].y, b[ct].w, acc[rt][ct]);
            }
    }
}

// nrt is one of two values per kernel variant: RTW (the first row half) or NLOW (the second)
template <int RTW, int CT, int S, bool SWAP = false, int NLOW = RTW>
__device__ __forceinline__ void gemm_pass(f32x4 (&acc)[RTW][CT], const float* arow, int nrt, const float* const (&wp)[CT],
                                          int NU, const float4 (&b0)[CT])
{
    if (NLOW == RTW || nrt == RTW) gemm_pass_n<RTW, CT, S, SWAP, RTW>(acc, arow, wp, NU, b0);
    else gemm_pass_n<RTW, CT, S, SWAP, NLOW>(acc, arow, wp, NU, b0);
}

// y = norm(res + acc) for the two column tiles of a wave, written back to HB in place (batch norm: per-column affine;
// instance norm: the sums are written here and normalised per channel after a barrier).
template <int RTW>
__device__ __forceinline__ void residual_norm_store(const f32x4 (&acc)[RTW][2], float* HB, int row0, int nrt, int cw, int j, int G,
                                                    int norm, const float* cst /* LDS: scale [E] | shift [E] */)
{
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int c = 32 * cw + 16 * ct + j;
        const float sc = cst[c], sh = cst[FE + c];
        float* col = HB + (c & 3) * GA + (c >> 2);
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt)
            if (rt < nrt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* p = col + (row0 + 16 * rt + 4 * G + r) * SA;
                    float v = *p + acc[rt][ct][r];
                    if (norm == EAMRL_NORM_BATCH_EVAL) v = fma_(v, sc, sh);
                    *p = v;
                }
            }
    }
}

// InstanceNorm1d(affine) in place on HB: thread = channel, sequential over the M nodes (the order of k_norm_instance).
__device__ __forceinline__ void instance_norm_lds(float* HB, int M, float eps, const float* cst /* LDS: gamma [E] | beta [E] */)
{
    const int c = threadIdx.x;
    if (c < FE) {
        float* col = HB + (c & 3) * GA + (c >> 2);
        float s = 0.0f;
        for (int n = 0; n < M; ++n) s = s + col[n * SA];
        const float mean = s / (float)M;
        float v = 0.0f;
        for (int n = 0; n < M; ++n) { const float d = col[n * SA] - mean; v = fma_(d, d, v); }
        const float inv = 1.0f / __builtin_sqrtf(v / (float)M + eps);
        const float g = cst[c], bt = cst[FE + c];
        for (int n = 0; n < M; ++n) {
            const float d = col[n * SA] - mean;
            col[n * SA] = fma_(d * inv, g, bt);
        }
    }
}

template <int RTT>
__global__ __launch_bounds__(512, 2) void k_encoder_fused(FusedArgs a)
{
    constexpr int RTA = (RTT + 1) / 2;          // row tiles of the first wave half
    constexpr int RTW = RTA;                    // accumulator row tiles per wave
    constexpr int ROWS = 16 * RTT;
    constexpr int SV = ROWS + 4;                // VT row stride
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* HB = lds;                            // [ROWS][SA]   h (residual stream), A layout K = 128
    float* QA = HB + ROWS * SA;                 // [ROWS][SQ]   q * 0.25 of the head group, then the attention output
    float* KB = QA + ROWS * SQ;                 // [ROWS][SQ]   k of the head group
    float* VT = KB + ROWS * SQ;                 // [64][SV]     v of the head group, transposed (column-major)
    float* HID = QA;                            // [ROWS][SA]   FFN hidden chunk (aliases QA | KB | VT, which are dead then)
    float* CST = VT + 64 * SV;                  // [NCST]       the layer's biases and normalisation constants
    static_assert(ROWS * SA <= 2 * ROWS * SQ + 64 * SV, "HID must fit QA | KB | VT");
    // CST: bqkv [3E] | bo [E] | b1 [F] | b2 [E] | norm1 (scale | shift, or gamma | beta) [2E] | norm2 [2E]
    constexpr int C_BQKV = 0, C_BO = 3 * FE, C_B1 = 4 * FE, C_B2 = 4 * FE + FF, C_N1 = 5 * FE + FF, C_N2 = 7 * FE + FF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform (SGPR): branches on it are scalar branches
    const int cw = wv & 3, rw = wv >> 2;
    const int j = lane & 15, G = lane >> 4;     // C layout: column j, rows 4G + r;  A / B layout: row / column j, k index G
    const int M = a.M;
    const int row0 = rw ? 16 * RTA : 0;         // first row of this wave's tiles
    const int nrt = rw ? RTT - RTA : RTA;       // this wave's row tiles
    const int64_t inst = blockIdx.x;
#ifdef EAMRL_STAMPS
    unsigned long long st_acc[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_readcyclecounter();
#endif

    // ---- h into the A layout; rows >= M are zero ---------------------------------------------------------------------
    if (a.h_in) {
        // from HBM (row-major): all of a thread's loads are issued before the first LDS store (a load-store loop pays the HBM
        // latency once per trip: 7 trips = 27 k of the kernel's 900 k cycles, profiles/r03b_stamps_encoder_fused.txt)
        constexpr int NLD = (ROWS * (FE / 4) + 511) / 512;
        const float* src = a.h_in + inst * (int64_t)M * FE;
        float4 v[NLD];
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int idx = tid + u * 512;
            const int row = idx / (FE / 4), q4 = idx % (FE / 4);
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < M) v[u] = *reinterpret_cast<const float4*>(src + (int64_t)row * FE + 4 * q4);
        }
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int idx = tid + u * 512;
            const int row = idx / (FE / 4), q4 = idx % (FE / 4);
            if (row < ROWS) {
                float* p = HB + row * SA + q4;
                p[0] = v[u].x; p[GA] = v[u].y; p[2 * GA] = v[u].z; p[3 * GA] = v[u].w;
            }
        }
    } else {
        // init embedding computed in place (nn/env_embeddings/init.py: Linear(F -> E) of the node features; depot envs: row 0
        // is Linear(2 -> E) of the depot coordinates): each output is chain_k(x[k], W[c][k], F, bias[c]), the order of
        // eamrl_linear.  Thread = (row, four adjacent columns); the weights of its columns stay in registers over the rows.
        const int q4 = tid % (FE / 4), r0 = tid / (FE / 4);          // 32 column groups x 16 row phases
        const int F = a.F;
        float w[4][8], wd[4][2], bb[4], bdv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bb[c] = a.bi ? a.bi[4 * q4 + c] : 0.0f;
            bdv[c] = (a.depot && a.bd) ? a.bd[4 * q4 + c] : 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) w[c][k] = (k < F) ? a.Wi[(4 * q4 + c) * F + k] : 0.0f;
#pragma unroll
            for (int k = 0; k < 2; ++k) wd[c][k] = a.depot ? a.Wd[(4 * q4 + c) * 2 + k] : 0.0f;
        }
        const float* fsrc = a.feat + inst * (int64_t)M * F;
        float* iout = a.init_out ? a.init_out + inst * (int64_t)M * FE : nullptr;
        for (int row = r0; row < ROWS; row += 16) {
            float y[4] = {0.f, 0.f, 0.f, 0.f};
            if (row < M) {
                if (a.depot && row == 0) {
                    const float x0 = a.depot[inst * a.depot_ld], x1 = a.depot[inst * a.depot_ld + 1];
#pragma unroll
                    for (int c = 0; c < 4; ++c) y[c] = fma_(x1, wd[c][1], fma_(x0, wd[c][0], bdv[c]));
                } else {
                    float x[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) x[k] = (k < F) ? fsrc[(int64_t)row * F + k] : 0.0f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float acc = bb[c];
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            if (k < F) acc = fma_(x[k], w[c][k], acc);
                        y[c] = acc;
                    }
                }
                if (iout) *reinterpret_cast<float4*>(iout + (int64_t)row * FE + 4 * q4) = make_float4(y[0], y[1], y[2], y[3]);
            }
            float* p = HB + row * SA + q4;
            p[0] = y[0]; p[GA] = y[1]; p[2 * GA] = y[2]; p[3 * GA] = y[3];
        }
    }
    __syncthreads();
    ESTAMP(0);

    for (int layer = 0; layer < a.nlayers; ++layer) {
        const FusedLayer& Ly = a.L[layer];
        // ---- the layer's biases and normalisation constants -> LDS (one exposed L2 latency per layer instead of one per
        //      pass: every accumulator is initialised with its bias before the first MFMA can issue) ------------------------
        for (int i = tid; i < NCST; i += blockDim.x) {
            float v;
            if (i < C_BO) v = Ly.bqkv[i];
            else if (i < C_B1) v = Ly.bo[i - C_BO];
            else if (i < C_B2) v = Ly.b1[i - C_B1];
            else if (i < C_N1) v = Ly.b2[i - C_B2];
            else {
                const bool second = i >= C_N2;
                const int k = (i - (second ? C_N2 : C_N1));
                const int c = k & (FE - 1);
                const float* gam = second ? Ly.n2g : Ly.n1g;
                const float* bet = second ? Ly.n2b : Ly.n1b;
                if (a.norm == EAMRL_NORM_BATCH_EVAL) {
                    const float* mean = second ? Ly.n2m : Ly.n1m;
                    const float* var = second ? Ly.n2v : Ly.n1v;
                    const float sc = gam[c] / __builtin_sqrtf(var[c] + a.eps);
                    const float ms = mean[c] * sc;
                    v = k < FE ? sc : bet[c] - ms;
                } else {
                    v = k < FE ? gam[c] : bet[c];
                }
            }
            CST[i] = v;
        }
        __syncthreads();
        // out_proj accumulators: this wave's column tiles 2 cw, 2 cw + 1
        f32x4 acc_o[RTW][2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const float b = CST[C_BO + 32 * cw + 16 * ct + j];
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) acc_o[rt][ct] = splat4(b);
        }
        for (int hg = 0; hg < 2; ++hg) {
            // weights of P3 (this head group's slice of out_proj): first fragments requested now, used after P1 and P2
            const float* wp3[2];
            float4 b3[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                wp3[ct] = Ly.Wo + ((int64_t)(2 * cw + ct) * (FE / 16) + 4 * hg) * 256 + lane * 4;
            // ---- P1: q | k | v of head 4 hg + cw ---------------------------------------------------------------
            {
                f32x4 acc[RTW][3];
                const int ctq = 4 * hg + cw;
                const float* wp[3];
                float4 b0[3];
#pragma unroll
                for (int x = 0; x < 3; ++x) wp[x] = Ly.Wqkv + ((int64_t)(8 * x + ctq) * (FE / 16)) * 256 + lane * 4;
                load_b0<3>(b0, wp);
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    const float b = CST[C_BQKV + 16 * (8 * x + ctq) + j];
#pragma unroll
                    for (int rt = 0; rt < RTW; ++rt) acc[rt][x] = splat4(b);
                }
                ESTAMP(16);
                gemm_pass<RTW, 3, SA, false, RTT - RTA>(acc, HB + (row0 + j) * SA + G * GA, nrt, wp, FE / 16, b0);
                ESTAMP(17);
                load_b0<2>(b3, wp3);
                // q (scaled) and k in the A layout of the 64-column buffers: column 16 cw + j -> (g = j & 3, t = 4 cw + (j >> 2))
                float* qcol = QA + (j & 3) * 16 + 4 * cw + (j >> 2);
                float* kcol = KB + (j & 3) * 16 + 4 * cw + (j >> 2);
                float* vrow = VT + (16 * cw + j) * SV;
#pragma unroll
                for (int rt = 0; rt < RTW; ++rt)
                    if (rt < nrt) {
                        const int rbase = row0 + 16 * rt + 4 * G;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            qcol[(rbase + r) * SQ] = acc[rt][0][r] * 0.25f;
                            kcol[(rbase + r) * SQ] = acc[rt][1][r];
                        }
                        *reinterpret_cast<float4*>(vrow + rbase) =
                            make_float4(acc[rt][2][0], acc[rt][2][1], acc[rt][2][2], acc[rt][2][3]);
                    }
            }
            ESTAMP(1);
            __syncthreads();
            ESTAMP(2);
            // ---- P2: attention of head cw (of this group), query tiles of this wave half -----------------------------
            {
                // key on MFMA row rho of a key tile: 4 * (rho & 3) + (rho >> 2)  -> accumulator reg r of lane group G holds
                // key 4 r + G, i.e. the B operand of value k-step r
                const int pi = 4 * (j & 3) + (j >> 2);
                float kf[RTT][4];
#pragma unroll
                for (int kt = 0; kt < RTT; ++kt) {
                    const float* p = KB + (16 * kt + pi) * SQ + G * 16 + 4 * cw;
                    const float2 lo = *reinterpret_cast<const float2*>(p), hi = *reinterpret_cast<const float2*>(p + 2);
                    kf[kt][0] = lo.x; kf[kt][1] = lo.y; kf[kt][2] = hi.x; kf[kt][3] = hi.y;
                }
                float vf[4 * RTT];
                const int NT = (M + 3) >> 2;             // value k-steps (4 keys each)
                constexpr int FULLT = RTT == 7 ? 4 : RTT == 4 ? 2 : 0;       // launch_encoder_fused: M > 64 / M > 32 / any
#pragma unroll
                for (int t = 0; t < 4 * RTT; ++t) vf[t] = (t < NT) ? VT[(16 * cw + j) * SV + 4 * t + G] : 0.0f;
                for (int q = 0; q < nrt; ++q) {
                    const int qrow = row0 + 16 * q + j;
                    const float* qp = QA + qrow * SQ + G * 16 + 4 * cw;
                    const float2 qlo = *reinterpret_cast<const float2*>(qp), qhi = *reinterpret_cast<const float2*>(qp + 2);
                    f32x4 s[RTT];
                    float m = -INFINITY;
                    // the four dependent k-steps of a key tile are interleaved with those of the other tiles (an f32 16x16x4
                    // MFMA has a 40-cycle dependent latency against a 32-cycle issue interval)
#pragma unroll
                    for (int kt = 0; kt < RTT; ++kt) s[kt] = mfma4(kf[kt][0], qlo.x, splat4(0.0f));
#pragma unroll
                    for (int kt = 0; kt < RTT; ++kt) s[kt] = mfma4(kf[kt][1], qlo.y, s[kt]);
#pragma unroll
                    for (int kt = 0; kt < RTT; ++kt) s[kt] = mfma4(kf[kt][2], qhi.x, s[kt]);
#pragma unroll
                    for (int kt = 0; kt < RTT; ++kt) s[kt] = mfma4(kf[kt][3], qhi.y, s[kt]);
#pragma unroll
                    for (int kt = 0; kt < RTT; ++kt) {
                        // (uniform) the tile holds padded keys: the variant's M range leaves the first FULLT tiles always full
                        if (kt >= FULLT && 16 * kt + 16 > M) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (16 * kt + 4 * r + G >= M) s[kt][r] = -INFINITY;
                        }
                        m = vmax5_raw(m, s[kt][0], s[kt][1], s[kt][2], s[kt][3]);      // two v_max3_f32, one asm statement
                    }
                    {   // the four lane groups G share a query: max over lanes l, l^16, l^32, l^48
                        auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                        m = vmax_raw(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
                        auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                        m = vmax_raw(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
                    }
#pragma unroll
                    for (int kt = 0; kt < RTT; ++kt) {      // packed fp32 math, the two pairs of a tile statement by statement: each
                        f32x2 e01 = (f32x2){s[kt][0], s[kt][1]} - splat2(m), e23 = (f32x2){s[kt][2], s[kt][3]} - splat2(m);
                        d_expf2_nonpos_x2(e01, e23);                                          // element bit-identical to d_expf
                        s[kt] = (f32x4){e01.x, e01.y, e23.x, e23.y};
                    }
                    // Z (canonical order, encoder.hip ZRot): this lane holds the keys 4 r + G of every tile, i.e. one residue class
                    // mod 4 in ascending order -> its partial sum P_G in 4 RTT - 1 adds, then (P0 + P1) + (P2 + P3) over the lane groups
                    float zp = s[0][0];
#pragma unroll
                    for (int t = 1; t < 4 * RTT; ++t) zp = zp + s[t >> 2][t & 3];
                    {
                        auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(zp), __float_as_uint(zp), false, false);
                        zp = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
                        auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(zp), __float_as_uint(zp), false, false);
                        zp = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
                    }
                    f32x4 o = splat4(0.0f);
#pragma unroll
                    for (int kt = 0; kt < RTT; ++kt)        // a (uniform) test per key tile that may be empty, none per k-step: the padded
                                                            // k-steps of a started tile multiply zero weights with zero values, and
                                                            // fma(0, 0, o) == o (o is never -0: it starts from +0)
                        if (kt < FULLT || 16 * kt < M) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) o = mfma4(vf[4 * kt + r], s[kt][r], o);
                        }
                    // o: lane (query j, G), reg r -> head column e = 4 G + r -> A layout (g = r, t = 4 cw + G); overwrites q
                    float* op = QA + qrow * SQ + 4 * cw + G;
#pragma unroll
                    for (int r = 0; r < 4; ++r) op[r * 16] = o[r] / zp;
                }
            }
            ESTAMP(3);
            __syncthreads();
            ESTAMP(4);
            // ---- P3: out_proj partial over the 64 attention columns of this head group ------------------------------
            gemm_pass<RTW, 2, SQ, false, RTT - RTA>(acc_o, QA + (row0 + j) * SQ + G * 16, nrt, wp3, 4, b3);
            ESTAMP(5);
            __syncthreads();
            ESTAMP(6);
        }
        // ---- h1 = norm1(h + out_proj) -> HB ----------------------------------------------------------------------------
        residual_norm_store<RTW>(acc_o, HB, row0, nrt, cw, j, G, a.norm, CST + C_N1);
        __syncthreads();
        if (a.norm == EAMRL_NORM_INSTANCE) {
            instance_norm_lds(HB, M, a.eps, CST + C_N1);
            __syncthreads();
        }
        ESTAMP(7);
        // ---- FFN: 4 chunks of 128 hidden units ---------------------------------------------------------------------------
        f32x4 acc_f[RTW][2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const float b = CST[C_B2 + 32 * cw + 16 * ct + j];
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) acc_f[rt][ct] = splat4(b);
        }
        for (int ch = 0; ch < FF / 128; ++ch) {
            const float* wp5[2];
            float4 b5[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                wp5[ct] = Ly.W2 + ((int64_t)(2 * cw + ct) * (FF / 16) + 8 * ch) * 256 + lane * 4;
            {   // P4: hidden chunk = relu(h1 W1_ch^T + b1) -> HID
                f32x4 acc[RTW][2];
                const float* wp[2];
                float4 b0[2];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) wp[ct] = Ly.W1 + ((int64_t)(8 * ch + 2 * cw + ct) * (FE / 16)) * 256 + lane * 4;
                load_b0<2>(b0, wp);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const float b = CST[C_B1 + 16 * (8 * ch + 2 * cw + ct) + j];
#pragma unroll
                    for (int rt = 0; rt < RTW; ++rt) acc[rt][ct] = splat4(b);
                }
                ESTAMP(14);
                gemm_pass<RTW, 2, SA, false, RTT - RTA>(acc, HB + (row0 + j) * SA + G * GA, nrt, wp, FE / 16, b0);
                ESTAMP(15);
                load_b0<2>(b5, wp5);           // P5's first weights fly during the epilogue and the barrier
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int c = 32 * cw + 16 * ct + j;
                    float* col = HID + (c & 3) * GA + (c >> 2);
#pragma unroll
                    for (int rt = 0; rt < RTW; ++rt)
                        if (rt < nrt) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float v = acc[rt][ct][r];
                                col[(row0 + 16 * rt + 4 * G + r) * SA] = !(v > 0.0f) ? 0.0f : v;
                            }
                        }
                }
            }
            ESTAMP(8);
            __syncthreads();
            ESTAMP(9);
            // P5: ffn2 accumulators += hidden chunk x W2[:, 128 ch .. 128 ch + 127]^T
            gemm_pass<RTW, 2, SA, false, RTT - RTA>(acc_f, HID + (row0 + j) * SA + G * GA, nrt, wp5, 8, b5);
            ESTAMP(10);
            __syncthreads();
            ESTAMP(11);
        }
        // ---- h2 = norm2(h1 + ffn) -> HB ------------------------------------------------------------------------------------
        residual_norm_store<RTW>(acc_f, HB, row0, nrt, cw, j, G, a.norm, CST + C_N2);
        __syncthreads();
        if (a.norm == EAMRL_NORM_INSTANCE) {
            instance_norm_lds(HB, M, a.eps, CST + C_N2);
            __syncthreads();
        }
        ESTAMP(12);
    }
    // ---- store h (row-major, float4 per thread); skipped when the caller keeps only the decoder cache -----------------------
    if (a.h_out) {
        float* dst = a.h_out + inst * (int64_t)M * FE;
        for (int idx = tid; idx < M * (FE / 4); idx += blockDim.x) {
            const int row = idx / (FE / 4), q4 = idx % (FE / 4);
            const float* p = HB + row * SA + q4;
            *reinterpret_cast<float4*>(dst + (int64_t)row * FE + 4 * q4) = make_float4(p[0], p[GA], p[2 * GA], p[3 * GA]);
        }
    }
    ESTAMP(13);
    // ---- graph context: mean over the nodes in node order (k_mean_nodes), then a k-ordered chain per output (k_linear) ----
    if (a.gctx) {
        float* MEAN = CST;                          // the layer constants are dead
        if (tid < FE) {
            const float* col = HB + (tid & 3) * GA + (tid >> 2);
            float s = 0.0f;
            for (int n = 0; n < M; ++n) s = s + col[n * SA];
            MEAN[tid] = s / (float)M;
        }
        __syncthreads();
        if (tid < FE) {
            const float4* w = reinterpret_cast<const float4*>(a.Wg + (int64_t)tid * FE);
            float acc = 0.0f;
#pragma unroll 8
            for (int k4 = 0; k4 < FE / 4; ++k4) {
                const float4 wv4 = w[k4];
                acc = fma_(MEAN[4 * k4 + 0], wv4.x, acc);
                acc = fma_(MEAN[4 * k4 + 1], wv4.y, acc);
                acc = fma_(MEAN[4 * k4 + 2], wv4.z, acc);
                acc = fma_(MEAN[4 * k4 + 3], wv4.w, acc);
            }
            a.gctx[inst * FE + tid] = acc;
        }
    }
    // ---- decoder cache from the resident embeddings: nproj projections of h, then Lp = L Wout ------------------------------
    if (a.cache) {
        float* STG = QA;                           // L in the A layout (operand of the Lp pass); aliases the dead attention buffers
        float* crow = a.cache + (inst * (int64_t)M) * a.ld;
        for (int sl = 0; sl <= a.nproj; ++sl) {
            const bool lp = sl == a.nproj;
            f32x4 acc[RTW][2];
            const float* wp[2];
            float4 b0[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                wp[ct] = (lp ? a.WoT : a.Wc + (int64_t)sl * FE * FE) + ((int64_t)(2 * cw + ct) * (FE / 16)) * 256 + lane * 4;
            load_b0<2>(b0, wp);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int rt = 0; rt < RTW; ++rt) acc[rt][ct] = splat4(0.0f);
            if (lp) __syncthreads();               // every wave's part of L is in STG
            gemm_pass<RTW, 2, SA, true, RTT - RTA>(acc, (lp ? STG : HB) + (row0 + j) * SA + G * GA, nrt, wp, FE / 16, b0);
            // transposed tiles: lane = node row0 + 16 rt + j, registers = output columns 32 cw + 16 ct + 4 G + (0..3)
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt)
                if (rt < nrt) {
                    const int node = row0 + 16 * rt + j;
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const int c = 32 * cw + 16 * ct + 4 * G;
                        const f32x4 v = acc[rt][ct];
                        if (node < M)
                            *reinterpret_cast<float4*>(crow + (int64_t)node * a.ld + sl * FE + c) = make_float4(v[0], v[1], v[2], v[3]);
                        if (sl == 2) {             // L also feeds the Lp pass: element (node, c + r) -> A layout (g = r, t = c / 4)
                            float* p = STG + node * SA + (c >> 2);
                            p[0] = v[0]; p[GA] = v[1]; p[2 * GA] = v[2]; p[3 * GA] = v[3];
                        }
                    }
                }
        }
    }
#ifdef EAMRL_STAMPS
    if (lane == 0) {
        for (int i = 0; i < 20; ++i) atomicAdd(&g_enc_stamps[i], st_acc[i]);
        atomicAdd(&g_enc_stamps[20], 1ull);
    }
#endif
}

// Wp[ct][u][16 g + j][q] = W[16 ct + j][16 u + 4 q + g]: lane (j, g) of a 16x16x4 MFMA finds its B values of four consecutive
// k-steps in one float4, and a wavefront's 64 float4 are 1 KB contiguous.
__global__ void k_pack_mfma_b(const float* __restrict__ W, float* __restrict__ Wp, int N, int K)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * K) return;
    const int q = (int)(idx & 3), l = (int)((idx >> 2) & 63);
    const int64_t blk = idx >> 8;                 // ct * (K / 16) + u
    const int u = (int)(blk % (K / 16)), ct = (int)(blk / (K / 16));
    const int j = l & 15, g = l >> 4;
    Wp[idx] = W[(int64_t)(16 * ct + j) * K + 16 * u + 4 * q + g];
}

int launch_pack_mfma_b(const float* W, float* Wp, int N, int K, hipStream_t st)
{
    const int64_t n = (int64_t)N * K;
    hipLaunchKernelGGL(k_pack_mfma_b, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, Wp, N, K);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

template <int RTT>
static int launch_fused_t(const FusedArgs& a, int64_t B, hipStream_t st)
{
    constexpr int ROWS = 16 * RTT;
    const size_t lds = ((size_t)ROWS * SA + 2 * (size_t)ROWS * SQ + 64 * (size_t)(ROWS + 4) + NCST) * sizeof(float);
    auto k = k_encoder_fused<RTT>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return EAMRL_E_LAUNCH;
    hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(512), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

#ifdef EAMRL_STAMPS
extern "C" __attribute__((visibility("default"))) int eamrl_debug_read_enc_stamps(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_enc_stamps), sizeof(g_enc_stamps)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[24] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_enc_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

bool encoder_fused_supports(int M, int E, int H, int FFdim, int nlayers)
{
    return M >= 1 && M <= 112 && E == FE && H == FH && FFdim == FF && nlayers >= 1 && nlayers <= MAX_FUSED_LAYERS;
}

int launch_encoder_fused(const float* h_in, float* h_out, int64_t B, int M, int nlayers, int norm, float eps,
                         const eamrl_encoder_layer* layers, const eamrl_encoder_cache* cache, const eamrl_encoder_init* init,
                         hipStream_t st)
{
    if (B <= 0) return 0;
    FusedArgs a;
    a.h_in = h_in; a.h_out = h_out; a.M = M; a.nlayers = nlayers; a.norm = norm; a.eps = eps;
    a.feat = nullptr; a.F = 0; a.Wi = a.bi = a.depot = a.Wd = a.bd = nullptr; a.depot_ld = 0; a.init_out = nullptr;
    if (init) {
        a.h_in = nullptr;
        a.feat = init->feat; a.F = init->F; a.Wi = init->W; a.bi = init->b; a.depot = init->depot; a.depot_ld = init->depot_ld;
        a.Wd = init->Wd; a.bd = init->bd; a.init_out = init->init_out;
    }
    a.Wc = nullptr; a.WoT = nullptr; a.cache = nullptr; a.ld = 0; a.nproj = 0; a.Wg = nullptr; a.gctx = nullptr;
    if (cache) {
        a.Wc = cache->Wc; a.WoT = cache->WoutT; a.cache = cache->out; a.ld = cache->ld; a.nproj = cache->nproj;
        if (cache->Wg && cache->gctx) { a.Wg = cache->Wg; a.gctx = cache->gctx; }
    }
    for (int l = 0; l < nlayers; ++l) {
        const eamrl_encoder_layer& s = layers[l];
        a.L[l] = FusedLayer{s.Wqkv, s.bqkv, s.Wo, s.bo, s.W1, s.b1, s.W2, s.b2, s.n1_gamma, s.n1_beta, s.n1_mean, s.n1_var,
                            s.n2_gamma, s.n2_beta, s.n2_mean, s.n2_var};
    }
    if (M <= 32) return launch_fused_t<2>(a, B, st);
    if (M <= 64) return launch_fused_t<4>(a, B, st);
    return launch_fused_t<7>(a, B, st);
}

}  // namespace eamrl
