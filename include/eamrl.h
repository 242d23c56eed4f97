/*
 * eamrl.h -- C ABI of libeamrl_hip.so: the MI355X (gfx950) construction-rollout hot path.
 *
 * The reference (RL4CO fork, /root/reference) is pure Python and has no FFI; its seam for this
 * path is the RL4COEnvBase / AttentionModelPolicy Python interface.  The
 * entry points below are what a binding for that seam calls; each cites the reference code it
 * replaces (paths relative to /root/reference).  INTEGRATION.md shows the ctypes stubs.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); kernels are only enqueued:
 *     no entry point allocates, frees, copies to the host or synchronises;
 *   - return value: 0 = enqueued, <0 = rejected before launch (EAMRL_E_*); text via eamrl_last_error();
 *   - data-dependent failures found ON the device (the reference's asserts) are OR-ed into the
 *     caller's `status` word (device int32, caller zeroes it): EAMRL_ST_* bits;
 *   - floats are fp32, indices int64 (torch.long), masks/visited uint8 (torch.bool / torch.uint8
 *     storage), all row-major and contiguous unless a leading dimension `ld*` (in elements) is given;
 *   - "rows" R are rollout rows: one per instance, or S*B rows in the reference's "(s b)" order for
 *     multistart (row r reads the cache of instance r % B)   [rl4co/utils/ops.py:13-56];
 *   - arithmetic follows the defined order documented in DESIGN.md ("Canonical arithmetic"); results
 *     are bit-identical to oracle/eamrl_oracle.c.
 */
#ifndef EAMRL_H
#define EAMRL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EAMRL_VERSION 100 /* 0.1.0 */

/* environments */
#define EAMRL_ENV_TSP 0
#define EAMRL_ENV_CVRP 1
#define EAMRL_ENV_SDVRP 2 /* split delivery: CVRP instances, customers may be served in several visits */
#define EAMRL_ENV_PCTSP 3 /* prize collecting TSP: return to the depot once the collected prize reaches 1 */
#define EAMRL_ENV_OP 4    /* orienteering: collect prizes within a maximum tour length */
#define EAMRL_ENV_CVRPTW 5 /* CVRP with time windows: a customer must be reached before its window closes */
/* selection modes  [rl4co/utils/decoding.py:430-465] */
#define EAMRL_GREEDY 0
#define EAMRL_SAMPLE 1   /* argmax(p / noise), noise ~ Exp(1) supplied by the caller (== torch.multinomial) */
#define EAMRL_EVALUATE 2 /* teacher forcing: action given */
/* normalisation kinds  [rl4co/models/nn/ops.py:32-56] */
#define EAMRL_NORM_BATCH_EVAL 0
#define EAMRL_NORM_INSTANCE 1

/* host-side rejections */
#define EAMRL_E_ARG (-1)     /* bad size / null pointer / unsupported shape */
#define EAMRL_E_LAUNCH (-2)  /* hipLaunchKernel failed */
/* device-side status bits (the reference's asserts) */
#define EAMRL_ST_NAN_LOGITS 1u     /* "Logits contain NaNs"          nn/attention.py:303-304 */
#define EAMRL_ST_INFEASIBLE 2u     /* "infeasible action selected"   utils/decoding.py:397-399,413-415 */
#define EAMRL_ST_STEP_OVERRUN 4u   /* decode loop hit t_max before all rows were done   constructive/base.py:246-250 */

int eamrl_version(void);
const char* eamrl_last_error(void);
/* Diagnostic knobs (tests / profiling only; results never depend on them, only the kernel variant used).
 * key 0: 1 = run eamrl_linear on the VALU cross-check kernel instead of the MFMA kernel.
 * key 1: 1 = eamrl_am_rollout always uses the streaming kernel (never the register-resident one).
 * key 4: 1 = eamrl_linear uses 128-row tiles instead of 64-row tiles.
 * key 3: 1 = eamrl_mha_encoder uses the one-row-per-thread kernel even where the blocked one applies.
 * key 10: 1 = eamrl_linear configures its epilogue at run time even where a compile-time variant applies.
 * key 6: 1 = eamrl_am_rollout does not use the start-sharing kernel for multistart batches (R = S*B rows).
 * key 11: 1 = eamrl_am_rollout does not use the MFMA start-sharing kernel (TSP multistart) but the VALU ones.
 * key 13: 1 = the MFMA start-sharing kernel never splits an instance's starts over several workgroups (small batches).
 * key 14: 1 = multistart batches of the depot envs (CVRP, CVRPTW, SDVRP, PCTSP, OP) use the VALU start-sharing kernel, not the MFMA one. */
int eamrl_debug_set(int key, int value);

/* ---- environment state machines ---------------------------------------------------------------- */

/* TSPEnv._step  [rl4co/envs/routing/tsp/env.py:62-88].  In place on (mask, first, cur, istep, done).
 * mask [R][N] u8 (1 = still available), first/cur/istep [R] i64, action [R] i64, done [R] u8. */
int eamrl_tsp_step(uint8_t* mask, int64_t* first, int64_t* cur, int64_t* istep, const int64_t* action,
                   uint8_t* done, int64_t R, int N, void* stream);

/* CVRPEnv.get_action_mask  [rl4co/envs/routing/cvrp/env.py:132-144].  M = N + 1 nodes, depot = 0.
 * visited [R][M] u8, used/vcap [R] f32, demand [B][N] f32, cur [R] i64 -> mask [R][M] u8 (1 = feasible). */
int eamrl_cvrp_mask(const uint8_t* visited, const float* used, const float* vcap, const float* demand,
                    const int64_t* cur, uint8_t* mask, int64_t R, int64_t B, int N, void* stream);

/* CVRPEnv._step + get_action_mask  [rl4co/envs/routing/cvrp/env.py:68-100,132-144].  In place. */
int eamrl_cvrp_step_mask(uint8_t* visited, float* used, const float* vcap, const float* demand, int64_t* cur,
                         const int64_t* action, uint8_t* mask, uint8_t* done, int64_t R, int64_t B, int N,
                         void* stream);

/* SDVRPEnv._step + get_action_mask  [rl4co/envs/routing/sdvrp/env.py:58-92,137-146].  In place.
 * rem [R][M] f32 = demand_with_depot (remaining demand, depot slot 0), used/vcap [R] f32, cur [R] i64,
 * mask [R][M] u8, done [R] u8.  action == NULL: only the mask is recomputed (get_action_mask; done may be NULL).
 * The vehicle delivers min(rem[action], vcap - used); a row is done when no remaining demand is > 0. */
int eamrl_sdvrp_step_mask(float* rem, float* used, const float* vcap, int64_t* cur, const int64_t* action,
                          uint8_t* mask, uint8_t* done, int64_t R, int M, void* stream);

/* PCTSPEnv._step + get_action_mask  [rl4co/envs/routing/pctsp/env.py:64-97,156-163].  In place.
 * visited [R][M] u8, prize_tot [R] f32 (cur_total_prize), pen_tot [R] f32 (cur_total_penalty) or NULL, prize / penalty
 * [B][M] f32 with a zero depot slot (penalty may be NULL with pen_tot), cur / istep [R] i64, mask [R][M] u8, done [R] u8.
 * action == NULL: only the mask is recomputed.  A customer is feasible until visited and until the depot was visited;
 * the depot is infeasible while prize_tot < 1 and a customer is unvisited; done = (istep > 0 and action == depot). */
int eamrl_pctsp_step_mask(uint8_t* visited, float* prize_tot, float* pen_tot, const float* prize, const float* penalty,
                          int64_t* cur, int64_t* istep, const int64_t* action, uint8_t* mask, uint8_t* done, int64_t R,
                          int64_t B, int M, void* stream);

/* OPEnv._step + get_action_mask  [rl4co/envs/routing/op/env.py:69-102,149-165].  In place.
 * visited [R][M] u8, tour_len [R] f32, prize_tot [R] f32 (current_total_prize) or NULL with prize [B][M] (zero depot
 * slot), locs [B][M][2] f32 (depot first), maxlen [B][M] f32 (the reset state's per-node arrival limit: max_length -
 * distance to the depot - 1e-6), cur / istep [R] i64, mask [R][M] u8, done [R] u8.  action == NULL: mask only.
 * A customer is feasible until visited, until the depot was visited and while tour_len + distance(cur, n) <= maxlen[n];
 * the depot is always feasible; done = (action == depot and istep > 0).  Distances are sqrtf(fmaf(dy, dy, dx*dx)),
 * which is what torch's norm(p=2, dim=-1) computes for two components. */
int eamrl_op_step_mask(uint8_t* visited, float* tour_len, float* prize_tot, const float* prize, const float* locs,
                       const float* maxlen, int64_t* cur, int64_t* istep, const int64_t* action, uint8_t* mask,
                       uint8_t* done, int64_t R, int64_t B, int M, void* stream);

/* CVRPTWEnv._step + get_action_mask  [rl4co/envs/routing/cvrptw/env.py:103-138] = the CVRP transition and mask plus the
 * clock: time = (a != 0) * (max(time + distance(cur, a), tw[a][0]) + dur[a]); a node is feasible only if additionally
 * time + distance(cur, n) <= tw[n][1].  time [R] f32 (current_time), locs [B][M][2], tw [B][M][2] f32 (start, end; the
 * reference's int32 windows converted, exact), dur [B][M] f32; the rest as eamrl_cvrp_step_mask.  action == NULL: mask
 * only.  In place. */
int eamrl_cvrptw_step_mask(uint8_t* visited, float* used, const float* vcap, const float* demand, int64_t* cur, float* time,
                           const float* locs, const float* tw, const float* dur, const int64_t* action, uint8_t* mask,
                           uint8_t* done, int64_t R, int64_t B, int N, void* stream);

/* The time-window replay of CVRPTWEnv.check_solution_validity  [cvrptw/env.py:203-227] (arrival times truncated to
 * integers as there): bad[0] += rows that start a service after its window closed.  The CVRP part of the check is
 * eamrl_check_solution(EAMRL_ENV_CVRP). */
int eamrl_cvrptw_check_time(const int64_t* actions, const float* locs, const float* tw, const float* dur, int64_t R,
                            int64_t B, int M, int T, int32_t* bad, void* stream);

/* ---- one-shot encoder + cache ------------------------------------------------------------------- */

/* y[r][j] = bias[j] + sum_k x[r][k] * W[j][k]  (k-ordered fma chain), optional ReLU, optional residual:
 * y = res + (...) .  torch.nn.Linear as used by init embeddings, Wqkv/out_proj, the FFN and the decoder's
 * projections  [models/nn/env_embeddings/init.py:55-68,115-138; nn/attention.py:112-136; nn/mlp.py:52-61;
 * zoo/am/decoder.py:206-235].  W is [out][in] (ldw = in unless given).  x rows have stride ldx, y rows ldy. */
int eamrl_linear(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* bias, const float* res,
                 int64_t ldres, float* y, int64_t ldy, int64_t rows, int in_dim, int out_dim, int relu, void* stream);

/* eamrl_linear followed by Normalization(batch, eval) on the result, in one launch:
 * y = BN_eval(res + (bias + x W^T))   [SkipConnection + Normalization, nn/graph/attnnet.py:47-57, nn/ops.py:45-47].
 * Same arithmetic as eamrl_linear then eamrl_normalize(EAMRL_NORM_BATCH_EVAL). */
int eamrl_linear_bn(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* bias, const float* res,
                    int64_t ldres, float* y, int64_t ldy, int64_t rows, int in_dim, int out_dim,
                    const float* gamma, const float* beta, const float* mean, const float* var, float eps, void* stream);

/* y[r][j] = sum_k x[r][k] * Wt[k][j]  (right-multiplication; folds PointerAttention.project_out into the
 * logit key: Lp = L * Wout)  [nn/attention.py:296-301]. */
int eamrl_matmul_right(const float* x, int64_t ldx, const float* Wt, float* y, int64_t ldy, int64_t rows,
                       int in_dim, int out_dim, void* stream);

/* Weight and bias gradient of torch.nn.Linear for the training graph (the encoder's and the cache projections' Linears as
 * differentiated by loss.backward() of the REINFORCE / POMO / EAM trainers  [models/rl/reinforce/reinforce.py:62-64,103-106;
 * zoo/pomo/model.py:103-112; zoo/earl/model.py:179-195]):  dW[o][i] = sum_r dy[r][o] x[r][i]  ([out_dim][in_dim], written,
 * not accumulated), db[o] = sum_r dy[r][o] (or NULL).  out_dim and in_dim multiples of 128; rows of dy / x 16-byte aligned.
 * scratch: eamrl_linear_wgrad_scratch(rows, out_dim, in_dim) floats (-1: shape not supported).  Tile-order sums (not part
 * of the bit-exact rollout path). */
int64_t eamrl_linear_wgrad_scratch(int64_t rows, int out_dim, int in_dim);
int eamrl_linear_wgrad(const float* dy, int64_t ldy, const float* x, int64_t ldx, int64_t rows, int out_dim, int in_dim,
                       float* dW, float* db, float* scratch, int64_t scratch_floats, void* stream);

/* The same for the init embeddings' Linears (in_dim = 2 .. 8 node features -> E  [nn/env_embeddings/init.py:55-68,115-138]):
 * dW [out_dim][in_dim], db [out_dim] (or NULL).  scratch: eamrl_small_linear_wgrad_scratch(rows, out_dim) floats. */
int64_t eamrl_small_linear_wgrad_scratch(int64_t rows, int out_dim);
int eamrl_small_linear_wgrad(const float* dy, int64_t ldy, const float* x, int64_t ldx, int64_t rows, int out_dim, int in_dim,
                             float* dW, float* db, float* scratch, int64_t scratch_floats, void* stream);

/* Gradient of eamrl_batchnorm_train (BatchNorm1d with batch statistics under loss.backward(), nn/ops.py:45-47):
 * x, dy [rows][E] (x = the INPUT of the normalisation), save_mean / save_var as written by eamrl_batchnorm_train,
 * gamma [E] (NULL: ones) -> dx [rows][E], dgamma [E], dbeta [E] (either may be NULL).  E a multiple of 4.
 * scratch: eamrl_batchnorm_backward_scratch(rows, E) floats.  Chunk-order sums (gradient path, not bit-exact). */
int64_t eamrl_batchnorm_backward_scratch(int64_t rows, int E);
int eamrl_batchnorm_backward(const float* x, const float* dy, const float* save_mean, const float* save_var, const float* gamma,
                             float eps, int64_t rows, int E, float* dx, float* dgamma, float* dbeta, float* scratch,
                             int64_t scratch_floats, void* stream);

/* Instance augmentation of coordinates for the evaluators  [data/transforms.py:16-90: dihedral_8_augmentation,
 * symmetric_transform / symmetric_augmentation; tasks/eval.py:138-297]: out [R][N][2], row r = a * B + b is instance b of
 * xy [B][N][2] (the "(a b)" batchify order) transformed by code[r]: 0..7 = the dihedral variant (x, y), (1-x, y), (x, 1-y),
 * (1-x, 1-y), (y, x), (1-y, x), (y, 1-x), (1-y, 1-x); 8 = rotation about (offset, offset) by the angle whose (cos, sin) is
 * cs[r][0..1]; 9 = that rotation followed by x <-> y.  cs may be NULL when no code is >= 8.  Separately rounded fp32
 * products and sums (no fma), as the reference's tensor expressions evaluate them. */
int eamrl_augment_xy(const float* xy, const float* cs, const int32_t* code, float* out, int64_t R, int64_t B, int N, float offset,
                     void* stream);

/* Encoder self-attention on packed qkv [B][N][3E] ("b s (three h d)"), no mask -> out [B][N][E]
 * [nn/attention.py:112-136 MultiHeadAttention.forward]. */
int eamrl_mha_encoder(const float* qkv, float* out, int64_t B, int N, int E, int H, void* stream);

/* Gradient of eamrl_mha_encoder for the training graph (MultiHeadAttention.forward under loss.backward(),
 * nn/attention.py:112-136): qkv [B][N][3E] as given to the forward, dout [B][N][E] -> dqkv [B][N][3E] (written).  The softmax
 * is recomputed (hardware exp; tile-order sums: held to 1e-5 of torch's scaled_dot_product_attention gradient, not part of
 * the bit-exact path).  N <= 112, E = 128, H = 8 (eamrl_mha_encoder_backward_supported). */
int eamrl_mha_encoder_backward_supported(int N, int E, int H);
int eamrl_mha_encoder_backward(const float* qkv, const float* dout, float* dqkv, int64_t B, int N, int E, int H, void* stream);

/* Normalization.forward in place on x [B][N][E]  [nn/ops.py:32-56].
 * BATCH_EVAL uses running stats (mean, var); INSTANCE ignores them (may be NULL). */
int eamrl_normalize(float* x, int64_t B, int N, int E, int kind, const float* gamma, const float* beta,
                    const float* mean, const float* var, float eps, void* stream);

/* InstanceNorm1d(affine) of Normalization("instance") for the TRAINING graph  [nn/ops.py:32-56 as differentiated by
 * reinforce.py:103-106]: forward y = (x - mean) rstd gamma + beta on x, y [B][N][E] (out of place), keeping mean / rstd
 * [B][E]; backward dx [B][N][E] and dgamma / dbeta [E] (ACCUMULATED, may be NULL).  Plain fp32 (the rollout's own
 * normalisation is eamrl_normalize); within 1e-6 of torch.nn.functional.instance_norm and its autograd. */
int eamrl_instance_norm_forward(const float* x, float* y, float* mean, float* rstd, int64_t B, int N, int E, const float* gamma,
                                const float* beta, float eps, void* stream);
int eamrl_instance_norm_backward(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                                 float* dx, float* dgamma, float* dbeta, int64_t B, int N, int E, void* stream);

/* Normalization(batch) with BATCH statistics, in place on x [rows][E] -- BatchNorm1d in training mode, what the
 * reference's encoder runs under policy.train()  [nn/ops.py:45-47, SURVEY Appendix A9].  Mean and biased variance per
 * channel in a defined order (chunks of 128 rows summed sequentially, chunk sums ascending; variance = mean of
 * fma(d, d, .), d = x - mean), y = fma(x, gamma / sqrt(var + eps), beta - mean * scale).  save_mean / save_var [E]
 * receive the batch statistics; running_mean / running_var (both or neither) are updated as torch does:
 * running = (1 - momentum) * running + momentum * stat, the variance one unbiased.  ws: scratch of at least
 * ceil(rows / 128) * E floats (ws_floats = its size). */
int eamrl_batchnorm_train(float* x, int64_t rows, int E, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, float momentum, float eps, float* save_mean, float* save_var, float* ws,
                          int64_t ws_floats, void* stream);

/* ---- fused encoder: all MultiHeadAttentionLayers of an instance in one workgroup ------------------ */

/* Weights of one MultiHeadAttentionLayer  [nn/graph/attnnet.py:16-57]: the four Linear weights in the packed MFMA
 * fragment order produced by eamrl_pack_linear_weight, biases and normalisation parameters as they are.  The running
 * statistics are read for EAMRL_NORM_BATCH_EVAL only. */
typedef struct eamrl_encoder_layer {
    const float* Wqkv;     /* packed [3E][E]  (MultiHeadAttention.Wqkv.weight, "(three h d)" rows) */
    const float* bqkv;     /* [3E] */
    const float* Wo;       /* packed [E][E]   (out_proj.weight) */
    const float* bo;       /* [E] */
    const float* W1;       /* packed [F][E]   (MLP lins.0.weight) */
    const float* b1;       /* [F] */
    const float* W2;       /* packed [E][F]   (MLP lins.1.weight) */
    const float* b2;       /* [E] */
    const float *n1_gamma, *n1_beta, *n1_mean, *n1_var;     /* Normalization after the attention sub-layer */
    const float *n2_gamma, *n2_beta, *n2_mean, *n2_var;     /* Normalization after the feed-forward sub-layer */
} eamrl_encoder_layer;

/* Optional tail of eamrl_encoder_fused: AttentionModelDecoder._precompute_cache  [zoo/am/decoder.py:206-235] computed from
 * the final embeddings while they are still in LDS.  Wc: the stacked projection weights [nproj * E][E], packed --
 * glimpse key | glimpse value | logit key (project_node_embeddings) | first context half (| second context half), i.e.
 * slots 0 .. nproj-1 of the slot-major decoder cache; WoutT: project_out.weight TRANSPOSED ([E][E], packed), for the
 * folded logit key Lp = L Wout written to slot nproj.  out [B][M][ld] (ld >= (nproj + 1) E floats).  Same values as
 * eamrl_linear(h, Wc) followed by eamrl_matmul_right(L, Wout). */
typedef struct eamrl_encoder_cache {
    const float* Wc;
    const float* WoutT;
    float* out;
    int64_t ld;
    int nproj;          /* 4 (depot envs: K V L Pa) or 5 (TSP: K V L Pa Pb) */
    /* optional graph context  [zoo/am/decoder.py:221-226]: gctx [B][E] = embeddings.mean(1) project_fixed_context^T, the mean
     * in node order and the projection as a k-ordered chain (= eamrl_mean_nodes + eamrl_linear).  Wg: the PLAIN row-major
     * weight [E][E] (16-byte aligned).  Both NULL: not computed. */
    const float* Wg;
    float* gctx;
} eamrl_encoder_cache;

/* Optional head of eamrl_encoder_fused_init: the init embedding  [nn/env_embeddings/init.py:55-68 (TSP), 115-138 (VRP),
 * 141-157 (VRPTW), 227-286 (PCTSP, OP); zoo/am/encoder.py:86-87] computed straight into the workgroup's LDS instead of being
 * written to and read back from HBM.  Row n of instance b is Linear(F -> E) of feat[b][n][0..F-1] (W [E][F] row-major, b [E] or
 * NULL), each output the k-ordered fma chain of eamrl_linear; with depot != NULL row 0 is instead Linear(2 -> E) of the depot
 * coordinates depot[b * depot_ld + 0..1] (Wd [E][2], bd) and feat's row 0 is not read.  init_out (may be NULL): [B][M][E],
 * also receives the init embeddings (return_init_embeds). */
typedef struct eamrl_encoder_init {
    const float* feat;      /* [B][M][F] */
    int F;                  /* 1 .. 8 */
    const float* W;         /* [E][F] init_embed.weight */
    const float* b;         /* [E] init_embed.bias or NULL */
    const float* depot;     /* NULL (TSP) or the depot coordinates, instance stride depot_ld floats */
    int64_t depot_ld;
    const float* Wd;        /* [E][2] init_embed_depot.weight */
    const float* bd;        /* [E] or NULL */
    float* init_out;        /* NULL or [B][M][E] */
} eamrl_encoder_init;

/* Wp = W [out_dim][in_dim] (torch.nn.Linear.weight) re-ordered for the fused encoder: block (ct, u) of 256 floats holds,
 * for lane l = 16 g + j of a wavefront, the float4 { W[16 ct + j][16 u + 4 q + g] : q = 0..3 } -- the B operand of four
 * consecutive v_mfma_f32_16x16x4_f32 k-steps.  out_dim, in_dim multiples of 16.  Redo after every weight update. */
int eamrl_pack_linear_weight(const float* W, float* Wp, int out_dim, int in_dim, void* stream);

/* 1 if eamrl_encoder_fused handles this shape (M <= 112 nodes, E = 128, H = 8, feed-forward 512, <= 8 layers). */
int eamrl_encoder_fused_supported(int M, int E, int H, int ff_hidden, int nlayers);

/* GraphAttentionNetwork.forward  [nn/graph/attnnet.py:94-103; zoo/am/encoder.py:88-89]: h_out = layers(h_in) for
 * h_in / h_out [B][M][E] (the init embeddings in, the node embeddings out; may alias), one workgroup per instance,
 * activations resident in LDS, every Linear on fp32 MFMA.  Bit-identical to the sequence eamrl_linear /
 * eamrl_mha_encoder / eamrl_linear_bn (or eamrl_normalize) per layer.  norm: EAMRL_NORM_BATCH_EVAL or
 * EAMRL_NORM_INSTANCE (batch statistics are a cross-instance reduction: use the unfused calls + eamrl_batchnorm_train). */
int eamrl_encoder_fused(const float* h_in, float* h_out, int64_t B, int M, int E, int H, int ff_hidden, int nlayers, int norm,
                        float eps, const eamrl_encoder_layer* layers, const eamrl_encoder_cache* cache /* may be NULL */,
                        void* stream);

/* AttentionModelEncoder.forward in one launch  [zoo/am/encoder.py:70-91]: init embedding (eamrl_encoder_init) -> all layers ->
 * (optionally) the decoder cache.  h_out may be NULL when only the cache is wanted (the node embeddings then never leave
 * LDS: 2 x B*M*E*4 bytes of HBM traffic and the init-embedding launch less than eamrl_linear + eamrl_encoder_fused).
 * Same values as those two calls. */
int eamrl_encoder_fused_init(const eamrl_encoder_init* init, float* h_out /* may be NULL */, int64_t B, int M, int E, int H,
                             int ff_hidden, int nlayers, int norm, float eps, const eamrl_encoder_layer* layers,
                             const eamrl_encoder_cache* cache /* may be NULL */, void* stream);

/* out[b][e] = (sum_n emb[b][n][e]) / M   (embeddings.mean(1), zoo/am/decoder.py:225-227) */
int eamrl_mean_nodes(const float* emb, float* out, int64_t B, int M, int E, void* stream);

/* PointerAttention.forward in one launch -- the module the reference's decoder accepts through its constructor
 * injection point `AttentionModelDecoder(pointer=...)`  [zoo/am/decoder.py:82,109-124; nn/attention.py:282-328]:
 * logits[b][l][n] = (project_out(MHA(query, key, value, mask))[b][l] . logit_key[b][n]) / sqrt(E).
 * query [B][L][E]; key / value / logit_key [B][M][.] with row stride ld (>= E); mask (may be NULL) [B][M] bytes, or
 * [B][L][M] when mask_per_query, non-zero = may attend (applied when mask_inner); Wout [E][E] = project_out.weight
 * (y = x W^T), bout [E] or NULL; logits [B][L][M] (masked nodes keep their finite raw logit, as in the reference;
 * a row with no feasible node yields NaN, which the reference's check_nan assert reports). */
int eamrl_pointer_attention(const float* query, const float* key, const float* value, const float* logit_key, int64_t ld,
                            const uint8_t* mask, int mask_per_query, const float* Wout, const float* bout, float* logits,
                            int64_t B, int L, int M, int E, int H, int mask_inner, void* stream);

/* ---- teacher-forced re-evaluation: the gradient path of training ------------------------------------- */

/* All decode steps of finished rollouts at once, `policy(td, env, actions=...)` with decode type "evaluate"
 * [models/common/constructive/base.py:203-229, utils/decoding.py:452-465], forward and backward.  Rows are the rollout's
 * R = S * B rows in "(s b)" order (row r belongs to instance r % B); per (row, step) the caller supplies what the env state
 * would be (obtained by replaying the env transitions: eamrl_*_step_mask + eamrl_pack_mask_bits, or eamrl_tsp_mask_bits):
 *   maskbits [R][T][4]   bit n of the 128-bit word = node n feasible at that step (graphs up to 112 nodes)
 *   idxA / idxB [R][T]   node whose Pa / Pb row enters the context query, -1 = none (idxB may be NULL)
 *   sc [NC][R][T]        state scalars multiplying the state columns Cvec [NC][E] (free capacity, time, placeholder switch)
 * query = Pa[idxA] + Pb[idxB] + gctx + sum_k sc_k Cvec_k; the rest is the decode step of eamrl_am_decode_step with the same
 * weight folds (Lp = logit key times project_out).  Steps t < tstart (the multistart start column) get log-prob 0.
 * Not bit-exact by contract: hardware exp / log, tile-order sums; log-probs within 1e-5 of the rollout's. */
typedef struct eamrl_reeval {
    const float *K, *V, *Lp, *Pa, *Pb; int64_t ld;          /* [B][M][.] fp32, common row stride ld (floats), 16-byte aligned rows */
    const float* gctx; const float* Cvec; int NC;           /* [B][E] or NULL; [NC][E] (NC <= 4 forward, <= 2 backward) */
    const int32_t* idxA; const int32_t* idxB; const float* sc;
    const uint32_t* maskbits; const int64_t* actions;       /* actions [R][T] */
    int64_t B, R; int S, T, M, tstart, nchunk; float clip, temp;      /* nchunk: workgroups per instance (rows split) */
    float* logp; float* lse;                                /* [R][T]: forward writes both; backward reads lse -- or, with
                                                             * lse == NULL (no forward pass run), logp = the ROLLOUT's per-step
                                                             * log-probs of `actions`, from which the normaliser is recovered */
    const float* glogp; float* dheads;                      /* backward: dL/dlogp [R][T]; scratch [R][T][E], R * T < 2^31 */
    const float* heads; int heads_T;                        /* forward and backward, optional: the rollout's glimpse outputs [R][heads_T][E]
                                                             * (eamrl_state.heads_out) of exactly these actions, decode step
                                                             * t - tstart of row r at (r * heads_T + t - tstart) * E; NULL: recomputed */
    float* entropy;                                         /* forward, optional: [R][T] entropy of each step's distribution
                                                             * over the feasible nodes (calculate_entropy, utils/ops.py) */
    float *dK, *dV, *dLp, *dPa, *dPb; int64_t ldg;          /* gradients [B][M][.] (row stride ldg), ACCUMULATED into (+=) */
    float *dgctx, *dCvec;                                   /* [B][E] or NULL, [NC][E]; accumulated */
    /* SDVRP only (NULL otherwise): the dynamic embedding  [nn/env_embeddings/dynamic.py:59-78] -- every step adds
     * rem[n] * (wk | wv | wl) to row n of the glimpse key / value / logit key, rem = demand_with_depot at that step.
     * rem [R][T][128] (rows zero padded; eamrl_replay_states_sdvrp records them; [R][T][nkc][128] with key chunks),
     * dyn = wk | wv | lw [3][E] with lw = wl folded through project_out like Lp; ddyn [3][E] accumulated (backward).
     * Excludes `heads`. */
    const float* rem; const float* dyn; float* ddyn;
    /* Graphs above 112 nodes (M <= 1024; not with `heads` or a NULL lse): the keys are split into nkc = ceil(M / 112) chunks,
     * one workgroup per (instance, row chunk, key chunk); the softmax statistics of the glimpse and of the logits are combined
     * across the chunks by small kernels, and the backward uses rs = heads . dheads instead of a row sum over all keys.
     * nkc <= 1: the single-chunk kernels.  Otherwise maskbits is [R][T][nkc][4] (bit i of chunk c = node 112 c + i;
     * eamrl_tsp_mask_bits_chunked / eamrl_pack_mask_bits_chunked), `scratch` holds eamrl_reeval_scratch_floats(R, T, M) floats
     * that must survive from the forward to the backward call, and dheads is [nkc + nkc][R][T][E] (per-chunk partials, then the
     * per-chunk query gradients).  The mc_* fields are set by the library. */
    int nkc; float* scratch;
    int mc_koff, mc_mstride; float *mc_part_o, *mc_part_s, *mc_gstat, *mc_lpart, *mc_rsq, *mc_dq;
} eamrl_reeval;

int eamrl_reeval_supported(int M, int E, int H);                  /* 1 for M <= 112 (1024 with key chunks), E = 128, H = 8 */
int64_t eamrl_reeval_scratch_floats(int64_t R, int T, int M);     /* 0 for M <= 112 */
int eamrl_reeval_forward(const eamrl_reeval* p, void* stream);    /* -> logp, lse (lse may be NULL) [, entropy] */
int eamrl_reeval_backward(const eamrl_reeval* p, void* stream);   /* glogp, lse (or rollout logp) -> dK dV dLp dPa dPb dgctx dCvec */

/* bits[(r * T + t) * 4 + n / 32] bit (n % 32) = mask[r][n] for step t (call after every replayed env transition). */
int eamrl_pack_mask_bits(const uint8_t* mask, uint32_t* bits, int64_t R, int M, int T, int t, void* stream);
/* TSP: all T steps from the action rows (node n feasible at step t iff not among a_0 .. a_{t-1})  [tsp/env.py:62-88]. */
int eamrl_tsp_mask_bits(const int64_t* actions, uint32_t* bits, int64_t R, int M, int T, void* stream);
/* The same two in the chunked layout of graphs above 112 nodes: bits [R][T][nkc][4], nkc = ceil(M / 112), bit i of chunk c = node 112 c + i. */
int eamrl_pack_mask_bits_chunked(const uint8_t* mask, uint32_t* bits, int64_t R, int M, int T, int t, void* stream);
int eamrl_tsp_mask_bits_chunked(const int64_t* actions, uint32_t* bits, int64_t R, int M, int T, void* stream);

/* ---- per-step decode ------------------------------------------------------------------------------ */

/* The decoder cache of one batch (AttentionModelDecoder._precompute_cache, zoo/am/decoder.py:206-235),
 * with the two weight folds of DESIGN.md:  Pa/Pb = node embeddings already multiplied by the halves of
 * context_embedding.project_context (nn/env_embeddings/context.py:50-74,105-157) and Lp = logit key times
 * pointer.project_out.  All [B][M][*] with a common row stride `ld` (elements). */
typedef struct eamrl_cache {
    const float* K;    /* glimpse key   */
    const float* V;    /* glimpse value */
    const float* Lp;   /* logit key * project_out */
    const float* Pa;   /* TSP: first-node half; CVRP: current-node part */
    const float* Pb;   /* TSP: current-node half; CVRP: NULL */
    const float* cvec; /* [E]  TSP: project_context(W_placeholder); CVRP: capacity column of project_context;
                        * CVRPTW: [2][E] capacity and current-time columns */
    const float* gctx; /* [B][E] graph context or NULL (POMO: use_graph_context=False) */
    int64_t ld;        /* row stride of K/V/Lp/Pa/Pb in floats (>= E) */
    int64_t B;         /* instances */
    int32_t M, E, H;   /* nodes (incl. depot), embed dim, heads */
    const float* dyn;  /* SDVRP only, else NULL: [3][E] dynamic-embedding vectors wk | wv | lw (SDVRPDynamicEmbedding,
                        * nn/env_embeddings/dynamic.py:59-78; lw = the logit-key column of the projection times
                        * pointer.project_out).  The reference adds rem[n] * vector to row n of the cached key /
                        * value / logit key each step; the kernels fold that rank-1 update. */
} eamrl_cache;

/* Per-row rollout state (the TensorDict keys the reference's env keeps, tsp/env.py:105-115,
 * cvrp/env.py:110-130).  Unused members for an env may be NULL. */
typedef struct eamrl_state {
    int64_t* first;    /* [R] TSP first_node */
    int64_t* cur;      /* [R] current_node */
    int64_t* istep;    /* [R] TSP i */
    float* used;       /* [R] CVRP / SDVRP used_capacity; PCTSP cur_total_prize; OP tour_length */
    const float* vcap; /* [R] CVRP / SDVRP vehicle_capacity; PCTSP prize_required; OP max_length[:, 0] of the row's instance */
    const float* demand; /* [B][N] CVRP demand; PCTSP: [B][M] real_prize with a zero depot slot; OP: [B][M] max_length */
    uint8_t* mask;     /* [R][M] action_mask (1 = feasible) */
    uint8_t* visited;  /* [R][M] CVRP visited */
    uint8_t* done;     /* [R] */
    float* rem;        /* [R][M] SDVRP demand_with_depot (remaining demand), else NULL */
    const float* locs; /* [B][M][2] OP / CVRPTW node coordinates (depot first), else NULL */
    float* time;       /* [R] CVRPTW current_time, else NULL */
    const float* tw;   /* [B][M][2] CVRPTW time windows (start, end) as f32, else NULL */
    const float* dur;  /* [B][M] CVRPTW service durations, else NULL */
    float* heads_out;  /* optional, eamrl_am_rollout / _seeded only: [R][t_max][E], the glimpse output ("heads", the input of the
                        * logit projection, nn/attention.py:282-301) of every decode step -- the training graph's backward
                        * (eamrl_reeval.heads) then does not recompute it.  Written only where eamrl_rollout_rng_native() is 1
                        * (the start-sharing kernel); rows that are done, and steps after an instance's last, get zeros. */
} eamrl_state;

/* One decode step for R rows = AttentionModelDecoder.forward + DecodingStrategy.step
 * [zoo/am/decoder.py:161-198; nn/attention.py:282-328; utils/decoding.py:140-190,346-465]:
 * context query -> 8-head masked glimpse -> logits -> tanh clip -> mask -> /temperature -> [top-k] -> [top-p] ->
 * log_softmax -> greedy / sampling / evaluate.  top_k > 0: entries below the k-th largest scaled logit are dropped
 * (ties kept); 0 < top_p < 1: the lower tail whose running softmax mass (ascending order) is <= 1 - top_p is dropped
 * [utils/decoding.py:110-136,170-176]; 0 = off.  Reads the state, does not modify it unless fuse_env_step != 0, in which
 * case it also applies TSPEnv._step / CVRPEnv._step / SDVRPEnv._step (+mask) with the selected action.
 * EAMRL_ENV_SDVRP: cache->dyn and state->rem are required (dynamic embedding of the remaining demand, see eamrl_cache).
 * noise [R][M] (SAMPLE) / given [R] (EVALUATE) else NULL.  Outputs: action [R], logp [R];
 * optional logprobs_all [R][M] (store_all_logp) and logits_raw [R][M] (pre-clip decoder logits). */
int eamrl_am_decode_step(int env, const eamrl_cache* cache_host, const eamrl_state* state_host, int64_t R,
                         int mode, const float* noise, const int64_t* given, float tanh_clip, float temperature,
                         int top_k, float top_p, int fuse_env_step, int64_t* action, float* logp, float* logprobs_all,
                         float* logits_raw, uint32_t* status, void* stream);

/* Whole decode loop in one launch (ConstructivePolicy.forward's while-loop, constructive/base.py:236-250):
 * repeats {decode step, env step} until every row is done or t_max steps were taken.  actions/logps are
 * [R][t_max] (right-padded: finished CVRP rows keep selecting the depot, logp 0).  noise [R][t_max][M],
 * given [R][t_given].  steps_out (device int32): number of steps executed = max over rows.  With top_k / top_p
 * filtering the streaming kernel is used (the register-resident one does not filter). */
int eamrl_am_rollout(int env, const eamrl_cache* cache_host, const eamrl_state* state_host, int64_t R, int mode,
                     const float* noise, const int64_t* given, int t_given, float tanh_clip, float temperature,
                     int top_k, float top_p, int t_max, int64_t* actions, float* logps, int32_t* steps_out,
                     uint32_t* status, void* stream);

/* Sampling without a noise tensor.  The Exp(1) draw of (row r, step t, node n) is a pure function of the call's seed:
 * Philox4x32-10 on the counter (n / 4, t, r) with the seed as key, word n % 4 -> u = (2 (x >> 9) + 1) 2^-24 -> -log(u)
 * with the library's defined log (csrc/dmath.hpp; the CPU oracle has the same function).  eamrl_exp1_noise writes the
 * draws as a [R][T][M] tensor; eamrl_am_rollout_seeded == eamrl_am_rollout(EAMRL_SAMPLE) fed with that tensor, bit for
 * bit, but the start-sharing kernel computes the draws in place (a POMO batch of 1024 x 100 starts x 100 nodes would need
 * a 4 GB tensor).  eamrl_rollout_rng_native: 1 if this (env, cache shape, R) runs on a kernel with in-place noise
 * (noise_scratch may then be NULL); otherwise noise_scratch [R][t_max][M] is filled and the tensor path runs.
 * seed_dev (may be NULL): a device word XOR-ed into the seed when the kernel starts -- a launch recorded in a HIP graph gets a
 * fresh seed per replay from memory, since its by-value arguments are frozen at capture. */
int eamrl_exp1_noise(uint64_t seed, const uint64_t* seed_dev, float* noise, int64_t R, int T, int M, void* stream);
int eamrl_rollout_rng_native(int env, const eamrl_cache* cache_host, int64_t R);
int eamrl_am_rollout_seeded(int env, const eamrl_cache* cache_host, const eamrl_state* state_host, int64_t R, uint64_t seed,
                            const uint64_t* seed_dev, float* noise_scratch, float tanh_clip, float temperature, int t_max, int64_t* actions,
                            float* logps, int32_t* steps_out, uint32_t* status, void* stream);

/* ---- reward ----------------------------------------------------------------------------------------- */

/* reward[r] = -(closed tour length)  [utils/ops.py:59-95; tsp/env.py:152-159; cvrp/env.py:146-155].
 * locs [B][M][2]; actions [R][T]; with_depot prepends node 0 (CVRP). */
int eamrl_tour_length(const float* locs, const int64_t* actions, float* reward, int64_t R, int64_t B, int M, int T,
                      int with_depot, void* stream);

/* PCTSPEnv._get_reward  [pctsp/env.py:165-187]: reward[r] = sum_t penalty[a_t] - (tour length from / to the depot +
 * sum of all penalties).  locs [B][M][2], penalty [B][M] (zero depot slot), actions [R][T]. */
int eamrl_pctsp_reward(const float* locs, const float* penalty, const int64_t* actions, float* reward, int64_t R, int64_t B,
                       int M, int T, void* stream);

/* OPEnv._get_reward  [op/env.py:167-177]: reward[r] = sum_t prize[a_t] (lane tree over the steps).  prize [B][M]. */
int eamrl_op_reward(const float* prize, const int64_t* actions, float* reward, int64_t R, int64_t B, int M, int T,
                    void* stream);

/* OPEnv.check_solution_validity  [op/env.py:179-212]: bad[0] += rows with a customer visited twice, bad[1] += rows whose
 * closed length over the actions exceeds ((maxlen[n] + distance(depot, n)) + 1e-6) + 1e-5 for some node n. */
int eamrl_op_check_solution(const int64_t* actions, const float* locs, const float* maxlen, int64_t R, int64_t B, int M,
                            int T, int32_t* bad, void* stream);

/* out[r] = sum_t logp[r][t], sequential  (get_log_likelihood, utils/decoding.py:38-64) */
int eamrl_sum_logp(const float* logp, int64_t ld, float* out, int64_t R, int T, void* stream);

/* The epilogue of a TSP / CVRP rollout in one launch: reward[r] = get_reward [tsp/env.py:90-100, cvrp/env.py:146-155]
 * (as eamrl_tour_length), ll[r] = sum_t logp[r][t] (as eamrl_sum_logp), bad += check_solution_validity (as
 * eamrl_check_solution) -- each in the order of the single-purpose entry point, so the results are bit-identical.
 * locs [B][M][2] (CVRP: depot first, M = N + 1); demand [B][M-1], vcap [R] for CVRP; reward / ll / bad may be NULL. */
int eamrl_rollout_finish(int env, const float* locs, const int64_t* actions, const float* logp, int64_t ld,
                         const float* demand, const float* vcap, float* reward, float* ll, int32_t* bad, int64_t R,
                         int64_t B, int M, int T, void* stream);

/* n <= EAMRL_MULTI_COPY_MAX device-to-device copies (src[i] != NULL) or zero fills (src[i] == NULL) of bytes[i] bytes
 * in one launch: the state clones and output buffers of a rollout (the reference clones its TensorDict per call,
 * constructive/base.py:203-217) and the input refresh of a replayed HIP graph.  src / dst / bytes are HOST arrays. */
#define EAMRL_MULTI_COPY_MAX 16
int eamrl_multi_copy(int n, const void* const* src, void* const* dst, const int64_t* bytes, void* stream);

/* check_solution_validity on the device: bad[0] += invalid tours, bad[1] += over-capacity rows
 * [tsp/env.py:161-168; cvrp/env.py:157-185].  bad: device int32[2], caller zeroes.
 * EAMRL_ENV_PCTSP [pctsp/env.py:189-205]: demand = real_prize [B][N+1], vcap unused; bad[0] += rows with a customer
 * visited twice, bad[1] += rows that neither collect a total prize >= 1 - 1e-5 nor visit every customer.
 * EAMRL_ENV_SDVRP replays the deliveries [sdvrp/env.py:148-171]: bad[0] += rows with demand left at the end,
 * bad[1] += rows that visit the depot twice in a row while any entry of the replay's demand vector is nonzero. */
int eamrl_check_solution(int env, const int64_t* actions, const float* demand, const float* vcap, int64_t R,
                         int64_t B, int N, int T, int32_t* bad, void* stream);

/* The per-step inputs of eamrl_reeval_* for the depot envs (EAMRL_ENV_CVRP / _CVRPTW / _PCTSP / _OP) in one launch: starting
 * from the given state (read-only; the fields of eamrl_state that the env's step kernel uses, per row as there) the env
 * transitions of actions [R][T] are replayed, and BEFORE each step t are recorded: bits[r][t][4] the feasibility mask,
 * idxA[r][t] the current node, sc[0][r][t] = vcap - used (PCTSP: clamped at 0: prize still to collect; OP: length still
 * allowed) and, CVRPTW, sc[1][r][t] = the current time.  Same transitions as eamrl_*_step_mask (the same code), i.e. the
 * same result as T rounds of {eamrl_pack_mask_bits, copy, eamrl_*_step_mask}.  M <= 128. 
 * Graphs above 112 nodes (M <= 1024): bits is [R][T][nkc][4] in the chunked layout of eamrl_reeval (nkc = ceil(M / 112)). */
int eamrl_replay_states(int env, const eamrl_state* state, int64_t R, int64_t B, int M, const int64_t* actions, int T,
                        uint32_t* bits, int32_t* idxA, float* sc, void* stream);
/* The same for SDVRP  [sdvrp/env.py:58-92,137-146]: state->rem [R][M] (demand_with_depot), used, vcap, cur; additionally
 * rem_out [R][T][128] = the remaining demands before each step (the rows the dynamic embedding reads), zero padded.
 * Graphs above 112 nodes (M <= 1024): bits [R][T][nkc][4] and rem_out [R][T][nkc][128] in the chunked layout (node n at chunk
 * n / 112, slot n % 112); rem_out must arrive zero-filled. */
int eamrl_replay_states_sdvrp(const eamrl_state* state, int64_t R, int M, const int64_t* actions, int T, uint32_t* bits,
                              int32_t* idxA, float* sc, float* rem_out, void* stream);

/* ---- beam search ------------------------------------------------------------------------------------------ */

/* BeamSearch._make_beam_step  [rl4co/utils/decoding.py:573-608].  Rows in "(w b)" order, R = beam_width * B.
 * logprobs [R][M] f32 (the step's log-probs of every row, -inf where masked), parent [R] f32 (cumulative log-prob of
 * each beam).  For every instance the beam_width largest logprobs[w*B+b][n] + parent[w*B+b], descending, ties to
 * the lower w*M + n; output row k*B + b: node [R] i64, beam [R] i32 (parent beam w), cum [R] f32 (new cumulative
 * log-prob), step_logp [R] f32 (= logprobs of the chosen (beam, node)).  beam_width * M <= 36000. */
int eamrl_beam_topk(const float* logprobs, const float* parent, int64_t B, int beam_width, int M, int64_t* node,
                    int32_t* beam, float* cum, float* step_logp, void* stream);

/* ---- evolutionary improvement (the fork's EA) ---------------------------------------------------------------- */

/* EA.run for TSP populations  [rl4co/models/zoo/earl/evolution.py:252-354; operators :356-362 (fitness),
 * :1103-1108 (elitism_selection), :392-488 (order_crossover_tsp), :490-517 (inverse_mutate_tsp)].
 * One population per instance, all generations in one launch; replaces the per-instance CPU thread pool of
 * evolution_worker (:28-123).
 *   locs [B][N][2] f32;  pop [B][S][N] i64, in: initial tours, out: evolved tours;  fitness [B][S] f32 out
 *   (= f32(1.5*N) - tour length).  S, N <= 128.
 * Elites: ne = S if S <= 2 else int(selection_rate*S) (0 -> S); pairs P = ne/2; offspring O = 2P per generation.
 * Replacement: if the first nodes of the initial population are pairwise distinct, position s keeps the best of
 * itself and the offspring starting at its first node; otherwise the S fittest of population ++ offspring.
 * The reference draws inside the operators with numba's per-thread np.random; here the draws are inputs:
 *   cross_rand [G][B][P] f64 uniforms (pair 0 always crosses, the others with the adjusted rate),
 *   cross_idx  [G][B][P][2] i32 cut points in [1, N)   (read only where the pair crosses),
 *   mut_rand   [G][B][O] f64 uniforms,  mut_idx [G][B][O][2] i32 in [1, N)  (read only where mut_rand < rate).
 * crossover_rate: pass the float32-rounded value the reference's signature implies. */
int eamrl_ea_tsp_run(const float* locs, int64_t* pop, float* fitness, int64_t B, int S, int N, int num_generations,
                     double mutation_rate, double crossover_rate, double selection_rate, const double* cross_rand,
                     const int32_t* cross_idx, const double* mut_rand, const int32_t* mut_idx, void* stream);

/* EA.run for CVRP populations  [evolution.py:252-354; :364-370 (fitness = f32(2.5*L) - cost), :585-788
 * (order_crossover_cvrp: the parent's first `end` routes, then the missing customers in index order, split by a
 * float64 load against the capacity), :519-553 (inverse_mutate_cvrp: reverse a segment inside one route)].
 *   locs [B][N+1][2] f32, depot first;  demand [B][N] f32 (normalised);  vcap [B] f32;
 *   pop [B][S][L] i64 chromosomes = action rows, 0 = depot, zero-padded; in/out;  fitness [B][S] f32 out.
 *   S <= 128, N <= 127, L <= 256, S*L <= 24000.  top_k != 0: the `method == "am"` replacement (S fittest of
 *   population ++ offspring) even when the start nodes are distinct.
 * An initial mutation pass precedes the generations (EA.run :273-274).  Draws are uniforms u in [0, 1);
 * randint(lo, hi) = lo + min(floor(u*(hi-lo)), hi-lo-1):
 *   init_mut_rand [B][S], init_mut_u [B][S][3] (route, segment start, segment end);
 *   cross_rand [G][B][P], cross_u [G][B][P] (number of routes kept);  mut_rand [G][B][O], mut_u [G][B][O][3]. */
int eamrl_ea_cvrp_run(const float* locs, const float* demand, const float* vcap, int64_t* pop, float* fitness, int64_t B,
                      int S, int N, int L, int num_generations, double mutation_rate, double crossover_rate,
                      double selection_rate, int top_k, const double* init_mut_rand, const double* init_mut_u,
                      const double* cross_rand, const double* cross_u, const double* mut_rand, const double* mut_u,
                      void* stream);

/* EA.run for PCTSP (env = EAMRL_ENV_PCTSP) and OP (env = EAMRL_ENV_OP) populations  [evolution.py:252-354;
 *  PCTSP: :364-370 (fitness = f32(2.5*L) - cost, cost = -PCTSPEnv reward), :905-1101 (cycle_crossover_pctsp: cycles of the
 *  pair found from the first parent's side, children deduplicated and topped up by descending float32 prize/penalty
 *  ratio until the float64 prize sum reaches 1 - 1e-5), :555-583 (inverse_mutate_pctsp: reverse [i1, i2) inside the
 *  visited prefix, or swap two neighbours when the two draws coincide);
 *  OP: :372-378 (fitness = collected prize), :1110-1346 (order_crossover_op: acts only on parents that are all zeros
 *  from index 1 on; the first `end` entries, then customers 1..L in index order while route + leg + way back fits
 *  max - 0.1, post-checked against max - 1e-5), :1468-1572 (inverse_mutate_op: reverse [s, e] if the float64 route
 *  length of float32 distances stays within max - 1e-5)].
 *   locs [B][N+1][2] f32, depot first;  prize [B][N+1] f32 with the depot's 0 first (PCTSP: td["real_prize"], OP:
 *   td["prize"]);  aux [B][N+1] f32: PCTSP td["penalty"], OP td["max_length"] (only entry 0 of an instance is used);
 *   pop [B][S][L] i64 action rows, 0 = depot, zero-padded; in/out;  fitness [B][S] f32 out.
 *   S <= 128, N <= 127, L <= 128.  top_k as in eamrl_ea_cvrp_run.
 * Defined where the reference is not: the cycle crossover's `next(iter(set))` starts at the smallest remaining node;
 * ties in a sort keep index order; customers beyond the node count are skipped by the OP crossover.
 * An initial mutation pass precedes the generations.  Draws are uniforms in [0, 1), randint as in eamrl_ea_cvrp_run:
 *   init_mut_rand [B][S], init_mut_u [B][S][2];  cross_rand [G][B][P], cross_u [G][B][P] (OP: the cut; PCTSP: unused,
 *   may be NULL);  mut_rand [G][B][O], mut_u [G][B][O][2]. */
int eamrl_ea_prize_run(int env, const float* locs, const float* prize, const float* aux, int64_t* pop, float* fitness,
                       int64_t B, int S, int N, int L, int num_generations, double mutation_rate, double crossover_rate,
                       double selection_rate, int top_k, const double* init_mut_rand, const double* init_mut_u,
                       const double* cross_rand, const double* cross_u, const double* mut_rand, const double* mut_u,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EAMRL_H */
