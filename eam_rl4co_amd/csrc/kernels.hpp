// Launch entry points shared between the kernel translation units and abi.hip.
#pragma once
#include "decode_common.hpp"

namespace eamrl {

struct GemmArgs {
    const float* x; int64_t ldx;
    const float* W; int64_t ldw; int wt;     // wt: W is [in][out] (right-multiplication)
    const float* bias; const float* res; int64_t ldres;
    float* y; int64_t ldy;
    int64_t rows; int in_dim, out_dim, relu;
    // optional fused BatchNorm1d(eval) on the output columns (applied after residual): all null or all set
    const float* bn_gamma; const float* bn_beta; const float* bn_mean; const float* bn_var; float bn_eps;
};

extern int g_debug[16];   // eamrl_debug_set knobs

int launch_linear(const GemmArgs& g, hipStream_t st);
int launch_mha_encoder(const float* qkv, float* out, int64_t B, int N, int E, int H, hipStream_t st);
int launch_ea_cvrp(const float* locs, const float* demand, const float* vcap, int64_t* pop, float* fitness, int64_t B, int S,
                   int N, int L, int G, double mutation_rate, double crossover_rate, double selection_rate, int top_k,
                   const double* init_mut_rand, const double* init_mut_u, const double* cross_rand, const double* cross_u,
                   const double* mut_rand, const double* mut_u, hipStream_t st);
int launch_ea_prize(int env, const float* locs, const float* prize, const float* aux, int64_t* pop, float* fitness, int64_t B,
                    int S, int N, int L, int G, double mutation_rate, double crossover_rate, double selection_rate, int top_k,
                    const double* init_mut_rand, const double* init_mut_u, const double* cross_rand, const double* cross_u,
                    const double* mut_rand, const double* mut_u, hipStream_t st);
int launch_normalize(float* x, int64_t B, int N, int E, int kind, const float* gamma, const float* beta,
                     const float* mean, const float* var, float eps, hipStream_t st);
int launch_instnorm_train_fwd(const float* x, float* y, float* mean, float* rstd, int64_t B, int N, int E, const float* gamma,
                              const float* beta, float eps, hipStream_t st);
int launch_instnorm_train_bwd(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma, float* dx,
                              float* dgamma, float* dbeta, int64_t B, int N, int E, hipStream_t st);
int launch_batchnorm_train(float* x, int64_t rows, int E, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float momentum, float eps, float* save_mean, float* save_var, float* ws,
                           hipStream_t st);
int launch_pointer_attention(const float* q, const float* K, const float* V, const float* Lk, int64_t ld, const uint8_t* mask,
                             int mask_per_query, const float* Wout, const float* bout, float* logits, int64_t B, int L, int M,
                             int E, int H, int mask_inner, hipStream_t st);
int launch_augment_xy(const float* xy, const float* cs, const int32_t* code, float* out, int64_t R, int64_t B, int N, float offset,
                      hipStream_t st);
bool mha_encoder_mfma_supports(int N, int E, int H);
int launch_mha_encoder_mfma(const float* qkv, float* out, int64_t B, int N, int E, int H, hipStream_t st);
bool encoder_fused_supports(int M, int E, int H, int FFdim, int nlayers);
int launch_encoder_fused(const float* h_in, float* h_out, int64_t B, int M, int nlayers, int norm, float eps,
                         const eamrl_encoder_layer* layers, const eamrl_encoder_cache* cache, const eamrl_encoder_init* init,
                         hipStream_t st);
int launch_pack_mfma_b(const float* W, float* Wp, int N, int K, hipStream_t st);
// BatchNorm-train backward and the tiny-K Linear weight gradient (train_norm.hip)
int64_t batchnorm_backward_scratch(int64_t rows, int E);
int launch_batchnorm_backward(const float* x, const float* dy, const float* mean, const float* var, const float* gamma, float eps,
                              int64_t rows, int E, float* dx, float* dgamma, float* dbeta, float* ws, hipStream_t st);
int64_t small_linear_wgrad_scratch(int64_t rows, int out_dim);
int launch_small_linear_wgrad(const float* dy, int64_t ldy, const float* x, int64_t ldx, int64_t rows, int out_dim, int K, float* dW,
                              float* db, float* ws, hipStream_t st);
// Linear weight gradient (train_gemm.hip)
bool linear_wgrad_supports(int out_dim, int in_dim);
int64_t linear_wgrad_scratch(int64_t rows, int out_dim, int in_dim);
int launch_linear_wgrad(const float* dy, int64_t ldy, const float* x, int64_t ldx, int64_t rows, int out_dim, int in_dim,
                        float* dW, float* db, float* scratch, hipStream_t st);
// teacher-forced re-evaluation (reeval.hip)
typedef eamrl_reeval ReevalArgs;
bool reeval_supports(int M, int E, int H);
int64_t reeval_scratch_floats(int64_t R, int T, int M);
int launch_pack_mask_bits_chunked(const uint8_t* mask, uint32_t* bits, int64_t R, int M, int T, int t, hipStream_t st);
int launch_tsp_mask_bits_chunked(const int64_t* actions, uint32_t* bits, int64_t R, int M, int T, hipStream_t st);
int launch_reeval_fwd(const ReevalArgs& a, hipStream_t st);
int launch_reeval_bwd(const ReevalArgs& a, hipStream_t st);
bool mha_encoder_bwd_supports(int N, int E, int H);
int launch_mha_encoder_bwd(const float* qkv, const float* dout, float* dqkv, int64_t B, int N, hipStream_t st);
int launch_pack_mask_bits(const uint8_t* mask, uint32_t* bits, int64_t R, int M, int T, int t, hipStream_t st);
int launch_tsp_mask_bits(const int64_t* actions, uint32_t* bits, int64_t R, int M, int T, hipStream_t st);
int launch_mean_nodes(const float* emb, float* out, int64_t B, int M, int E, hipStream_t st);
int launch_tsp_step(uint8_t* mask, int64_t* first, int64_t* cur, int64_t* istep, const int64_t* action,
                    uint8_t* done, int64_t R, int N, hipStream_t st);
int launch_cvrp(int step, uint8_t* visited, float* used, const float* vcap, const float* demand, int64_t* cur,
                const int64_t* action, uint8_t* mask, uint8_t* done, int64_t R, int64_t B, int N, hipStream_t st);
int launch_sdvrp(float* rem, float* used, const float* vcap, int64_t* cur, const int64_t* action, uint8_t* mask,
                 uint8_t* done, int64_t R, int M, hipStream_t st);
int launch_pctsp(uint8_t* visited, float* prize_tot, float* pen_tot, const float* prize, const float* penalty, int64_t* cur,
                 int64_t* istep, const int64_t* action, uint8_t* mask, uint8_t* done, int64_t R, int64_t B, int M,
                 hipStream_t st);
int launch_cvrptw(uint8_t* visited, float* used, const float* vcap, const float* demand, int64_t* cur, float* time,
                  const float* locs, const float* tw, const float* dur, const int64_t* action, uint8_t* mask, uint8_t* done,
                  int64_t R, int64_t B, int N, hipStream_t st);
int launch_cvrptw_check(const int64_t* actions, const float* locs, const float* tw, const float* dur, int64_t R, int64_t B,
                        int M, int T, int32_t* bad, hipStream_t st);
int launch_op(uint8_t* visited, float* tour_len, float* prize_tot, const float* prize, const float* locs, const float* maxlen,
              int64_t* cur, int64_t* istep, const int64_t* action, uint8_t* mask, uint8_t* done, int64_t R, int64_t B, int M,
              hipStream_t st);
int launch_op_reward(const float* prize, const int64_t* actions, float* reward, int64_t R, int64_t B, int M, int T,
                     hipStream_t st);
int launch_op_check(const int64_t* actions, const float* locs, const float* maxlen, int64_t R, int64_t B, int M, int T,
                    int32_t* bad, hipStream_t st);
int launch_tour_length(const float* locs, const int64_t* actions, float* reward, int64_t R, int64_t B, int M, int T,
                       int with_depot, hipStream_t st, const float* penalty = nullptr);
int launch_sum_logp(const float* logp, int64_t ld, float* out, int64_t R, int T, hipStream_t st);
int launch_rollout_finish(int env, const float* locs, const int64_t* actions, const float* logp, int64_t ld, const float* demand,
                          const float* vcap, float* reward, float* ll, int32_t* bad, int64_t R, int64_t B, int M, int T,
                          hipStream_t st);
int launch_replay_sdvrp(const float* rem, const float* used, const float* vcap, const int64_t* cur, const int64_t* actions,
                        uint32_t* bits, int32_t* idxA, float* sc, float* rem_out, int64_t R, int M, int T, hipStream_t st);
int launch_replay_states(int env, const uint8_t* mask, const uint8_t* visited, const float* used, const float* vcap,
                         const int64_t* cur, const int64_t* istep, const float* time, const float* demand, const float* locs,
                         const float* tw, const float* dur, const int64_t* actions, uint32_t* bits, int32_t* idxA, float* sc,
                         int64_t R, int64_t B, int M, int T, hipStream_t st);
int launch_multi_copy(int n, const void* const* src, void* const* dst, const int64_t* bytes, hipStream_t st);
int launch_check_solution(int env, const int64_t* actions, const float* demand, const float* vcap, int64_t R, int64_t B,
                          int N, int T, int32_t* bad, hipStream_t st);
int launch_beam_topk(const float* logprobs, const float* parent, int64_t B, int BW, int M, int64_t* node, int32_t* beam,
                     float* cum, float* step_lp, hipStream_t st);
int launch_decode_step(int env, const DecArgs& a, hipStream_t st);
int launch_rollout_stream(int env, const DecArgs& a, hipStream_t st);
void launch_rollout_pad(int env, const DecArgs& a, hipStream_t st);   // final state of rows that finished early
int launch_rollout_resident(int env, const DecArgs& a, hipStream_t st);
bool rollout_resident_supports(int env, const DecArgs& a);
bool rollout_ms_mfma_supports(int env, const DecArgs& a, bool shape_only = false);
int launch_rollout_ms_mfma(int env, const DecArgs& a, hipStream_t st);
int launch_exp1_noise(uint64_t seed, const uint64_t* seed_dev, float* noise, int64_t R, int T, int M, hipStream_t st);
int launch_ea_tsp(const float* locs, int64_t* pop, float* fitness, int64_t B, int S, int N, int G, double mutation_rate,
                  double crossover_rate, double selection_rate, const double* cross_rand, const int32_t* cross_idx,
                  const double* mut_rand, const int32_t* mut_idx, hipStream_t st);

}  // namespace eamrl
