// extern "C" boundary of libeamrl_hip.so (see include/eamrl.h).  Argument validation happens here,
// before anything is enqueued: a kernel that faults can reset the whole GPU host, so every shape the
// kernels index with is checked against what they assume.
#include <cstdarg>
#include <cstdio>

#include "kernels.hpp"


using namespace eamrl;

namespace eamrl { int g_debug[16] = {0}; }

static thread_local char g_err[256] = "";

static int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int launched(int rc, const char* what)
{
    if (rc == 0) return 0;
    if (rc == EAMRL_E_ARG) return fail(rc, "%s: unsupported shape", what);
    return fail(rc, "%s: launch failed: %s", what, hipGetErrorString(hipGetLastError()));
}

#define REQUIRE(cond, what)                                                        \
    do {                                                                           \
        if (!(cond)) return fail(EAMRL_E_ARG, "%s: requirement failed: %s", what, #cond); \
    } while (0)

extern "C" {

__attribute__((visibility("default"))) int eamrl_version(void) { return EAMRL_VERSION; }
__attribute__((visibility("default"))) const char* eamrl_last_error(void) { return g_err; }
__attribute__((visibility("default"))) int eamrl_debug_set(int key, int value)
{
    if (key < 0 || key >= 16) return fail(EAMRL_E_ARG, "eamrl_debug_set: unknown key %d", key);
    g_debug[key] = value;
    return 0;
}

__attribute__((visibility("default"))) int eamrl_tsp_step(uint8_t* mask, int64_t* first, int64_t* cur, int64_t* istep,
                                                         const int64_t* action, uint8_t* done, int64_t R, int N,
                                                         void* stream)
{
    REQUIRE(mask && first && cur && istep && action && done, "eamrl_tsp_step");
    REQUIRE(R >= 0 && N > 0, "eamrl_tsp_step");
    if (R == 0) return 0;
    return launched(launch_tsp_step(mask, first, cur, istep, action, done, R, N, (hipStream_t)stream), "eamrl_tsp_step");
}

__attribute__((visibility("default"))) int eamrl_cvrp_mask(const uint8_t* visited, const float* used, const float* vcap,
                                                          const float* demand, const int64_t* cur, uint8_t* mask,
                                                          int64_t R, int64_t B, int N, void* stream)
{
    REQUIRE(visited && used && vcap && demand && cur && mask, "eamrl_cvrp_mask");
    REQUIRE(R >= 0 && B > 0 && N > 0, "eamrl_cvrp_mask");
    if (R == 0) return 0;
    return launched(launch_cvrp(0, const_cast<uint8_t*>(visited), const_cast<float*>(used), vcap, demand,
                                const_cast<int64_t*>(cur), nullptr, mask, nullptr, R, B, N, (hipStream_t)stream),
                    "eamrl_cvrp_mask");
}

__attribute__((visibility("default"))) int eamrl_cvrp_step_mask(uint8_t* visited, float* used, const float* vcap,
                                                               const float* demand, int64_t* cur, const int64_t* action,
                                                               uint8_t* mask, uint8_t* done, int64_t R, int64_t B, int N,
                                                               void* stream)
{
    REQUIRE(visited && used && vcap && demand && cur && action && mask && done, "eamrl_cvrp_step_mask");
    REQUIRE(R >= 0 && B > 0 && N > 0, "eamrl_cvrp_step_mask");
    if (R == 0) return 0;
    return launched(launch_cvrp(1, visited, used, vcap, demand, cur, action, mask, done, R, B, N, (hipStream_t)stream),
                    "eamrl_cvrp_step_mask");
}

__attribute__((visibility("default"))) int eamrl_sdvrp_step_mask(float* rem, float* used, const float* vcap, int64_t* cur,
                                                                const int64_t* action, uint8_t* mask, uint8_t* done,
                                                                int64_t R, int M, void* stream)
{
    REQUIRE(rem && used && vcap && cur && mask, "eamrl_sdvrp_step_mask");
    REQUIRE(!action || done, "eamrl_sdvrp_step_mask");
    REQUIRE(R >= 0 && M >= 2, "eamrl_sdvrp_step_mask");
    if (R == 0) return 0;
    return launched(launch_sdvrp(rem, used, vcap, cur, action, mask, done, R, M, (hipStream_t)stream),
                    "eamrl_sdvrp_step_mask");
}

__attribute__((visibility("default"))) int eamrl_pctsp_step_mask(uint8_t* visited, float* prize_tot, float* pen_tot,
                                                                const float* prize, const float* penalty, int64_t* cur,
                                                                int64_t* istep, const int64_t* action, uint8_t* mask,
                                                                uint8_t* done, int64_t R, int64_t B, int M, void* stream)
{
    REQUIRE(visited && prize_tot && mask, "eamrl_pctsp_step_mask");
    REQUIRE(!action || (prize && cur && istep && done), "eamrl_pctsp_step_mask");
    REQUIRE(!pen_tot || penalty, "eamrl_pctsp_step_mask");
    REQUIRE(R >= 0 && B > 0 && M >= 2, "eamrl_pctsp_step_mask");
    if (R == 0) return 0;
    return launched(launch_pctsp(visited, prize_tot, pen_tot, prize, penalty, cur, istep, action, mask, done, R, B, M,
                                 (hipStream_t)stream), "eamrl_pctsp_step_mask");
}

__attribute__((visibility("default"))) int eamrl_cvrptw_step_mask(uint8_t* visited, float* used, const float* vcap,
                                                                 const float* demand, int64_t* cur, float* time,
                                                                 const float* locs, const float* tw, const float* dur,
                                                                 const int64_t* action, uint8_t* mask, uint8_t* done,
                                                                 int64_t R, int64_t B, int N, void* stream)
{
    REQUIRE(visited && used && vcap && demand && cur && time && locs && tw && mask, "eamrl_cvrptw_step_mask");
    REQUIRE(!action || (dur && done), "eamrl_cvrptw_step_mask");
    REQUIRE(R >= 0 && B > 0 && N > 0, "eamrl_cvrptw_step_mask");
    if (R == 0) return 0;
    return launched(launch_cvrptw(visited, used, vcap, demand, cur, time, locs, tw, dur, action, mask, done, R, B, N,
                                  (hipStream_t)stream), "eamrl_cvrptw_step_mask");
}

__attribute__((visibility("default"))) int eamrl_cvrptw_check_time(const int64_t* actions, const float* locs, const float* tw,
                                                                  const float* dur, int64_t R, int64_t B, int M, int T,
                                                                  int32_t* bad, void* stream)
{
    REQUIRE(actions && locs && tw && dur && bad && R >= 0 && B > 0 && M >= 2 && T > 0, "eamrl_cvrptw_check_time");
    if (R == 0) return 0;
    return launched(launch_cvrptw_check(actions, locs, tw, dur, R, B, M, T, bad, (hipStream_t)stream),
                    "eamrl_cvrptw_check_time");
}

__attribute__((visibility("default"))) int eamrl_op_step_mask(uint8_t* visited, float* tour_len, float* prize_tot,
                                                             const float* prize, const float* locs, const float* maxlen,
                                                             int64_t* cur, int64_t* istep, const int64_t* action,
                                                             uint8_t* mask, uint8_t* done, int64_t R, int64_t B, int M,
                                                             void* stream)
{
    REQUIRE(visited && tour_len && locs && maxlen && cur && mask, "eamrl_op_step_mask");
    REQUIRE(!action || (istep && done), "eamrl_op_step_mask");
    REQUIRE(!prize_tot || prize, "eamrl_op_step_mask");
    REQUIRE(R >= 0 && B > 0 && M >= 2, "eamrl_op_step_mask");
    if (R == 0) return 0;
    return launched(launch_op(visited, tour_len, prize_tot, prize, locs, maxlen, cur, istep, action, mask, done, R, B, M,
                              (hipStream_t)stream), "eamrl_op_step_mask");
}

__attribute__((visibility("default"))) int eamrl_op_reward(const float* prize, const int64_t* actions, float* reward,
                                                          int64_t R, int64_t B, int M, int T, void* stream)
{
    REQUIRE(prize && actions && reward && R >= 0 && B > 0 && M >= 2 && T > 0, "eamrl_op_reward");
    if (R == 0) return 0;
    return launched(launch_op_reward(prize, actions, reward, R, B, M, T, (hipStream_t)stream), "eamrl_op_reward");
}

__attribute__((visibility("default"))) int eamrl_op_check_solution(const int64_t* actions, const float* locs,
                                                                  const float* maxlen, int64_t R, int64_t B, int M, int T,
                                                                  int32_t* bad, void* stream)
{
    REQUIRE(actions && locs && maxlen && bad && R >= 0 && B > 0 && M >= 2 && M <= 4096 && T > 0, "eamrl_op_check_solution");
    if (R == 0) return 0;
    return launched(launch_op_check(actions, locs, maxlen, R, B, M, T, bad, (hipStream_t)stream), "eamrl_op_check_solution");
}

__attribute__((visibility("default"))) int eamrl_pctsp_reward(const float* locs, const float* penalty, const int64_t* actions,
                                                             float* reward, int64_t R, int64_t B, int M, int T,
                                                             void* stream)
{
    REQUIRE(locs && penalty && actions && reward && R >= 0 && B > 0 && M >= 2 && T > 0, "eamrl_pctsp_reward");
    if (R == 0) return 0;
    return launched(launch_tour_length(locs, actions, reward, R, B, M, T, 1, (hipStream_t)stream, penalty),
                    "eamrl_pctsp_reward");
}

__attribute__((visibility("default"))) int eamrl_linear(const float* x, int64_t ldx, const float* W, int64_t ldw,
                                                       const float* bias, const float* res, int64_t ldres, float* y,
                                                       int64_t ldy, int64_t rows, int in_dim, int out_dim, int relu,
                                                       void* stream)
{
    REQUIRE(x && W && y, "eamrl_linear");
    REQUIRE(rows >= 0 && in_dim > 0 && out_dim > 0, "eamrl_linear");
    REQUIRE(ldx >= in_dim && ldw >= in_dim && ldy >= out_dim && (!res || ldres >= out_dim), "eamrl_linear");
    REQUIRE((rows + 127) / 128 <= 0x7fffffffLL, "eamrl_linear");
    GemmArgs g{x, ldx, W, ldw, 0, bias, res, ldres, y, ldy, rows, in_dim, out_dim, relu, nullptr, nullptr, nullptr, nullptr, 0.f};
    return launched(launch_linear(g, (hipStream_t)stream), "eamrl_linear");
}

__attribute__((visibility("default"))) int eamrl_linear_bn(const float* x, int64_t ldx, const float* W, int64_t ldw,
                                                          const float* bias, const float* res, int64_t ldres, float* y,
                                                          int64_t ldy, int64_t rows, int in_dim, int out_dim,
                                                          const float* gamma, const float* beta, const float* mean,
                                                          const float* var, float eps, void* stream)
{
    REQUIRE(x && W && y && gamma && beta && mean && var, "eamrl_linear_bn");
    REQUIRE(rows >= 0 && in_dim > 0 && out_dim > 0, "eamrl_linear_bn");
    REQUIRE(ldx >= in_dim && ldw >= in_dim && ldy >= out_dim && (!res || ldres >= out_dim), "eamrl_linear_bn");
    REQUIRE((rows + 127) / 128 <= 0x7fffffffLL, "eamrl_linear_bn");
    GemmArgs g{x, ldx, W, ldw, 0, bias, res, ldres, y, ldy, rows, in_dim, out_dim, 0, gamma, beta, mean, var, eps};
    return launched(launch_linear(g, (hipStream_t)stream), "eamrl_linear_bn");
}

__attribute__((visibility("default"))) int eamrl_matmul_right(const float* x, int64_t ldx, const float* Wt, float* y,
                                                             int64_t ldy, int64_t rows, int in_dim, int out_dim,
                                                             void* stream)
{
    REQUIRE(x && Wt && y, "eamrl_matmul_right");
    REQUIRE(rows >= 0 && in_dim > 0 && out_dim > 0 && ldx >= in_dim && ldy >= out_dim, "eamrl_matmul_right");
    GemmArgs g{x, ldx, Wt, out_dim, 1, nullptr, nullptr, 0, y, ldy, rows, in_dim, out_dim, 0, nullptr, nullptr, nullptr, nullptr, 0.f};
    return launched(launch_linear(g, (hipStream_t)stream), "eamrl_matmul_right");
}

__attribute__((visibility("default"))) int eamrl_mha_encoder_backward_supported(int N, int E, int H)
{
    return mha_encoder_bwd_supports(N, E, H) ? 1 : 0;
}

__attribute__((visibility("default"))) int eamrl_mha_encoder_backward(const float* qkv, const float* dout, float* dqkv, int64_t B,
                                                                     int N, int E, int H, void* stream)
{
    REQUIRE(qkv && dout && dqkv && B >= 0, "eamrl_mha_encoder_backward");
    REQUIRE(mha_encoder_bwd_supports(N, E, H), "eamrl_mha_encoder_backward");
    REQUIRE((uintptr_t)qkv % 16 == 0 && (uintptr_t)dout % 16 == 0 && (uintptr_t)dqkv % 16 == 0, "eamrl_mha_encoder_backward");
    return launched(launch_mha_encoder_bwd(qkv, dout, dqkv, B, N, (hipStream_t)stream), "eamrl_mha_encoder_backward");
}

__attribute__((visibility("default"))) int64_t eamrl_linear_wgrad_scratch(int64_t rows, int out_dim, int in_dim)
{
    if (rows < 0 || !linear_wgrad_supports(out_dim, in_dim)) return -1;
    return linear_wgrad_scratch(rows, out_dim, in_dim);
}

__attribute__((visibility("default"))) int eamrl_linear_wgrad(const float* dy, int64_t ldy, const float* x, int64_t ldx,
                                                             int64_t rows, int out_dim, int in_dim, float* dW, float* db,
                                                             float* scratch, int64_t scratch_floats, void* stream)
{
    REQUIRE(dy && x && dW && scratch, "eamrl_linear_wgrad");
    REQUIRE(rows >= 0 && linear_wgrad_supports(out_dim, in_dim) && ldy >= out_dim && ldx >= in_dim, "eamrl_linear_wgrad");
    REQUIRE(ldy % 4 == 0 && ldx % 4 == 0 && (uintptr_t)dy % 16 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)dW % 16 == 0 &&
                (uintptr_t)scratch % 16 == 0 && (!db || (uintptr_t)db % 16 == 0), "eamrl_linear_wgrad");
    REQUIRE(scratch_floats >= linear_wgrad_scratch(rows, out_dim, in_dim), "eamrl_linear_wgrad");
    return launched(launch_linear_wgrad(dy, ldy, x, ldx, rows, out_dim, in_dim, dW, db, scratch, (hipStream_t)stream),
                    "eamrl_linear_wgrad");
}

__attribute__((visibility("default"))) int eamrl_mha_encoder(const float* qkv, float* out, int64_t B, int N, int E, int H,
                                                            void* stream)
{
    REQUIRE(qkv && out, "eamrl_mha_encoder");
    REQUIRE(B >= 0 && N > 0 && E > 0 && H > 0 && E % H == 0, "eamrl_mha_encoder");
    if (B == 0) return 0;
    return launched(launch_mha_encoder(qkv, out, B, N, E, H, (hipStream_t)stream), "eamrl_mha_encoder");
}

__attribute__((visibility("default"))) int eamrl_normalize(float* x, int64_t B, int N, int E, int kind,
                                                          const float* gamma, const float* beta, const float* mean,
                                                          const float* var, float eps, void* stream)
{
    REQUIRE(x && gamma && beta, "eamrl_normalize");
    REQUIRE(B >= 0 && N > 0 && E > 0 && E <= 8192, "eamrl_normalize");
    return launched(launch_normalize(x, B, N, E, kind, gamma, beta, mean, var, eps, (hipStream_t)stream),
                    "eamrl_normalize");
}

__attribute__((visibility("default"))) int eamrl_batchnorm_train(float* x, int64_t rows, int E, const float* gamma,
                                                                const float* beta, float* running_mean, float* running_var,
                                                                float momentum, float eps, float* save_mean, float* save_var,
                                                                float* ws, int64_t ws_floats, void* stream)
{
    REQUIRE(x && gamma && beta && save_mean && save_var && ws, "eamrl_batchnorm_train");
    REQUIRE(rows >= 0 && E > 0 && E <= 8192, "eamrl_batchnorm_train");
    REQUIRE((running_mean == nullptr) == (running_var == nullptr), "eamrl_batchnorm_train");
    REQUIRE(ws_floats >= ((rows + 127) / 128) * (int64_t)E, "eamrl_batchnorm_train");
    return launched(launch_batchnorm_train(x, rows, E, gamma, beta, running_mean, running_var, momentum, eps, save_mean,
                                           save_var, ws, (hipStream_t)stream), "eamrl_batchnorm_train");
}

__attribute__((visibility("default"))) int eamrl_pointer_attention(const float* query, const float* key, const float* value,
                                                                  const float* logit_key, int64_t ld, const uint8_t* mask,
                                                                  int mask_per_query, const float* Wout, const float* bout,
                                                                  float* logits, int64_t B, int L, int M, int E, int H,
                                                                  int mask_inner, void* stream)
{
    REQUIRE(query && key && value && logit_key && Wout && logits, "eamrl_pointer_attention");
    REQUIRE(B >= 0 && L > 0 && M > 0 && E > 0 && H > 0 && E % H == 0 && E / H <= 32 && ld >= E, "eamrl_pointer_attention");
    REQUIRE((3 * (int64_t)E + (int64_t)H * M) * 4 <= 64 * 1024, "eamrl_pointer_attention");
    return launched(launch_pointer_attention(query, key, value, logit_key, ld, mask, mask_per_query, Wout, bout, logits, B, L, M,
                                             E, H, mask_inner, (hipStream_t)stream), "eamrl_pointer_attention");
}

__attribute__((visibility("default"))) int eamrl_pack_linear_weight(const float* W, float* Wp, int out_dim, int in_dim,
                                                                   void* stream)
{
    REQUIRE(W && Wp && out_dim > 0 && in_dim > 0 && out_dim % 16 == 0 && in_dim % 16 == 0, "eamrl_pack_linear_weight");
    return launched(launch_pack_mfma_b(W, Wp, out_dim, in_dim, (hipStream_t)stream), "eamrl_pack_linear_weight");
}

__attribute__((visibility("default"))) int eamrl_encoder_fused_supported(int M, int E, int H, int ff_hidden, int nlayers)
{
    return encoder_fused_supports(M, E, H, ff_hidden, nlayers) ? 1 : 0;
}

__attribute__((visibility("default"))) int64_t eamrl_small_linear_wgrad_scratch(int64_t rows, int out_dim)
{
    if (rows < 0 || out_dim <= 0) return -1;
    return small_linear_wgrad_scratch(rows, out_dim);
}

__attribute__((visibility("default"))) int eamrl_small_linear_wgrad(const float* dy, int64_t ldy, const float* x, int64_t ldx,
                                                                   int64_t rows, int out_dim, int in_dim, float* dW, float* db,
                                                                   float* scratch, int64_t scratch_floats, void* stream)
{
    REQUIRE(dy && x && dW && scratch, "eamrl_small_linear_wgrad");
    REQUIRE(rows >= 0 && out_dim > 0 && in_dim >= 1 && in_dim <= 8 && ldy >= out_dim && ldx >= in_dim, "eamrl_small_linear_wgrad");
    REQUIRE(scratch_floats >= small_linear_wgrad_scratch(rows, out_dim), "eamrl_small_linear_wgrad");
    return launched(launch_small_linear_wgrad(dy, ldy, x, ldx, rows, out_dim, in_dim, dW, db, scratch, (hipStream_t)stream),
                    "eamrl_small_linear_wgrad");
}

__attribute__((visibility("default"))) int64_t eamrl_batchnorm_backward_scratch(int64_t rows, int E)
{
    if (rows < 0 || E <= 0) return -1;
    return batchnorm_backward_scratch(rows, E);
}

__attribute__((visibility("default"))) int eamrl_batchnorm_backward(const float* x, const float* dy, const float* save_mean,
                                                                   const float* save_var, const float* gamma, float eps, int64_t rows,
                                                                   int E, float* dx, float* dgamma, float* dbeta, float* scratch,
                                                                   int64_t scratch_floats, void* stream)
{
    REQUIRE(x && dy && save_mean && save_var && dx && scratch, "eamrl_batchnorm_backward");
    REQUIRE(rows >= 0 && E > 0 && E % 4 == 0, "eamrl_batchnorm_backward");
    REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)dy % 16 == 0) && ((uintptr_t)dx % 16 == 0), "eamrl_batchnorm_backward");
    REQUIRE(scratch_floats >= batchnorm_backward_scratch(rows, E), "eamrl_batchnorm_backward");
    return launched(launch_batchnorm_backward(x, dy, save_mean, save_var, gamma, eps, rows, E, dx, dgamma, dbeta, scratch,
                                              (hipStream_t)stream), "eamrl_batchnorm_backward");
}

__attribute__((visibility("default"))) int eamrl_augment_xy(const float* xy, const float* cs, const int32_t* code, float* out, int64_t R,
                                                           int64_t B, int N, float offset, void* stream)
{
    REQUIRE(xy && code && out && R >= 0 && B > 0 && N > 0, "eamrl_augment_xy");
    REQUIRE(((uintptr_t)xy % 8 == 0) && ((uintptr_t)out % 8 == 0), "eamrl_augment_xy");
    return launched(launch_augment_xy(xy, cs, code, out, R, B, N, offset, (hipStream_t)stream), "eamrl_augment_xy");
}

static int check_encoder_fused(int64_t B, int M, int E, int H, int ff_hidden, int nlayers, int norm,
                               const eamrl_encoder_layer* layers, const eamrl_encoder_cache* cache, const char* what)
{
    if (cache) {
        REQUIRE(cache->Wc && cache->WoutT && cache->out && cache->nproj >= 3 && cache->nproj <= 5, what);
        REQUIRE(cache->ld >= (int64_t)(cache->nproj + 1) * E && cache->ld % 4 == 0 && ((uintptr_t)cache->out % 16 == 0) &&
                    ((uintptr_t)cache->Wc % 16 == 0) && ((uintptr_t)cache->WoutT % 16 == 0), what);
    }
    REQUIRE(layers && B >= 0, what);
    REQUIRE(encoder_fused_supports(M, E, H, ff_hidden, nlayers), what);
    REQUIRE(norm == EAMRL_NORM_BATCH_EVAL || norm == EAMRL_NORM_INSTANCE, what);
    for (int l = 0; l < nlayers; ++l) {
        const eamrl_encoder_layer& s = layers[l];
        REQUIRE(s.Wqkv && s.bqkv && s.Wo && s.bo && s.W1 && s.b1 && s.W2 && s.b2 && s.n1_gamma && s.n1_beta && s.n2_gamma &&
                    s.n2_beta, what);
        REQUIRE(((uintptr_t)s.Wqkv % 16 == 0) && ((uintptr_t)s.Wo % 16 == 0) && ((uintptr_t)s.W1 % 16 == 0) &&
                    ((uintptr_t)s.W2 % 16 == 0), what);
        if (norm == EAMRL_NORM_BATCH_EVAL) REQUIRE(s.n1_mean && s.n1_var && s.n2_mean && s.n2_var, what);
    }
    return 0;
}

__attribute__((visibility("default"))) int eamrl_encoder_fused(const float* h_in, float* h_out, int64_t B, int M, int E, int H,
                                                              int ff_hidden, int nlayers, int norm, float eps,
                                                              const eamrl_encoder_layer* layers,
                                                              const eamrl_encoder_cache* cache, void* stream)
{
    if (int rc = check_encoder_fused(B, M, E, H, ff_hidden, nlayers, norm, layers, cache, "eamrl_encoder_fused")) return rc;
    REQUIRE(h_in && h_out, "eamrl_encoder_fused");
    REQUIRE(((uintptr_t)h_in % 16 == 0) && ((uintptr_t)h_out % 16 == 0), "eamrl_encoder_fused");
    return launched(launch_encoder_fused(h_in, h_out, B, M, nlayers, norm, eps, layers, cache, nullptr, (hipStream_t)stream),
                    "eamrl_encoder_fused");
}

__attribute__((visibility("default"))) int eamrl_encoder_fused_init(const eamrl_encoder_init* init, float* h_out, int64_t B, int M,
                                                                   int E, int H, int ff_hidden, int nlayers, int norm, float eps,
                                                                   const eamrl_encoder_layer* layers,
                                                                   const eamrl_encoder_cache* cache, void* stream)
{
    const char* what = "eamrl_encoder_fused_init";
    if (int rc = check_encoder_fused(B, M, E, H, ff_hidden, nlayers, norm, layers, cache, what)) return rc;
    REQUIRE(init && init->feat && init->W && init->F >= 1 && init->F <= 8, what);
    REQUIRE(!init->depot || (init->Wd && init->depot_ld >= 2), what);
    REQUIRE(h_out || cache, what);          // something must leave the kernel
    REQUIRE((!h_out || (uintptr_t)h_out % 16 == 0) && (!init->init_out || (uintptr_t)init->init_out % 16 == 0), what);
    return launched(launch_encoder_fused(nullptr, h_out, B, M, nlayers, norm, eps, layers, cache, init, (hipStream_t)stream), what);
}

static int check_reeval(const eamrl_reeval* p, const char* what, bool bwd)
{
    REQUIRE(p, what);
    REQUIRE(reeval_supports(p->M, 128, 8), what);
    if (p->M > 112)         // key chunks: scratch from the caller, the forward pass always run, no rollout heads / dynamic embedding
        REQUIRE(p->scratch && p->lse && !p->heads && (uintptr_t)p->scratch % 16 == 0 && p->NC <= 2, what);
    REQUIRE(p->K && p->V && p->Lp && p->Pa && p->idxA && p->maskbits && p->actions && p->logp, what);
    REQUIRE((p->idxB == nullptr) == (p->Pb == nullptr) || p->Pb, what);
    REQUIRE(p->B > 0 && p->S > 0 && p->T > 0 && p->R == p->B * p->S && p->nchunk >= 1 && p->nchunk <= p->S, what);
    REQUIRE(p->ld >= 128 && p->ld % 4 == 0 && p->NC >= 0 && p->NC <= 4 && (p->NC == 0 || (p->Cvec && p->sc)), what);
    REQUIRE(p->temp > 0.0f && p->tstart >= 0, what);
    REQUIRE(!p->heads || (p->heads_T > 0 && (uintptr_t)p->heads % 16 == 0), what);
    REQUIRE((p->dyn == nullptr) == (p->rem == nullptr), what);
    REQUIRE(!p->dyn || (!p->heads && (uintptr_t)p->rem % 16 == 0 && (!bwd || p->ddyn)), what);
    REQUIRE(((uintptr_t)p->Pa % 16 == 0) && (!p->Pb || (uintptr_t)p->Pb % 16 == 0) && (!p->gctx || (uintptr_t)p->gctx % 16 == 0) &&
                (!p->Cvec || (uintptr_t)p->Cvec % 16 == 0) && ((uintptr_t)p->maskbits % 16 == 0), what);
    if (bwd) {
        REQUIRE(p->glogp && p->dheads && p->dK && p->dV && p->dLp && p->dPa && p->ldg >= 128 && p->ldg % 4 == 0, what);
        REQUIRE((p->Pb == nullptr) == (p->dPb == nullptr) && (p->gctx == nullptr) == (p->dgctx == nullptr) &&
                    (p->NC == 0 || p->dCvec) && ((uintptr_t)p->dheads % 16 == 0), what);
        REQUIRE(p->NC <= 2 && p->R * (int64_t)p->T < (int64_t)1 << 31, what);     // the backward kernels' query indices

    }
    return 0;
}

__attribute__((visibility("default"))) int eamrl_reeval_supported(int M, int E, int H) { return reeval_supports(M, E, H) ? 1 : 0; }
__attribute__((visibility("default"))) int64_t eamrl_reeval_scratch_floats(int64_t R, int T, int M) { return reeval_scratch_floats(R, T, M); }

__attribute__((visibility("default"))) int eamrl_reeval_forward(const eamrl_reeval* p, void* stream)
{
    if (int rc = check_reeval(p, "eamrl_reeval_forward", false)) return rc;
    return launched(launch_reeval_fwd(*p, (hipStream_t)stream), "eamrl_reeval_forward");
}

__attribute__((visibility("default"))) int eamrl_reeval_backward(const eamrl_reeval* p, void* stream)
{
    if (int rc = check_reeval(p, "eamrl_reeval_backward", true)) return rc;
    return launched(launch_reeval_bwd(*p, (hipStream_t)stream), "eamrl_reeval_backward");
}

__attribute__((visibility("default"))) int eamrl_replay_states(int env, const eamrl_state* s, int64_t R, int64_t B, int M,
                                                              const int64_t* actions, int T, uint32_t* bits, int32_t* idxA,
                                                              float* sc, void* stream)
{
    REQUIRE(env == EAMRL_ENV_CVRP || env == EAMRL_ENV_CVRPTW || env == EAMRL_ENV_PCTSP || env == EAMRL_ENV_OP,
            "eamrl_replay_states (CVRP, CVRPTW, PCTSP or OP)");
    REQUIRE(s && actions && bits && idxA && sc && R >= 0 && B > 0 && R % B == 0 && M >= 2 && M <= 1024 && T > 0 &&
                ((uintptr_t)bits % 16 == 0), "eamrl_replay_states");
    REQUIRE(s->mask && s->visited && s->used && s->vcap && s->cur && s->demand, "eamrl_replay_states (state)");
    if (env == EAMRL_ENV_PCTSP || env == EAMRL_ENV_OP) REQUIRE(s->istep != nullptr, "eamrl_replay_states (istep)");
    if (env == EAMRL_ENV_OP) REQUIRE(s->locs != nullptr, "eamrl_replay_states (locs)");
    if (env == EAMRL_ENV_CVRPTW) REQUIRE(s->time && s->locs && s->tw && s->dur, "eamrl_replay_states (time windows)");
    if (R == 0) return 0;
    return launched(launch_replay_states(env, s->mask, s->visited, s->used, s->vcap, s->cur, s->istep, s->time, s->demand, s->locs,
                                         s->tw, s->dur, actions, bits, idxA, sc, R, B, M, T, (hipStream_t)stream),
                    "eamrl_replay_states");
}

__attribute__((visibility("default"))) int eamrl_replay_states_sdvrp(const eamrl_state* s, int64_t R, int M, const int64_t* actions,
                                                                    int T, uint32_t* bits, int32_t* idxA, float* sc,
                                                                    float* rem_out, void* stream)
{
    REQUIRE(s && actions && bits && idxA && sc && rem_out && R >= 0 && M >= 2 && M <= 1024 && T > 0 && ((uintptr_t)bits % 16 == 0),
            "eamrl_replay_states_sdvrp");
    REQUIRE(s->rem && s->used && s->vcap && s->cur, "eamrl_replay_states_sdvrp (state)");
    if (R == 0) return 0;
    return launched(launch_replay_sdvrp(s->rem, s->used, s->vcap, s->cur, actions, bits, idxA, sc, rem_out, R, M, T,
                                        (hipStream_t)stream), "eamrl_replay_states_sdvrp");
}

__attribute__((visibility("default"))) int eamrl_pack_mask_bits(const uint8_t* mask, uint32_t* bits, int64_t R, int M, int T,
                                                               int t, void* stream)
{
    REQUIRE(mask && bits && R >= 0 && M > 0 && M <= 128 && T > 0 && t >= 0 && t < T, "eamrl_pack_mask_bits");
    return launched(launch_pack_mask_bits(mask, bits, R, M, T, t, (hipStream_t)stream), "eamrl_pack_mask_bits");
}

__attribute__((visibility("default"))) int eamrl_pack_mask_bits_chunked(const uint8_t* mask, uint32_t* bits, int64_t R, int M, int T,
                                                                       int t, void* stream)
{
    REQUIRE(mask && bits && R >= 0 && M > 0 && M <= 1024 && T > 0 && t >= 0 && t < T && ((uintptr_t)bits % 16 == 0),
            "eamrl_pack_mask_bits_chunked");
    return launched(launch_pack_mask_bits_chunked(mask, bits, R, M, T, t, (hipStream_t)stream), "eamrl_pack_mask_bits_chunked");
}

__attribute__((visibility("default"))) int eamrl_tsp_mask_bits_chunked(const int64_t* actions, uint32_t* bits, int64_t R, int M, int T,
                                                                      void* stream)
{
    REQUIRE(actions && bits && R >= 0 && M > 0 && M <= 1024 && T > 0 && ((uintptr_t)bits % 16 == 0), "eamrl_tsp_mask_bits_chunked");
    return launched(launch_tsp_mask_bits_chunked(actions, bits, R, M, T, (hipStream_t)stream), "eamrl_tsp_mask_bits_chunked");
}

__attribute__((visibility("default"))) int eamrl_tsp_mask_bits(const int64_t* actions, uint32_t* bits, int64_t R, int M, int T,
                                                              void* stream)
{
    REQUIRE(actions && bits && R >= 0 && M > 0 && M <= 128 && T > 0 && ((uintptr_t)bits % 16 == 0), "eamrl_tsp_mask_bits");
    return launched(launch_tsp_mask_bits(actions, bits, R, M, T, (hipStream_t)stream), "eamrl_tsp_mask_bits");
}

__attribute__((visibility("default"))) int eamrl_instance_norm_forward(const float* x, float* y, float* mean, float* rstd, int64_t B,
                                                                      int N, int E, const float* gamma, const float* beta,
                                                                      float eps, void* stream)
{
    REQUIRE(x && y && mean && rstd && B >= 0 && N > 0 && E > 0, "eamrl_instance_norm_forward");
    if (B == 0) return 0;
    return launched(launch_instnorm_train_fwd(x, y, mean, rstd, B, N, E, gamma, beta, eps, (hipStream_t)stream),
                    "eamrl_instance_norm_forward");
}

__attribute__((visibility("default"))) int eamrl_instance_norm_backward(const float* x, const float* dy, const float* mean,
                                                                       const float* rstd, const float* gamma, float* dx,
                                                                       float* dgamma, float* dbeta, int64_t B, int N, int E,
                                                                       void* stream)
{
    REQUIRE(x && dy && mean && rstd && dx && B >= 0 && N > 0 && E > 0, "eamrl_instance_norm_backward");
    if (B == 0) return 0;
    return launched(launch_instnorm_train_bwd(x, dy, mean, rstd, gamma, dx, dgamma, dbeta, B, N, E, (hipStream_t)stream),
                    "eamrl_instance_norm_backward");
}

__attribute__((visibility("default"))) int eamrl_mean_nodes(const float* emb, float* out, int64_t B, int M, int E,
                                                           void* stream)
{
    REQUIRE(emb && out && B >= 0 && M > 0 && E > 0, "eamrl_mean_nodes");
    return launched(launch_mean_nodes(emb, out, B, M, E, (hipStream_t)stream), "eamrl_mean_nodes");
}

static int fill_args(const char* what, int env, const eamrl_cache* c, const eamrl_state* s, int64_t R, int mode,
                     const float* noise, const int64_t* given, float clip, float temp, int top_k, float top_p,
                     uint32_t* status, DecArgs& a, bool seeded = false)
{
    REQUIRE(c && s, what);
    REQUIRE(env >= EAMRL_ENV_TSP && env <= EAMRL_ENV_CVRPTW, what);
    REQUIRE(mode == EAMRL_GREEDY || mode == EAMRL_SAMPLE || mode == EAMRL_EVALUATE, what);
    REQUIRE(c->K && c->V && c->Lp && c->Pa && c->cvec, what);
    REQUIRE(c->B > 0 && c->M > 0 && c->E > 0 && c->H > 0, what);
    REQUIRE(c->E % c->H == 0 && (c->E / c->H) % 4 == 0 && c->E % (4 * EAMRL_NCHUNK) == 0, what);
    REQUIRE(c->ld >= c->E && c->ld % 4 == 0, what);
    REQUIRE(((uintptr_t)c->K % 16 == 0) && ((uintptr_t)c->V % 16 == 0) && ((uintptr_t)c->Lp % 16 == 0), what);
    REQUIRE(R > 0 && R % c->B == 0 && R <= 0x7fffffffLL, what);
    REQUIRE(s->cur && s->mask && status, what);
    REQUIRE(temp > 0.0f && top_k >= 0 && top_p >= 0.0f && top_p <= 1.0f, what);
    if (env == EAMRL_ENV_TSP) REQUIRE(c->Pb && s->first && s->istep, what);
    if (env == EAMRL_ENV_CVRP) REQUIRE(s->used && s->vcap && c->M >= 2, what);
    if (env == EAMRL_ENV_SDVRP) REQUIRE(s->used && s->vcap && s->rem && c->dyn && c->M >= 2, what);
    if (env == EAMRL_ENV_PCTSP) REQUIRE(s->used && s->vcap && c->M >= 2, what);
    if (env == EAMRL_ENV_OP) REQUIRE(s->used && s->vcap && c->M >= 2, what);
    if (env == EAMRL_ENV_CVRPTW) REQUIRE(s->used && s->vcap && s->time && c->M >= 2, what);
    if (mode == EAMRL_SAMPLE) REQUIRE(noise != nullptr || seeded, what);
    if (mode == EAMRL_EVALUATE) REQUIRE(given != nullptr, what);
    a = DecArgs{};
    a.K = c->K; a.V = c->V; a.Lp = c->Lp; a.Pa = c->Pa; a.Pb = c->Pb; a.cvec = c->cvec; a.gctx = c->gctx;
    a.ld = c->ld; a.B = c->B; a.M = c->M; a.E = c->E; a.H = c->H;
    a.first = s->first; a.cur = s->cur; a.istep = s->istep; a.used = s->used; a.vcap = s->vcap; a.demand = s->demand;
    a.mask = s->mask; a.visited = s->visited; a.done = s->done;
    a.rem = s->rem; a.dyn = c->dyn; a.locs = s->locs; a.time = s->time; a.tw = s->tw; a.dur = s->dur;
    a.heads_out = s->heads_out;
    a.seed = 0; a.seed_dev = nullptr; a.use_rng = 0;
    a.R = R; a.mode = mode; a.noise = noise; a.given = given; a.clip = clip; a.temp = temp; a.top_k = top_k; a.top_p = top_p; a.status = status;
    return 0;
}

__attribute__((visibility("default"))) int eamrl_am_decode_step(int env, const eamrl_cache* cache_host,
                                                               const eamrl_state* state_host, int64_t R, int mode,
                                                               const float* noise, const int64_t* given, float tanh_clip,
                                                               float temperature, int top_k, float top_p,
                                                               int fuse_env_step, int64_t* action,
                                                               float* logp, float* logprobs_all, float* logits_raw,
                                                               uint32_t* status, void* stream)
{
    DecArgs a;
    int rc = fill_args("eamrl_am_decode_step", env, cache_host, state_host, R, mode, noise, given, tanh_clip, temperature,
                       top_k, top_p, status, a);
    if (rc) return rc;
    REQUIRE(action && logp, "eamrl_am_decode_step");
    if (fuse_env_step) {
        REQUIRE(a.done, "eamrl_am_decode_step");
        if (env == EAMRL_ENV_CVRP) REQUIRE(a.visited && a.demand, "eamrl_am_decode_step");
        if (env == EAMRL_ENV_PCTSP) REQUIRE(a.visited && a.demand && a.istep, "eamrl_am_decode_step");
        if (env == EAMRL_ENV_OP) REQUIRE(a.visited && a.demand && a.istep && a.locs, "eamrl_am_decode_step");
        if (env == EAMRL_ENV_CVRPTW) REQUIRE(a.visited && a.demand && a.locs && a.tw && a.dur, "eamrl_am_decode_step");
    }
    a.fuse_env = fuse_env_step;
    a.action = action; a.logp = logp; a.logprobs_all = logprobs_all; a.logits_raw = logits_raw;
    return launched(launch_decode_step(env, a, (hipStream_t)stream), "eamrl_am_decode_step");
}

__attribute__((visibility("default"))) int eamrl_am_rollout(int env, const eamrl_cache* cache_host,
                                                           const eamrl_state* state_host, int64_t R, int mode,
                                                           const float* noise, const int64_t* given, int t_given,
                                                           float tanh_clip, float temperature, int top_k, float top_p,
                                                           int t_max, int64_t* actions, float* logps, int32_t* steps_out,
                                                           uint32_t* status, void* stream)
{
    DecArgs a;
    int rc = fill_args("eamrl_am_rollout", env, cache_host, state_host, R, mode, noise, given, tanh_clip, temperature,
                       top_k, top_p, status, a);
    if (rc) return rc;
    REQUIRE(actions && logps && steps_out && a.done && t_max > 0, "eamrl_am_rollout");
    if (env == EAMRL_ENV_CVRP) REQUIRE(a.visited && a.demand, "eamrl_am_rollout");
    if (env == EAMRL_ENV_PCTSP) REQUIRE(a.visited && a.demand && a.istep, "eamrl_am_rollout");
    if (env == EAMRL_ENV_OP) REQUIRE(a.visited && a.demand && a.istep && a.locs, "eamrl_am_rollout");
    if (env == EAMRL_ENV_CVRPTW) REQUIRE(a.visited && a.demand && a.locs && a.tw && a.dur, "eamrl_am_rollout");
    if (mode == EAMRL_EVALUATE) REQUIRE(t_given > 0, "eamrl_am_rollout");
    a.fuse_env = 1; a.t_max = t_max; a.t_given = t_given;
    a.action = actions; a.logp = logps; a.steps_out = steps_out;
    const bool filtering = top_k > 0 || (top_p > 0.0f && top_p < 1.0f);        // only the streaming kernel filters
    if (!g_debug[11] && !filtering && rollout_ms_mfma_supports(env, a))
        return launched(launch_rollout_ms_mfma(env, a, (hipStream_t)stream), "eamrl_am_rollout");
    if (!g_debug[1] && !filtering && rollout_resident_supports(env, a))
        return launched(launch_rollout_resident(env, a, (hipStream_t)stream), "eamrl_am_rollout");
    return launched(launch_rollout_stream(env, a, (hipStream_t)stream), "eamrl_am_rollout");
}

__attribute__((visibility("default"))) int eamrl_exp1_noise(uint64_t seed, const uint64_t* seed_dev, float* noise, int64_t R,
                                                           int T, int M, void* stream)
{
    REQUIRE(noise && R >= 0 && T > 0 && M > 0, "eamrl_exp1_noise");
    return launched(launch_exp1_noise(seed, seed_dev, noise, R, T, M, (hipStream_t)stream), "eamrl_exp1_noise");
}

__attribute__((visibility("default"))) int eamrl_rollout_rng_native(int env, const eamrl_cache* cache_host, int64_t R)
{
    if (!cache_host || R <= 0 || g_debug[11]) return 0;
    DecArgs a{};
    a.B = cache_host->B; a.M = cache_host->M; a.E = cache_host->E; a.H = cache_host->H; a.ld = cache_host->ld; a.R = R;
    return rollout_ms_mfma_supports(env, a, true) ? 1 : 0;
}

__attribute__((visibility("default"))) int eamrl_am_rollout_seeded(int env, const eamrl_cache* cache_host,
                                                                  const eamrl_state* state_host, int64_t R, uint64_t seed,
                                                                  const uint64_t* seed_dev, float* noise_scratch,
                                                                  float tanh_clip, float temperature,
                                                                  int t_max, int64_t* actions, float* logps,
                                                                  int32_t* steps_out, uint32_t* status, void* stream)
{
    DecArgs a;
    int rc = fill_args("eamrl_am_rollout_seeded", env, cache_host, state_host, R, EAMRL_SAMPLE, nullptr, nullptr, tanh_clip,
                       temperature, 0, 0.0f, status, a, true);
    if (rc) return rc;
    REQUIRE(actions && logps && steps_out && a.done && t_max > 0, "eamrl_am_rollout_seeded");
    a.fuse_env = 1; a.t_max = t_max; a.t_given = 0;
    a.action = actions; a.logp = logps; a.steps_out = steps_out;
    if (!g_debug[11] && rollout_ms_mfma_supports(env, a)) {
        a.seed = seed; a.seed_dev = seed_dev; a.use_rng = 1;
        return launched(launch_rollout_ms_mfma(env, a, (hipStream_t)stream), "eamrl_am_rollout_seeded");
    }
    // kernels without in-place noise: the same draws as a tensor in the caller's scratch
    REQUIRE(noise_scratch != nullptr, "eamrl_am_rollout_seeded (noise_scratch [R][t_max][M] needed for this shape)");
    rc = launch_exp1_noise(seed, seed_dev, noise_scratch, R, t_max, a.M, (hipStream_t)stream);
    if (rc) return launched(rc, "eamrl_am_rollout_seeded");
    return eamrl_am_rollout(env, cache_host, state_host, R, EAMRL_SAMPLE, noise_scratch, nullptr, 0, tanh_clip, temperature, 0,
                            0.0f, t_max, actions, logps, steps_out, status, stream);
}

__attribute__((visibility("default"))) int eamrl_tour_length(const float* locs, const int64_t* actions, float* reward,
                                                            int64_t R, int64_t B, int M, int T, int with_depot,
                                                            void* stream)
{
    REQUIRE(locs && actions && reward && R >= 0 && B > 0 && M > 0 && T > 0, "eamrl_tour_length");
    if (R == 0) return 0;
    return launched(launch_tour_length(locs, actions, reward, R, B, M, T, with_depot, (hipStream_t)stream),
                    "eamrl_tour_length");
}

__attribute__((visibility("default"))) int eamrl_sum_logp(const float* logp, int64_t ld, float* out, int64_t R, int T,
                                                         void* stream)
{
    REQUIRE(logp && out && R >= 0 && T >= 0 && ld >= T, "eamrl_sum_logp");
    if (R == 0) return 0;
    return launched(launch_sum_logp(logp, ld, out, R, T, (hipStream_t)stream), "eamrl_sum_logp");
}

__attribute__((visibility("default"))) int eamrl_rollout_finish(int env, const float* locs, const int64_t* actions,
                                                               const float* logp, int64_t ld, const float* demand,
                                                               const float* vcap, float* reward, float* ll, int32_t* bad,
                                                               int64_t R, int64_t B, int M, int T, void* stream)
{
    REQUIRE(env == EAMRL_ENV_TSP || env == EAMRL_ENV_CVRP, "eamrl_rollout_finish (TSP or CVRP)");
    REQUIRE(locs && actions && R >= 0 && B > 0 && M >= 2 && M <= 4096 && T > 0, "eamrl_rollout_finish");
    REQUIRE(!ll || (logp && ld >= T), "eamrl_rollout_finish (logp)");
    if (env == EAMRL_ENV_CVRP && bad) REQUIRE(demand && vcap, "eamrl_rollout_finish (demand, vcap)");
    if (R == 0) return 0;
    return launched(launch_rollout_finish(env, locs, actions, logp, ld, demand, vcap, reward, ll, bad, R, B, M, T,
                                          (hipStream_t)stream), "eamrl_rollout_finish");
}

__attribute__((visibility("default"))) int eamrl_multi_copy(int n, const void* const* src, void* const* dst,
                                                           const int64_t* bytes, void* stream)
{
    REQUIRE(n >= 0 && n <= EAMRL_MULTI_COPY_MAX && (n == 0 || (src && dst && bytes)), "eamrl_multi_copy");
    for (int i = 0; i < n; ++i) REQUIRE(bytes[i] >= 0 && (bytes[i] == 0 || dst[i]), "eamrl_multi_copy (segment)");
    return launched(launch_multi_copy(n, src, dst, bytes, (hipStream_t)stream), "eamrl_multi_copy");
}

__attribute__((visibility("default"))) int eamrl_check_solution(int env, const int64_t* actions, const float* demand,
                                                               const float* vcap, int64_t R, int64_t B, int N, int T,
                                                               int32_t* bad, void* stream)
{
    REQUIRE(actions && bad && R >= 0 && B > 0 && N > 0 && T > 0, "eamrl_check_solution");
    REQUIRE(env >= EAMRL_ENV_TSP && env <= EAMRL_ENV_PCTSP, "eamrl_check_solution");
    if (env == EAMRL_ENV_CVRP || env == EAMRL_ENV_SDVRP) REQUIRE(demand && vcap, "eamrl_check_solution");
    if (env == EAMRL_ENV_PCTSP) REQUIRE(demand != nullptr, "eamrl_check_solution");
    if (R == 0) return 0;
    return launched(launch_check_solution(env, actions, demand, vcap, R, B, N, T, bad, (hipStream_t)stream),
                    "eamrl_check_solution");
}

__attribute__((visibility("default"))) int eamrl_beam_topk(const float* logprobs, const float* parent, int64_t B, int beam_width,
                                                          int M, int64_t* node, int32_t* beam, float* cum, float* step_logp,
                                                          void* stream)
{
    REQUIRE(logprobs && parent && node && beam && cum && step_logp, "eamrl_beam_topk");
    REQUIRE(B >= 0 && B <= 0x7fffffffLL && beam_width >= 1 && M >= 1 && (int64_t)beam_width * M <= 36000,
            "eamrl_beam_topk (beam_width * M <= 36000)");
    if (B == 0) return 0;
    return launched(launch_beam_topk(logprobs, parent, B, beam_width, M, node, beam, cum, step_logp, (hipStream_t)stream),
                    "eamrl_beam_topk");
}

__attribute__((visibility("default"))) int eamrl_ea_tsp_run(const float* locs, int64_t* pop, float* fitness, int64_t B,
                                                           int S, int N, int num_generations, double mutation_rate,
                                                           double crossover_rate, double selection_rate,
                                                           const double* cross_rand, const int32_t* cross_idx,
                                                           const double* mut_rand, const int32_t* mut_idx, void* stream)
{
    REQUIRE(locs && pop && fitness, "eamrl_ea_tsp_run");
    REQUIRE(B >= 0 && B <= 0x7fffffffLL && S >= 1 && S <= 128 && N >= 2 && N <= 128 && num_generations >= 0,
            "eamrl_ea_tsp_run (population and tour length are limited to 128)");
    REQUIRE(mutation_rate == mutation_rate && crossover_rate == crossover_rate && selection_rate >= 0.0,
            "eamrl_ea_tsp_run");
    int ne = S;
    if (S > 2) { ne = (int)(selection_rate * (double)S); if (ne <= 0 || ne > S) ne = S; }
    if (num_generations > 0 && ne / 2 > 0)      // an elite set of one produces no offspring and reads no draws
        REQUIRE(cross_rand && cross_idx && mut_rand && mut_idx, "eamrl_ea_tsp_run (draws)");
    if (B == 0) return 0;
    return launched(launch_ea_tsp(locs, pop, fitness, B, S, N, num_generations, mutation_rate, crossover_rate,
                                  selection_rate, cross_rand, cross_idx, mut_rand, mut_idx, (hipStream_t)stream),
                    "eamrl_ea_tsp_run");
}

__attribute__((visibility("default"))) int eamrl_ea_cvrp_run(const float* locs, const float* demand, const float* vcap,
                                                            int64_t* pop, float* fitness, int64_t B, int S, int N, int L,
                                                            int num_generations, double mutation_rate,
                                                            double crossover_rate, double selection_rate, int top_k,
                                                            const double* init_mut_rand, const double* init_mut_u,
                                                            const double* cross_rand, const double* cross_u,
                                                            const double* mut_rand, const double* mut_u, void* stream)
{
    REQUIRE(locs && demand && vcap && pop && fitness && init_mut_rand && init_mut_u, "eamrl_ea_cvrp_run");
    REQUIRE(B >= 0 && B <= 0x7fffffffLL && S >= 1 && S <= 128 && N >= 1 && N <= 127 && L >= 2 && L <= 256 &&
            num_generations >= 0, "eamrl_ea_cvrp_run (S <= 128, N <= 127 customers, L <= 256)");
    REQUIRE((int64_t)S * L <= 24000, "eamrl_ea_cvrp_run (S * L <= 24000: three int16 populations must fit LDS)");
    REQUIRE(mutation_rate == mutation_rate && crossover_rate == crossover_rate && selection_rate >= 0.0,
            "eamrl_ea_cvrp_run");
    int ne = S;
    if (S > 2) { ne = (int)(selection_rate * (double)S); if (ne <= 0 || ne > S) ne = S; }
    if (num_generations > 0 && ne / 2 > 0) REQUIRE(cross_rand && cross_u && mut_rand && mut_u, "eamrl_ea_cvrp_run (draws)");
    if (B == 0) return 0;
    return launched(launch_ea_cvrp(locs, demand, vcap, pop, fitness, B, S, N, L, num_generations, mutation_rate,
                                   crossover_rate, selection_rate, top_k, init_mut_rand, init_mut_u, cross_rand, cross_u,
                                   mut_rand, mut_u, (hipStream_t)stream), "eamrl_ea_cvrp_run");
}

__attribute__((visibility("default"))) int eamrl_ea_prize_run(int env, const float* locs, const float* prize, const float* aux,
                                                             int64_t* pop, float* fitness, int64_t B, int S, int N, int L,
                                                             int num_generations, double mutation_rate,
                                                             double crossover_rate, double selection_rate, int top_k,
                                                             const double* init_mut_rand, const double* init_mut_u,
                                                             const double* cross_rand, const double* cross_u,
                                                             const double* mut_rand, const double* mut_u, void* stream)
{
    REQUIRE(env == EAMRL_ENV_PCTSP || env == EAMRL_ENV_OP, "eamrl_ea_prize_run (env: PCTSP or OP)");
    REQUIRE(locs && prize && aux && pop && fitness && init_mut_rand && init_mut_u, "eamrl_ea_prize_run");
    REQUIRE(B >= 0 && B <= 0x7fffffffLL && S >= 1 && S <= 128 && N >= 1 && N <= 127 && L >= 2 && L <= 128 &&
            num_generations >= 0, "eamrl_ea_prize_run (S <= 128, N <= 127 customers, L <= 128)");
    REQUIRE(mutation_rate == mutation_rate && crossover_rate == crossover_rate && selection_rate >= 0.0,
            "eamrl_ea_prize_run");
    int ne = S;
    if (S > 2) { ne = (int)(selection_rate * (double)S); if (ne <= 0 || ne > S) ne = S; }
    if (num_generations > 0 && ne / 2 > 0)
        REQUIRE(cross_rand && mut_rand && mut_u && (cross_u || env == EAMRL_ENV_PCTSP), "eamrl_ea_prize_run (draws)");
    if (B == 0) return 0;
    return launched(launch_ea_prize(env, locs, prize, aux, pop, fitness, B, S, N, L, num_generations, mutation_rate,
                                    crossover_rate, selection_rate, top_k, init_mut_rand, init_mut_u, cross_rand, cross_u,
                                    mut_rand, mut_u, (hipStream_t)stream), "eamrl_ea_prize_run");
}

}  // extern "C"
