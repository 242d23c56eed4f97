// Shared declarations of the decode kernels (step API + whole-rollout).
#pragma once
#include "../../include/eamrl.h"
#include "dmath.hpp"

namespace eamrl {

// Flat kernel-argument copy of (eamrl_cache, eamrl_state) plus per-call parameters.
struct DecArgs {
    // cache
    const float* K; const float* V; const float* Lp; const float* Pa; const float* Pb;
    const float* cvec; const float* gctx;
    int64_t ld; int64_t B; int M, E, H;
    // state
    int64_t* first; int64_t* cur; int64_t* istep; float* used; const float* vcap; const float* demand;
    uint8_t* mask; uint8_t* visited; uint8_t* done;
    float* rem; const float* dyn;      // SDVRP: remaining demand [R][M], dynamic-embedding vectors [3][E]
    const float* locs;                 // OP, CVRPTW: node coordinates [B][M][2]
    float* time; const float* tw; const float* dur;   // CVRPTW: clock [R], windows [B][M][2], service times [B][M]
    // call
    int64_t R; int mode; const float* noise; const int64_t* given; int t_given;
    uint64_t seed; const uint64_t* seed_dev; int use_rng;   // SAMPLE without a noise tensor: exp1_noise4(seed ^ *seed_dev, row,
                                                            // step, node / 4) (dmath.hpp); seed_dev may be null
    float clip, temp; int fuse_env; int t_max;
    int top_k; float top_p;      // process_logits filtering (0 = off); handled by the step / streaming kernels
    int64_t* action; float* logp; float* logprobs_all; float* logits_raw;
    int32_t* steps_out; uint32_t* status;
    float* heads_out;   // optional [R][t_max][E]: the glimpse output of every decode step (start-sharing MFMA kernel only)
};

// LDS carve for one row handled by one workgroup.
struct RowLds {
    static constexpr int PAD = 4;   // floats between the per-head / per-chunk segments of q / heads (bank spreading)
    float* q;       // [H][D + PAD]
    float* heads;   // [NCHUNK][E / NCHUNK + PAD]
    float* w;       // [H][M] scores -> softmax weights
    float* x;       // [M]    processed logits -> log-probs
    float* partA;   // [NCHUNK][E]
    float* partZ;   // [NCHUNK][H]
    float* partL;   // [M][NCHUNK] logit partials, then exp terms [M]
    float* red;     // [64] reduction scratch
    int* redi;      // [64]
    uint8_t* msk;   // [M] (padded to 16 B)
};

inline size_t row_lds_bytes(int M, int E, int H)
{
    size_t f = 2 * (size_t)E + (size_t)RowLds::PAD * (H + EAMRL_NCHUNK) + (size_t)H * M + M + (size_t)EAMRL_NCHUNK * E + (size_t)EAMRL_NCHUNK * H +
               (size_t)M * EAMRL_NCHUNK + 64 + 64;
    return f * 4 + (((size_t)M + 15) & ~(size_t)15);
}

__device__ __forceinline__ RowLds carve_row_lds(char* base, int M, int E, int H)
{
    RowLds l;
    float* f = reinterpret_cast<float*>(base);
    l.q = f; f += E + RowLds::PAD * H;
    l.heads = f; f += E + RowLds::PAD * EAMRL_NCHUNK;
    l.w = f; f += (size_t)H * M;
    l.x = f; f += M;
    l.partA = f; f += EAMRL_NCHUNK * E;
    l.partZ = f; f += EAMRL_NCHUNK * H;
    l.partL = f; f += (size_t)M * EAMRL_NCHUNK;
    l.red = f; f += 64;
    l.redi = reinterpret_cast<int*>(f); f += 64;
    l.msk = reinterpret_cast<uint8_t*>(f);
    return l;
}

}  // namespace eamrl
