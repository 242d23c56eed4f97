// Evolutionary improvement of TSP / CVRP tour populations (the fork's EA.run), one workgroup per problem instance.
//
// Reference (numba on CPU threads, one Python thread per instance):
//   EA.run ................... rl4co/models/zoo/earl/evolution.py:252-354
//   calculate_fitness_tsp .... :356-362     elitism_selection ..... :1103-1108
//   order_crossover_tsp ...... :392-488     inverse_mutate_tsp .... :490-517
//   calculate_fitness_cvrp ... :364-370     order_crossover_cvrp .. :585-788     inverse_mutate_cvrp ... :519-553
//
// The whole run of an instance -- G generations of select / crossover / mutate / fitness / replace -- happens in
// LDS: population and offspring as int16 rows, fitness as fp32, nothing but the final population goes back to
// HBM.  The reference draws random numbers inside the operators (numba's per-thread generators); here every draw
// is an input (see eamrl.h), which is what makes the operators reproducible and testable.
//
// Integer results are bit-exact against oracle/ea_oracle.py.  Fitness uses the canonical tour length
// (sqrtf(fmaf(dy,dy,dx*dx)) per leg, lane tree over legs), identical to eamrl_tour_length.
// Sorting is stable ascending (ties keep index order) where the reference leaves tie order to numpy's argsort.
#include "kernels.hpp"

namespace eamrl {

namespace {

constexpr int EVB = 256;       // threads
constexpr int EV_MAX = 128;    // max population size and tour length

struct EaArgs {
    const float* locs; int64_t* pop; float* fitness;
    int64_t B; int S, N, G;
    double mutation_rate, crossover_rate, selection_rate;
    const double* cross_rand; const int32_t* cross_idx; const double* mut_rand; const int32_t* mut_idx;
    int ne, P;     // elites, crossover pairs (host-computed with the reference's integer rules)
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// tour length of row `tour` (int16 [N]) by one wavefront; every lane gets the result
__device__ __forceinline__ float wave_tour_length(const int16_t* tour, const float2* loc, int N, int lane)
{
    float total = 0.0f;
    for (int b0 = 0; b0 < N; b0 += 64) {
        const int t = b0 + lane;
        float d = 0.0f;
        if (t < N) {
            const float2 p0 = loc[tour[t]];
            const float2 p1 = loc[tour[(t + 1 == N) ? 0 : t + 1]];
            const float dx = p1.x - p0.x, dy = p1.y - p0.y;
            d = __builtin_sqrtf(fma_(dy, dy, dx * dx));
        }
        const float s = wave_tree_sum(d);
        total = (b0 == 0) ? s : total + s;
    }
    return total;
}

__global__ __launch_bounds__(EVB) void k_ea_tsp(EaArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = a.S, N = a.N, P = a.P, O = 2 * a.P;
    float2* loc = reinterpret_cast<float2*>(smem);                        // [N]
    float* fit = reinterpret_cast<float*>(loc + EV_MAX);                  // [S]
    float* ofit = fit + EV_MAX;                                           // [O]
    int16_t* first = reinterpret_cast<int16_t*>(ofit + EV_MAX);           // [S] first node of position s
    int16_t* order = first + EV_MAX;                                      // [S + O] sort scratch
    int16_t* sel = order + 2 * EV_MAX;                                    // [ne]
    int* flags = reinterpret_cast<int*>(sel + EV_MAX);                    // [0] duplicate first nodes
    int16_t* pop = reinterpret_cast<int16_t*>(flags + 4);                 // [S][N]
    int16_t* off = pop + (size_t)S * N;                                   // [O][N]
    int16_t* tmp = off + (size_t)S * N;                                   // [S][N] (top-k replacement only)

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t b = blockIdx.x;
    const float worst = (float)(1.5 * (double)N);

    // ---- load the instance ------------------------------------------------------------------------------------------
    for (int i = tid; i < N; i += EVB) loc[i] = *reinterpret_cast<const float2*>(a.locs + (b * N + i) * 2);
    for (int i = tid; i < S * N; i += EVB) pop[i] = (int16_t)clampi((int)a.pop[b * S * N + i], 0, N - 1);
    if (tid == 0) flags[0] = 0;
    __syncthreads();
    if (tid < S) first[tid] = pop[tid * N];
    for (int s = wv; s < S; s += EVB / 64) {
        const float len = wave_tour_length(pop + s * N, loc, N, lane);
        if (lane == 0) fit[s] = worst - len;
    }
    __syncthreads();
    if (tid < S) {
        int dup = 0;
        for (int j = 0; j < tid; ++j) dup |= (first[j] == first[tid]);
        if (dup) atomicOr(&flags[0], 1);
    }
    __syncthreads();
    const bool unique_first = flags[0] == 0;

    for (int g = 0; g < a.G && O > 0; ++g) {
        // ---- select: the ne fittest, in ascending fitness order (stable) ----------------------------------------------
        if (S <= 2) {
            if (tid < S) sel[tid] = (int16_t)tid;
        } else {
            if (tid < S) {
                const float f = fit[tid];
                int rank = 0;
                for (int j = 0; j < S; ++j) rank += (fit[j] < f) | ((fit[j] == f) & (j < tid));
                if (rank >= S - a.ne) sel[rank - (S - a.ne)] = (int16_t)tid;
            }
        }
        __syncthreads();

        // ---- order crossover + inversion mutation: thread t builds offspring t -----------------------------------------
        if (tid < O) {
            const int p = tid >> 1, role = tid & 1;
            const int16_t* own = pop + (int)sel[2 * p + role] * N;
            const int16_t* other = pop + (int)sel[2 * p + 1 - role] * N;
            int16_t* o = off + tid * N;
            const int64_t dp = ((int64_t)g * a.B + b) * P + p;
            double rate = a.crossover_rate;
            if (p > 0 && P > 1) {
                rate = ((double)P * a.crossover_rate - 1.0) / (double)(P - 1);
                rate = rate > 1.0 ? 1.0 : rate;
                rate = rate < 0.0 ? 0.0 : rate;
            }
            const double r = (p == 0) ? 0.0 : a.cross_rand[dp];
            if (r < rate) {
                const int i1 = clampi(a.cross_idx[2 * dp], 1, N - 1), i2 = clampi(a.cross_idx[2 * dp + 1], 1, N - 1);
                const int start = i1 < i2 ? i1 : i2, end = i1 < i2 ? i2 : i1;
                unsigned long long used_lo = 0ull, used_hi = 0ull;
                auto mark = [&](int node) { if (node < 64) used_lo |= 1ull << node; else used_hi |= 1ull << (node - 64); };
                auto is_used = [&](int node) { return node < 64 ? (used_lo >> node) & 1ull : (used_hi >> (node - 64)) & 1ull; };
                for (int i = 0; i < N; ++i) o[i] = -1;
                o[0] = own[0];
                mark(own[0]);
                for (int i = start; i < end; ++i) { o[i] = own[i]; mark(own[i]); }
                int pos = end % N, j = 0;
                for (int it = 0; it < N; ++it) {
                    if (pos != 0 && o[pos] == -1) {
                        while (j < N && is_used(other[j])) ++j;      // first node of `other` not yet in the child
                        if (j < N) { o[pos] = other[j]; mark(other[j]); ++j; }
                    }
                    pos = (pos + 1 == N) ? 0 : pos + 1;
                }
                for (int i = 0; i < N; ++i) if (o[i] < 0) o[i] = 0;   // only reachable with non-permutation input
            } else {
                for (int i = 0; i < N; ++i) o[i] = own[i];
            }
            const int64_t dm = ((int64_t)g * a.B + b) * O + tid;
            if (a.mut_rand[dm] < a.mutation_rate) {
                const int i1 = clampi(a.mut_idx[2 * dm], 1, N - 1), i2 = clampi(a.mut_idx[2 * dm + 1], 1, N - 1);
                const int start = i1 < i2 ? i1 : i2, end = i1 < i2 ? i2 : i1;
                if (start < end) {
                    for (int lo = start, hi = end - 1; lo < hi; ++lo, --hi) { const int16_t x = o[lo]; o[lo] = o[hi]; o[hi] = x; }
                } else if (start < N - 1) {
                    const int16_t x = o[start]; o[start] = o[start + 1]; o[start + 1] = x;
                }
            }
        }
        __syncthreads();

        // ---- fitness of the offspring -----------------------------------------------------------------------------------
        for (int t = wv; t < O; t += EVB / 64) {
            const float len = wave_tour_length(off + t * N, loc, N, lane);
            if (lane == 0) ofit[t] = worst - len;
        }
        __syncthreads();

        // ---- replacement ----------------------------------------------------------------------------------------------
        if (unique_first) {
            // position s keeps the best of pop[s] and the offspring starting at its node; earliest wins ties
            if (tid < S) {
                float best = fit[tid];
                int src = -1;
                for (int t = 0; t < O; ++t)
                    if (off[t * N] == first[tid] && ofit[t] > best) { best = ofit[t]; src = t; }
                order[tid] = (int16_t)src;
                if (src >= 0) fit[tid] = best;
            }
            __syncthreads();
            for (int i = tid; i < S * N; i += EVB) {
                const int s = i / N, src = order[s];
                if (src >= 0) pop[i] = off[src * N + (i - s * N)];
            }
        } else {
            // the S fittest of pop ++ offspring, descending = reversed stable ascending order
            const int C = S + O;
            if (tid < C) {
                const float f = tid < S ? fit[tid] : ofit[tid - S];
                int rank = 0;
                for (int j = 0; j < C; ++j) {
                    const float fj = j < S ? fit[j] : ofit[j - S];
                    rank += (fj < f) | ((fj == f) & (j < tid));
                }
                order[tid] = (int16_t)(C - 1 - rank);           // position in descending order
            }
            __syncthreads();
            for (int i = tid; i < C * N; i += EVB) {
                const int c = i / N, dst = order[c];
                if (dst < S) tmp[dst * N + (i - c * N)] = c < S ? pop[i] : off[i - S * N];
            }
            float keep = 0.0f;
            int dst = S;
            if (tid < C) { dst = order[tid]; keep = tid < S ? fit[tid] : ofit[tid - S]; }
            __syncthreads();
            if (dst < S) fit[dst] = keep;
            for (int i = tid; i < S * N; i += EVB) pop[i] = tmp[i];
        }
        __syncthreads();
    }

    for (int i = tid; i < S * N; i += EVB) a.pop[b * S * N + i] = pop[i];
    if (tid < S) a.fitness[b * S + tid] = fit[tid];
}

// ================================================================================================================
// CVRP
// ================================================================================================================
struct EaCvrpArgs {
    const float* locs; const float* demand; const float* vcap; int64_t* pop; float* fitness;
    int64_t B; int S, N, L, G, top_k;
    double mutation_rate, crossover_rate;
    const double* init_mut_rand; const double* init_mut_u;
    const double* cross_rand; const double* cross_u; const double* mut_rand; const double* mut_u;
    int ne, P;
};

// np.random.randint(lo, hi) from a uniform u in [0, 1): lo + min(floor(u * (hi - lo)), hi - lo - 1)
__device__ __forceinline__ int rint_u(int lo, int hi, double u)
{
    const int n = hi - lo;
    int k = (int)(u * (double)n);
    k = k > n - 1 ? n - 1 : (k < 0 ? 0 : k);
    return lo + k;
}

// closed tour depot -> row -> depot (L + 1 legs) by one wavefront; every lane gets the result
__device__ __forceinline__ float wave_route_length(const int16_t* row, const float2* loc, int L, int lane)
{
    float total = 0.0f;
    for (int b0 = 0; b0 <= L; b0 += 64) {
        const int t = b0 + lane;
        float d = 0.0f;
        if (t <= L) {
            const float2 p0 = loc[t == 0 ? 0 : row[t - 1]];
            const float2 p1 = loc[t == L ? 0 : row[t]];
            const float dx = p1.x - p0.x, dy = p1.y - p0.y;
            d = __builtin_sqrtf(fma_(dy, dy, dx * dx));
        }
        const float s = wave_tree_sum(d);
        total = (b0 == 0) ? s : total + s;
    }
    return total;
}

// inverse_mutate_cvrp on one row: reverse a random segment strictly inside one route
__device__ __forceinline__ void cvrp_mutate_row(int16_t* o, int L, const double* u3)
{
    int depots = 0;
    for (int j = 0; j < L; ++j) depots += (o[j] == 0);
    if (depots <= 1) return;
    const int r = rint_u(0, depots - 1, u3[0]);
    int seen = 0, z0 = -1, z1 = -1;
    for (int j = 0; j < L; ++j) {
        if (o[j] == 0) {
            if (seen == r) z0 = j;
            if (seen == r + 1) { z1 = j; break; }
            ++seen;
        }
    }
    const int start = z0 + 1, end = z1 - 1;
    if (end - start > 1) {
        const int s0 = rint_u(start, end, u3[1]);
        const int s1 = rint_u(s0 + 1, end + 1, u3[2]);
        for (int lo = s0, hi = s1 - 1; lo < hi; ++lo, --hi) { const int16_t x = o[lo]; o[lo] = o[hi]; o[hi] = x; }
    }
}

// routes of a parent as the crossover counts them: zeros before the last non-zero entry
__device__ __forceinline__ int cvrp_route_num(const int16_t* par, int L)
{
    int valid_end = 1;
    for (int j = L - 1; j >= 0; --j) if (par[j] != 0) { valid_end = j + 1; break; }
    int zeros = 0;
    for (int j = 0; j < valid_end; ++j) zeros += (par[j] == 0);
    return zeros;
}

__global__ __launch_bounds__(EVB) void k_ea_cvrp(EaCvrpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = a.S, N = a.N, L = a.L, M = a.N + 1, P = a.P, O = 2 * a.P;
    float2* loc = reinterpret_cast<float2*>(smem);                        // [M <= 128]
    float* dem = reinterpret_cast<float*>(loc + EV_MAX);                  // [N]
    float* fit = dem + EV_MAX;                                            // [S]
    float* ofit = fit + EV_MAX;                                           // [O]
    int16_t* first = reinterpret_cast<int16_t*>(ofit + EV_MAX);           // [S]
    int16_t* order = first + EV_MAX;                                      // [S + O]
    int16_t* sel = order + 2 * EV_MAX;                                    // [ne]
    int* flags = reinterpret_cast<int*>(sel + EV_MAX);
    int16_t* pop = reinterpret_cast<int16_t*>(flags + 4);                 // [S][L]
    int16_t* off = pop + (size_t)S * L;                                   // [O][L]
    int16_t* tmp = off + (size_t)S * L;                                   // [S][L] (top-k replacement only)

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t b = blockIdx.x;
    const float worst = (float)(2.5 * (double)L);
    const double vcap = (double)a.vcap[b];

    for (int i = tid; i < M; i += EVB) loc[i] = *reinterpret_cast<const float2*>(a.locs + (b * M + i) * 2);
    for (int i = tid; i < N; i += EVB) dem[i] = a.demand[b * N + i];
    for (int i = tid; i < S * L; i += EVB) pop[i] = (int16_t)clampi((int)a.pop[b * S * L + i], 0, N);
    if (tid == 0) flags[0] = 0;
    __syncthreads();
    if (tid < S) {
        first[tid] = pop[tid * L];                                        // node_to_position uses the INITIAL first nodes
        if (a.init_mut_rand[b * S + tid] < a.mutation_rate) cvrp_mutate_row(pop + tid * L, L, a.init_mut_u + (b * S + tid) * 3);
    }
    __syncthreads();
    for (int s = wv; s < S; s += EVB / 64) {
        const float len = wave_route_length(pop + s * L, loc, L, lane);
        if (lane == 0) fit[s] = worst - len;
    }
    if (tid < S) {
        int dup = 0;
        for (int j = 0; j < tid; ++j) dup |= (first[j] == first[tid]);
        if (dup) atomicOr(&flags[0], 1);
    }
    __syncthreads();
    const bool by_first = flags[0] == 0 && !a.top_k;

    for (int g = 0; g < a.G && O > 0; ++g) {
        if (S <= 2) {
            if (tid < S) sel[tid] = (int16_t)tid;
        } else if (tid < S) {
            const float f = fit[tid];
            int rank = 0;
            for (int j = 0; j < S; ++j) rank += (fit[j] < f) | ((fit[j] == f) & (j < tid));
            if (rank >= S - a.ne) sel[rank - (S - a.ne)] = (int16_t)tid;
        }
        __syncthreads();

        if (tid < O) {
            const int p = tid >> 1, role = tid & 1;
            const int16_t* own = pop + (int)sel[2 * p + role] * L;
            const int16_t* other = pop + (int)sel[2 * p + 1 - role] * L;
            int16_t* o = off + tid * L;
            const int64_t dp = ((int64_t)g * a.B + b) * P + p;
            double rate = a.crossover_rate;
            if (p > 0 && P > 1) {
                rate = ((double)P * a.crossover_rate - 1.0) / (double)(P - 1);
                rate = rate > 1.0 ? 1.0 : rate;
                rate = rate < 0.0 ? 0.0 : rate;
            }
            const double r = (p == 0) ? 0.0 : a.cross_rand[dp];
            bool keep_parent = !(r < rate);
            if (!keep_parent) {
                const int m0 = cvrp_route_num(own, L), m1 = cvrp_route_num(other, L);
                const int m = m0 < m1 ? m0 : m1;
                const int end = m > 1 ? rint_u(1, m, a.cross_u[dp]) : 0;
                int end_idx = 0;
                if (end > 0) {
                    int seen = 0;
                    for (int j = 0; j < L; ++j) if (own[j] == 0) { if (seen == end) { end_idx = j; break; } ++seen; }
                }
                unsigned long long used_lo = 0ull, used_hi = 0ull;
                bool dz = false;                               // two consecutive depot visits somewhere in the child
                int pos = 0, last = -1;
                auto push = [&](int x) {
                    if (pos < L) o[pos] = (int16_t)x;
                    if (pos >= 1 && x == 0 && last == 0) dz = true;
                    last = x;
                    ++pos;
                };
                for (int j = 0; j < end_idx; ++j) {
                    const int x = own[j];
                    if (x > 0) { if (x < 64) used_lo |= 1ull << x; else used_hi |= 1ull << (x - 64); }
                    push(x);
                }
                if (pos > 0 && last != 0) push(0);
                const int count = N - (__builtin_popcountll(used_lo) + __builtin_popcountll(used_hi));
                double load = 0.0;
                int i = 0, first_unused = 0;
                for (int node = 1; node <= N; ++node) {
                    const bool used = node < 64 ? (used_lo >> node) & 1ull : (used_hi >> (node - 64)) & 1ull;
                    if (used) continue;
                    if (first_unused == 0) first_unused = node;
                    if (pos >= 2 * L - 1) break;
                    const double d = (double)dem[node - 1];
                    if (load + d > vcap) {
                        if (pos > 0 && last == 0 && i < count - 1) { ++i; continue; }
                        push(0);
                        load = 0.0;
                        if (pos >= 2 * L - 1) break;
                    }
                    push(node);
                    load += d;
                    ++i;
                }
                if (pos < 2 * L && last != 0) {
                    // the reference's "all visited" test never sees the nodes it just appended: it fails exactly when
                    // the smallest customer missing from the copied prefix has an index below the number of missing ones
                    if (first_unused == 0)
                        for (int node = 1; node <= N && first_unused == 0; ++node) {
                            const bool used = node < 64 ? (used_lo >> node) & 1ull : (used_hi >> (node - 64)) & 1ull;
                            if (!used) first_unused = node;
                        }
                    const bool all_visited = (count == 0) || !(first_unused < count);
                    if (all_visited) push(0);
                }
                if ((dz && count > 0) || pos - 1 >= L) keep_parent = true;
                else for (int j = pos; j < L; ++j) o[j] = 0;
            }
            if (keep_parent) for (int j = 0; j < L; ++j) o[j] = own[j];
            const int64_t dm = ((int64_t)g * a.B + b) * O + tid;
            if (a.mut_rand[dm] < a.mutation_rate) cvrp_mutate_row(o, L, a.mut_u + dm * 3);
        }
        __syncthreads();

        for (int t = wv; t < O; t += EVB / 64) {
            const float len = wave_route_length(off + t * L, loc, L, lane);
            if (lane == 0) ofit[t] = worst - len;
        }
        __syncthreads();

        if (by_first) {
            if (tid < S) {
                float best = fit[tid];
                int src = -1;
                for (int t = 0; t < O; ++t)
                    if (off[t * L] == first[tid] && ofit[t] > best) { best = ofit[t]; src = t; }
                order[tid] = (int16_t)src;
                if (src >= 0) fit[tid] = best;
            }
            __syncthreads();
            for (int i = tid; i < S * L; i += EVB) {
                const int s = i / L, src = order[s];
                if (src >= 0) pop[i] = off[src * L + (i - s * L)];
            }
        } else {
            const int C = S + O;
            if (tid < C) {
                const float f = tid < S ? fit[tid] : ofit[tid - S];
                int rank = 0;
                for (int j = 0; j < C; ++j) {
                    const float fj = j < S ? fit[j] : ofit[j - S];
                    rank += (fj < f) | ((fj == f) & (j < tid));
                }
                order[tid] = (int16_t)(C - 1 - rank);
            }
            __syncthreads();
            for (int i = tid; i < C * L; i += EVB) {
                const int c = i / L, dst = order[c];
                if (dst < S) tmp[dst * L + (i - c * L)] = c < S ? pop[i] : off[i - S * L];
            }
            float keep = 0.0f;
            int dst = S;
            if (tid < C) { dst = order[tid]; keep = tid < S ? fit[tid] : ofit[tid - S]; }
            __syncthreads();
            if (dst < S) fit[dst] = keep;
            for (int i = tid; i < S * L; i += EVB) pop[i] = tmp[i];
        }
        __syncthreads();
    }

    for (int i = tid; i < S * L; i += EVB) a.pop[b * S * L + i] = pop[i];
    if (tid < S) a.fitness[b * S + tid] = fit[tid];
}

}  // namespace

int launch_ea_tsp(const float* locs, int64_t* pop, float* fitness, int64_t B, int S, int N, int G, double mutation_rate,
                  double crossover_rate, double selection_rate, const double* cross_rand, const int32_t* cross_idx,
                  const double* mut_rand, const int32_t* mut_idx, hipStream_t st)
{
    EaArgs a;
    a.locs = locs; a.pop = pop; a.fitness = fitness; a.B = B; a.S = S; a.N = N; a.G = G;
    a.mutation_rate = mutation_rate; a.crossover_rate = crossover_rate; a.selection_rate = selection_rate;
    a.cross_rand = cross_rand; a.cross_idx = cross_idx; a.mut_rand = mut_rand; a.mut_idx = mut_idx;
    int ne = S;
    if (S > 2) {
        ne = (int)(selection_rate * (double)S);          // int(selection_rate * pop.shape[0]); idx[-0:] is everything
        if (ne <= 0) ne = S;
        if (ne > S) ne = S;
    }
    a.ne = ne;
    a.P = ne / 2;
    const size_t lds = EV_MAX * sizeof(float2) + 2 * EV_MAX * sizeof(float) + 5 * EV_MAX * sizeof(int16_t) + 16 +
                       3 * (size_t)S * N * sizeof(int16_t);
    auto k = k_ea_tsp;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return EAMRL_E_LAUNCH;
    hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(EVB), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_ea_cvrp(const float* locs, const float* demand, const float* vcap, int64_t* pop, float* fitness, int64_t B, int S,
                   int N, int L, int G, double mutation_rate, double crossover_rate, double selection_rate, int top_k,
                   const double* init_mut_rand, const double* init_mut_u, const double* cross_rand, const double* cross_u,
                   const double* mut_rand, const double* mut_u, hipStream_t st)
{
    EaCvrpArgs a;
    a.locs = locs; a.demand = demand; a.vcap = vcap; a.pop = pop; a.fitness = fitness;
    a.B = B; a.S = S; a.N = N; a.L = L; a.G = G; a.top_k = top_k;
    a.mutation_rate = mutation_rate; a.crossover_rate = crossover_rate;
    a.init_mut_rand = init_mut_rand; a.init_mut_u = init_mut_u; a.cross_rand = cross_rand; a.cross_u = cross_u;
    a.mut_rand = mut_rand; a.mut_u = mut_u;
    int ne = S;
    if (S > 2) {
        ne = (int)(selection_rate * (double)S);
        if (ne <= 0) ne = S;
        if (ne > S) ne = S;
    }
    a.ne = ne;
    a.P = ne / 2;
    const size_t lds = EV_MAX * sizeof(float2) + 3 * EV_MAX * sizeof(float) + 5 * EV_MAX * sizeof(int16_t) + 16 +
                       3 * (size_t)S * L * sizeof(int16_t);
    if (lds > 150 * 1024) return EAMRL_E_ARG;
    auto k = k_ea_cvrp;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return EAMRL_E_LAUNCH;
    hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(EVB), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

}  // namespace eamrl
