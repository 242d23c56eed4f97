// Evolutionary improvement of PCTSP / OP tour populations (the fork's EA.run for the prize-collecting envs), one
// workgroup per problem instance; same skeleton as evolution.hip (population, offspring and fitness live in LDS for all
// generations, every random draw is an input).
//
// Reference (numba on CPU threads):
//   EA.run ................... rl4co/models/zoo/earl/evolution.py:252-354
//   inverse_mutate_pctsp ..... :555-583      cycle_crossover_pctsp .. :905-1101    calculate_fitness_cvrp (PCTSP) :364-370
//   order_crossover_op ....... :1110-1346    inverse_mutate_op ...... :1468-1572   calculate_fitness_op .......... :372-378
//   elitism_selection ........ :1103-1108
//
// Arithmetic restated from numba's typing (oracle/ea_oracle.py has the same notes): float32 array elements added to a
// `0.0` accumulator are summed in float64; PCTSP's prize/penalty ratios are rounded to float32 when stored; OP's distance
// matrix is float32 sqrt(dx*dx + dy*dy) with every operation rounded (no fused multiply-add).
// Defined choices: stable ascending sort where the reference leaves ties to numpy's argsort; the cycle crossover's
// `next(iter(set))` starts at the smallest remaining node (slot == value in CPython's and numba's tables for small ints);
// customers the OP crossover would look up beyond the node count (it reads out of bounds there) do not exist.
// Integer results are bit-exact against oracle/ea_oracle.py; fitness uses the canonical reward arithmetic of
// eamrl_pctsp_reward / eamrl_op_reward (lane-tree sums).
#include "kernels.hpp"

namespace eamrl {

namespace {

constexpr int EVB = 256;       // threads
constexpr int EV_MAX = 128;    // max population size, nodes and tour length

enum { PRIZE_PCTSP = 0, PRIZE_OP = 1 };

struct EaPrizeArgs {
    const float* locs; const float* prize; const float* aux;      // aux: PCTSP penalty [B,M], OP max_length [B,M]
    int64_t* pop; float* fitness;
    int64_t B; int S, N, L, G, top_k;
    double mutation_rate, crossover_rate;
    const double* init_mut_rand; const double* init_mut_u;        // [B,S], [B,S,2]
    const double* cross_rand; const double* cross_u;              // [G,B,P], [G,B,P] (OP only)
    const double* mut_rand; const double* mut_u;                  // [G,B,O], [G,B,O,2]
    int ne, P;
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// np.random.randint(lo, hi) from a uniform u in [0, 1): lo + min(floor(u * (hi - lo)), hi - lo - 1)
__device__ __forceinline__ int rint_u(int lo, int hi, double u)
{
    const int n = hi - lo;
    int k = (int)(u * (double)n);
    k = k > n - 1 ? n - 1 : (k < 0 ? 0 : k);
    return lo + k;
}

struct Bits128 {
    unsigned long long lo = 0ull, hi = 0ull;
    __device__ __forceinline__ bool test(int i) const { return i < 64 ? (lo >> i) & 1ull : (hi >> (i - 64)) & 1ull; }
    __device__ __forceinline__ void set(int i) { if (i < 64) lo |= 1ull << i; else hi |= 1ull << (i - 64); }
    __device__ __forceinline__ void clear(int i) { if (i < 64) lo &= ~(1ull << i); else hi &= ~(1ull << (i - 64)); }
    __device__ __forceinline__ bool any() const { return (lo | hi) != 0ull; }
    __device__ __forceinline__ int lowest() const { return lo ? __builtin_ctzll(lo) : 64 + __builtin_ctzll(hi); }
};

// closed tour depot -> row -> depot (L + 1 legs) by one wavefront, canonical leg and lane-tree order (eamrl_tour_length)
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

// lane tree over v[row[t]], t < L (orc lane_tree: 64-blocks summed ascending)
__device__ __forceinline__ float wave_gather_sum(const int16_t* row, const float* v, int L, int lane)
{
    float total = 0.0f;
    for (int b0 = 0; b0 < L; b0 += 64) {
        const int t = b0 + lane;
        const float s = wave_tree_sum(t < L ? v[row[t]] : 0.0f);
        total = (b0 == 0) ? s : total + s;
    }
    return total;
}

// float32 distance of EA.run's calculate_distance_matrix: sqrt(dx*dx + dy*dy), each operation rounded
__device__ __forceinline__ double dist32(const float2* loc, int a, int b)
{
    const float dx = __fsub_rn(loc[a].x, loc[b].x), dy = __fsub_rn(loc[a].y, loc[b].y);
    return (double)__fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
}

// index after the last non-zero entry at index >= 1; L when there is none
__device__ __forceinline__ int valid_end_from_one(const int16_t* row, int L)
{
    for (int j = L - 1; j >= 1; --j) if (row[j] != 0) return j + 1;
    return L;
}

// ---- PCTSP ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pctsp_mutate_row(int16_t* o, int L, const double* u2)
{
    int v = -1;
    for (int j = L - 1; j >= 0; --j) if (o[j] != 0) { v = j; break; }
    if (v < 2) return;
    const int i1 = rint_u(1, v, u2[0]), i2 = rint_u(1, v, u2[1]);
    const int start = i1 < i2 ? i1 : i2, end = i1 < i2 ? i2 : i1;
    if (start < end) {
        for (int lo = start, hi = end - 1; lo < hi; ++lo, --hi) { const int16_t x = o[lo]; o[lo] = o[hi]; o[hi] = x; }
    } else if (start < L - 1) {
        const int16_t x = o[start]; o[start] = o[start + 1]; o[start + 1] = x;
    }
}

// One child of cycle_crossover_pctsp.  p1 / p2 are the pair's parents IN PAIR ORDER (the cycles are found from p1's side for
// both children); role 0 -> o1, role 1 -> o2.  scratch: 2 * EV_MAX bytes of this thread.
__device__ __forceinline__ void pctsp_child(const int16_t* p1, const int16_t* p2, int role, int16_t* o, int L, int M,
                                            const float* prize, const float* penalty, uint8_t* scratch)
{
    uint8_t* pos2 = scratch;            // [M] position of a node in p2's customer list, 0xFF = absent
    uint8_t* p1c = scratch + EV_MAX;    // [n1] p1's customers in order
    for (int i = 0; i < M; ++i) pos2[i] = 0xFF;
    int e1 = 0, e2 = 0;
    for (int j = L - 1; j >= 0; --j) if (p1[j] != 0) { e1 = j + 1; break; }
    for (int j = L - 1; j >= 0; --j) if (p2[j] != 0) { e2 = j + 1; break; }
    int n1 = 0, n2 = 0;
    Bits128 remaining;
    for (int j = 0; j < e1; ++j) if (p1[j] > 0) { p1c[n1++] = (uint8_t)p1[j]; remaining.set(p1[j]); }
    for (int j = 0; j < e2; ++j) if (p2[j] > 0) pos2[p2[j]] = (uint8_t)n2++;

    Bits128 used;
    double total = 0.0;
    int count = 0;
    auto emit = [&](int node) {
        if (node > 0 && !used.test(node)) {
            if (count < L) o[count] = (int16_t)node;
            ++count;
            used.set(node);
            total += (double)prize[node];
        }
    };
    for (int c = 0; remaining.any(); ++c) {
        const int start = remaining.lowest();
        int node = start;
        while (true) {
            remaining.clear(node);
            // o1 receives every node of every cycle; o2 the whole odd cycles and, of the even ones, the nodes p2 also visits
            if (role == 0 || (c & 1) || pos2[node] != 0xFF) emit(node);
            if (pos2[node] == 0xFF) break;
            const int k = pos2[node];
            if (k >= n1) break;
            node = p1c[k];
            if (node == start || !remaining.test(node)) break;
        }
    }
    const int N = M - 1;
    while (total < 1.0 - 1e-5) {
        int best = 0;
        double best_ratio = -1.0;
        for (int i = 1; i <= N; ++i) {
            if (used.test(i)) continue;
            const float ratio = (float)((double)prize[i] / ((double)penalty[i] + 1e-10));
            if ((double)ratio > best_ratio) { best_ratio = (double)ratio; best = i; }
        }
        if (best == 0) break;
        emit(best);
    }
    for (int j = count < L ? count : L; j < L; ++j) o[j] = 0;
}

// ---- OP ---------------------------------------------------------------------------------------------------------------
// One child of order_crossover_op from `own` with the shared cut `end`; false = keep the parent.
__device__ __forceinline__ bool op_child(const int16_t* own, int end, int16_t* o, int L, int M, const float2* loc,
                                         double global_max)
{
    const double safe = global_max - 0.1;
    Bits128 used;
    for (int j = 0; j < end; ++j) { o[j] = own[j]; if (own[j] != 0) used.set(own[j]); }
    double cur = 0.0;
    for (int j = 1; j < end; ++j) cur += dist32(loc, o[j - 1], o[j]);
    cur += dist32(loc, 0, o[0]);
    int pos = end, last = o[end - 1];
    for (int node = 1; node <= L; ++node) {
        if (node >= M || used.test(node)) continue;
        const double nxt = dist32(loc, last, node), back = dist32(loc, node, 0);
        if (cur + nxt + back <= safe) {
            if (pos < L) o[pos] = (int16_t)node;
            cur += nxt;
            used.set(node);
            last = node;
            ++pos;
        }
        if (pos >= 2 * L - 2) break;
    }
    if (pos >= L) return false;                     // the closing depot visit would land at index >= L
    for (int j = pos; j < L; ++j) o[j] = 0;
    // post-check: closed by a depot visit, legs 1.. within max - 1e-5, no customer twice
    int ve = valid_end_from_one(o, L);
    if (o[ve - 1] != 0) {
        if (ve < L) { o[ve] = 0; ++ve; } else o[ve - 1] = 0;
    }
    double total = 0.0;
    bool dup = false;
    Bits128 seen;
    for (int j = 1; j < ve; ++j) {
        total += dist32(loc, o[j - 1], o[j]);
        if (o[j] != 0) {
            if (seen.test(o[j])) { dup = true; break; }
            seen.set(o[j]);
        }
    }
    return total <= global_max - 1e-5 && !dup;
}

__device__ __forceinline__ void op_mutate_row(int16_t* row, int L, const float2* loc, double global_max, const double* u2)
{
    const double safe = global_max - 1e-5;
    int ve = valid_end_from_one(row, L);
    if (ve <= 3) return;
    int cz = -1;
    int16_t cz_old = 0;
    if (row[ve - 1] != 0) {
        if (ve < L) { cz = ve; cz_old = row[ve]; row[ve] = 0; ++ve; }
        else { cz = ve - 1; cz_old = row[ve - 1]; row[ve - 1] = 0; }
    }
    double cur = dist32(loc, 0, row[0]);
    for (int j = 1; j < ve; ++j) cur += dist32(loc, row[j - 1], row[j]);
    const int s = rint_u(1, ve - 2, u2[0]);
    const int e = rint_u(s + 1, ve - 1, u2[1]);
    bool success = false;
    if (s < e) {
        auto t = [&](int j) -> int { return (j >= s && j <= e) ? row[s + e - j] : row[j]; };
        double old_sub = 0.0, new_sub = 0.0;
        for (int j = s; j < e; ++j) old_sub += dist32(loc, row[j], row[j + 1]);
        double old_conn = 0.0;
        old_conn += dist32(loc, row[s - 1], row[s]);
        if (e < ve - 1) old_conn += dist32(loc, row[e], row[e + 1]);
        for (int j = s; j < e; ++j) new_sub += dist32(loc, t(j), t(j + 1));
        double new_conn = 0.0;
        new_conn += dist32(loc, t(s - 1), t(s));
        if (e < ve - 1) new_conn += dist32(loc, t(e), t(e + 1));
        const double change = (new_sub + new_conn) - (old_sub + old_conn);
        if (cur + change <= safe) {
            bool dup = false;
            Bits128 seen;
            for (int j = 0; j < ve; ++j) {
                const int x = t(j);
                if (x != 0) {
                    if (seen.test(x)) { dup = true; break; }
                    seen.set(x);
                }
            }
            double total = dist32(loc, 0, t(0));
            for (int j = 1; j < ve; ++j) total += dist32(loc, t(j - 1), t(j));
            if (total <= safe && !dup) {
                for (int lo = s, hi = e; lo < hi; ++lo, --hi) { const int16_t x = row[lo]; row[lo] = row[hi]; row[hi] = x; }
                success = true;
            }
        }
    }
    if (!success && cz >= 0) row[cz] = cz_old;
}

template <int ENV>
__global__ __launch_bounds__(EVB) void k_ea_prize(EaPrizeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = a.S, L = a.L, M = a.N + 1, P = a.P, O = 2 * a.P;
    float2* loc = reinterpret_cast<float2*>(smem);                        // [M <= 128]
    float* prize = reinterpret_cast<float*>(loc + EV_MAX);                // [M]
    float* aux = prize + EV_MAX;                                          // [M] penalty (PCTSP) / max_length (OP)
    float* fit = aux + EV_MAX;                                            // [S]
    float* ofit = fit + EV_MAX;                                           // [O]
    int16_t* first = reinterpret_cast<int16_t*>(ofit + EV_MAX);           // [S]
    int16_t* order = first + EV_MAX;                                      // [S + O]
    int16_t* sel = order + 2 * EV_MAX;                                    // [ne]
    int* flags = reinterpret_cast<int*>(sel + EV_MAX);
    int16_t* pop = reinterpret_cast<int16_t*>(flags + 4);                 // [S][L]
    int16_t* off = pop + (size_t)S * L;                                   // [O][L]
    int16_t* tmp = off + (size_t)S * L;                                   // [S][L] (top-k replacement only)
    uint8_t* scratch = reinterpret_cast<uint8_t*>(tmp + (size_t)S * L);   // [O][2 * EV_MAX] (PCTSP crossover)

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t b = blockIdx.x;
    const float worst = ENV == PRIZE_PCTSP ? (float)(2.5 * (double)L) : 0.0f;

    for (int i = tid; i < M; i += EVB) {
        loc[i] = *reinterpret_cast<const float2*>(a.locs + (b * M + i) * 2);
        prize[i] = a.prize[b * M + i];
        aux[i] = a.aux[b * M + i];
    }
    for (int i = tid; i < S * L; i += EVB) pop[i] = (int16_t)clampi((int)a.pop[b * S * L + i], 0, M - 1);
    if (tid == 0) flags[0] = 0;
    __syncthreads();
    const double global_max = (double)aux[0];                             // OP: td["max_length"][0], the depot's entry
    float pen_total = 0.0f;
    if (ENV == PRIZE_PCTSP) {
        // penalties of all customers, lane tree over aux[1..M-1] (every wavefront computes the same value)
        for (int b0 = 0; b0 < M - 1; b0 += 64) {
            const int t = b0 + lane;
            const float s = wave_tree_sum(t < M - 1 ? aux[1 + t] : 0.0f);
            pen_total = (b0 == 0) ? s : pen_total + s;
        }
    }
    auto fitness_of = [&](const int16_t* row) -> float {
        if (ENV == PRIZE_PCTSP) {
            const float len = wave_route_length(row, loc, L, lane);
            const float saved = wave_gather_sum(row, aux, L, lane);
            const float reward = saved - (len + pen_total);
            return worst - (0.0f - reward);
        }
        return worst - (0.0f - wave_gather_sum(row, prize, L, lane));
    };

    if (tid < S) {
        first[tid] = pop[tid * L];                                        // node_to_position uses the INITIAL first nodes
        if (a.init_mut_rand[b * S + tid] < a.mutation_rate) {
            const double* u2 = a.init_mut_u + (b * S + tid) * 2;
            if (ENV == PRIZE_PCTSP) pctsp_mutate_row(pop + tid * L, L, u2);
            else op_mutate_row(pop + tid * L, L, loc, global_max, u2);
        }
    }
    __syncthreads();
    for (int s = wv; s < S; s += EVB / 64) {
        const float f = fitness_of(pop + s * L);
        if (lane == 0) fit[s] = f;
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
            const int16_t* p1 = pop + (int)sel[2 * p] * L;
            const int16_t* p2 = pop + (int)sel[2 * p + 1] * L;
            const int16_t* own = role ? p2 : p1;
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
                if (ENV == PRIZE_PCTSP) {
                    pctsp_child(p1, p2, role, o, L, M, prize, aux, scratch + (size_t)tid * 2 * EV_MAX);
                } else {
                    const int e1 = valid_end_from_one(p1, L), e2 = valid_end_from_one(p2, L);
                    int max_cross = e1 - 1 < e2 - 1 ? e1 - 1 : e2 - 1;
                    max_cross = max_cross < L - 1 ? max_cross : L - 1;
                    if (p1[e1 - 1] != 0 || p2[e2 - 1] != 0 || max_cross <= 1) keep_parent = true;
                    else keep_parent = !op_child(own, rint_u(1, max_cross, a.cross_u[dp]), o, L, M, loc, global_max);
                }
            }
            if (keep_parent) for (int j = 0; j < L; ++j) o[j] = own[j];
            const int64_t dm = ((int64_t)g * a.B + b) * O + tid;
            if (a.mut_rand[dm] < a.mutation_rate) {
                if (ENV == PRIZE_PCTSP) pctsp_mutate_row(o, L, a.mut_u + dm * 2);
                else op_mutate_row(o, L, loc, global_max, a.mut_u + dm * 2);
            }
        }
        __syncthreads();

        for (int t = wv; t < O; t += EVB / 64) {
            const float f = fitness_of(off + t * L);
            if (lane == 0) ofit[t] = f;
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

int launch_ea_prize(int env, const float* locs, const float* prize, const float* aux, int64_t* pop, float* fitness, int64_t B,
                    int S, int N, int L, int G, double mutation_rate, double crossover_rate, double selection_rate, int top_k,
                    const double* init_mut_rand, const double* init_mut_u, const double* cross_rand, const double* cross_u,
                    const double* mut_rand, const double* mut_u, hipStream_t st)
{
    EaPrizeArgs a;
    a.locs = locs; a.prize = prize; a.aux = aux; a.pop = pop; a.fitness = fitness;
    a.B = B; a.S = S; a.N = N; a.L = L; a.G = G; a.top_k = top_k;
    a.mutation_rate = mutation_rate; a.crossover_rate = crossover_rate;
    a.init_mut_rand = init_mut_rand; a.init_mut_u = init_mut_u; a.cross_rand = cross_rand; a.cross_u = cross_u;
    a.mut_rand = mut_rand; a.mut_u = mut_u;
    int ne = S;
    if (S > 2) {
        ne = (int)(selection_rate * (double)S);          // int(selection_rate * pop.shape[0]); idx[-0:] is everything
        if (ne <= 0) ne = S;
        if (ne > S) ne = S;
    }
    a.ne = ne;
    a.P = ne / 2;
    const bool pctsp = env == EAMRL_ENV_PCTSP;
    const size_t lds = EV_MAX * sizeof(float2) + 4 * EV_MAX * sizeof(float) + 5 * EV_MAX * sizeof(int16_t) + 16 +
                       3 * (size_t)S * L * sizeof(int16_t) + (pctsp ? (size_t)2 * a.P * 2 * EV_MAX : 0);
    if (lds > 150 * 1024) return EAMRL_E_ARG;
    auto k = pctsp ? k_ea_prize<PRIZE_PCTSP> : k_ea_prize<PRIZE_OP>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return EAMRL_E_LAUNCH;
    hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(EVB), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

}  // namespace eamrl
