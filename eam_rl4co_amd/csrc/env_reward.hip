// Stand-alone environment transitions, reward and validity kernels (integer / byte work, HBM-bound).
//
//   k_tsp_step         TSPEnv._step                        rl4co/envs/routing/tsp/env.py:62-88
//   k_cvrp_step_mask   CVRPEnv._step + get_action_mask     rl4co/envs/routing/cvrp/env.py:68-100,132-144
//   k_sdvrp_step_mask  SDVRPEnv._step + get_action_mask    rl4co/envs/routing/sdvrp/env.py:58-92,137-146
//   k_pctsp_step_mask  PCTSPEnv._step + get_action_mask    rl4co/envs/routing/pctsp/env.py:64-97,156-163
//   k_tour_length      get_reward (+ PCTSP penalties)                               rl4co/utils/ops.py:59-95, tsp/env.py:152-159, cvrp/env.py:146-155
//   k_sum_logp         get_log_likelihood                  rl4co/utils/decoding.py:38-64
//   k_check_*          check_solution_validity             tsp/env.py:161-168, cvrp/env.py:157-185
//
// One 64-lane wavefront per row: the row's mask / visited bytes are read as one coalesced run, the
// "any / all" reductions are wave ballots, and the tour length uses the canonical lane tree.
#include "kernels.hpp"

namespace eamrl {

constexpr int EB = 256;          // threads per block
constexpr int ROWS_PER_BLOCK = EB / 64;

__global__ __launch_bounds__(EB) void k_tsp_step(uint8_t* mask, int64_t* first, int64_t* cur, int64_t* istep,
                                                 const int64_t* action, uint8_t* done, int64_t R, int N)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= R) return;
    const int64_t a = action[r];
    uint8_t* m = mask + r * N;
    int any = 0;
    for (int n = lane; n < N; n += 64) {
        uint8_t v = m[n];
        if (n == a) { v = 0; m[n] = 0; }
        any |= v;
    }
    const bool remaining = __ballot(any != 0) != 0ull;
    if (lane == 0) {
        if (istep[r] == 0) first[r] = a;
        cur[r] = a;
        istep[r] += 1;
        done[r] = remaining ? 0 : 1;
    }
}

// STEP = 0: mask only (get_action_mask); STEP = 1: transition then mask
template <int STEP>
__device__ __forceinline__ void cvrp_step_mask_row(uint8_t* visited, float* used, const float* vcap,
                                                       const float* demand, int64_t* cur, uint8_t* mask, uint8_t* done, int N, int64_t rs, int64_t bi, int lane, int64_t act)
{
    const int M = N + 1;
    const float* dem = demand + bi * N;
    uint8_t* vis = visited + rs * M;
    float u = used[rs];
    int64_t c = cur[rs];
    int64_t a = -1;
    if (STEP) {
        a = act;
        int64_t di = a - 1;
        di = di < 0 ? 0 : (di > N - 1 ? N - 1 : di);
        u = (u + dem[di]) * (a != 0 ? 1.0f : 0.0f);
        c = a;
    }
    const float lim = vcap[rs] + 1e-5f;
    int any_free = 0;
    int all_vis = 1;
    for (int j = lane; j < N; j += 64) {
        int v = vis[j + 1] != 0;
        if (STEP && j + 1 == a) { v = 1; vis[j + 1] = 1; }
        const float load = dem[j] + u;
        const int blocked = v | (load > lim);
        mask[rs * M + 1 + j] = !blocked;
        any_free |= !blocked;
        all_vis &= v;
    }
    const bool any = __ballot(any_free != 0) != 0ull;
    const bool allc = __ballot(all_vis == 0) == 0ull;
    if (lane == 0) {
        int v0 = vis[0] != 0;
        if (STEP && a == 0) { v0 = 1; vis[0] = 1; }
        mask[rs * M] = !((c == 0) && any);
        if (STEP) {
            used[rs] = u;
            cur[rs] = c;
            done[rs] = (allc && v0) ? 1 : 0;
        }
    }
}

template <int STEP>
__global__ __launch_bounds__(EB) void k_cvrp_step_mask(uint8_t* visited, float* used, const float* vcap,
                                                       const float* demand, int64_t* cur, const int64_t* action,
                                                       uint8_t* mask, uint8_t* done, int64_t R, int64_t B, int N)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= R) return;
    cvrp_step_mask_row<STEP>(visited, used, vcap, demand, cur, mask, done, N, r, r % B, lane, STEP ? action[r] : (int64_t)-1);
}

// SDVRP: rem = demand_with_depot.  STEP = 0: mask only; STEP = 1: deliver min(rem[action], free capacity), then mask.
template <int STEP>
__global__ __launch_bounds__(EB) void k_sdvrp_step_mask(float* rem, float* used, const float* vcap, int64_t* cur,
                                                        const int64_t* action, uint8_t* mask, uint8_t* done, int64_t R,
                                                        int M)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= R) return;
    float* rr = rem + r * M;
    float u = used[r];
    const float cap = vcap[r];
    int64_t c = cur[r];
    int64_t a = -1;
    float left = 0.0f;
    if (STEP) {
        a = action[r];
        a = a < 0 ? 0 : (a > M - 1 ? M - 1 : a);     // an out-of-range action must not become an out-of-bounds access
        const float sel = rr[a];
        const float free_cap = cap - u;
        const float delivered = sel < free_cap ? sel : free_cap;
        u = (u + delivered) * (a != 0 ? 1.0f : 0.0f);
        left = sel + (-delivered);
        c = a;
    }
    const bool full = u >= cap;
    int any_free = 0, any_rem = 0;
    for (int n = lane; n < M; n += 64) {
        float rv = rr[n];
        if (STEP && n == a) { rv = left; rr[n] = left; }
        any_rem |= rv > 0.0f;
        if (n >= 1) {
            const int blocked = (rv == 0.0f) | full;
            mask[r * M + n] = !blocked;
            any_free |= !blocked;
        }
    }
    const bool anyf = __ballot(any_free != 0) != 0ull;
    const bool anyr = __ballot(any_rem != 0) != 0ull;
    if (lane == 0) {
        mask[r * M] = !((c == 0) && anyf);
        if (STEP) {
            used[r] = u;
            cur[r] = c;
            done[r] = anyr ? 0 : 1;
        }
    }
}

// SDVRPEnv.check_solution_validity (sdvrp/env.py:148-171): replay the deliveries over the vector
// (-vehicle_capacity, demand...).  One wavefront per row, the vector in LDS; the replay is sequential by definition.
__global__ __launch_bounds__(EB) void k_check_sdvrp(const int64_t* actions, const float* demand, const float* vcap,
                                                    int64_t R, int64_t B, int N, int T, int32_t* bad)
{
    extern __shared__ float dem_all[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wv;
    if (r >= R) return;
    const int M = N + 1;
    float* d = dem_all + (size_t)wv * M;
    const float cap = vcap[r];
    const float* dem = demand + (r % B) * N;
    for (int n = lane; n < M; n += 64) d[n] = n == 0 ? -cap : dem[n - 1];
    __builtin_amdgcn_wave_barrier();
    const int64_t* act = actions + r * T;
    float usedc = 0.0f;
    int64_t prev = -1;
    int twice = 0, range = 0;
    for (int t = 0; t < T; ++t) {
        const int64_t a = act[t];
        if (a < 0 || a > N) { range = 1; break; }
        if (t > 0 && prev == 0 && a == 0) {           // "Cannot visit depot twice if any nonzero demand"
            int nz = 0;
            for (int n = lane; n < M; n += 64) nz |= d[n] != 0.0f;
            if (__ballot(nz != 0) != 0ull) twice = 1;
        }
        const float da = d[a];
        const float free_cap = cap - usedc;
        const float dl = da < free_cap ? da : free_cap;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) d[a] = da - dl;
        __builtin_amdgcn_wave_barrier();
        usedc = usedc + dl;
        if (a == 0) usedc = 0.0f;
        prev = a;
    }
    int nz = 0;
    for (int n = lane; n < M; n += 64) nz |= d[n] != 0.0f;
    const bool left = __ballot(nz != 0) != 0ull;
    if (lane != 0) return;
    if (range || left) atomicAdd(&bad[0], 1);
    if (twice) atomicAdd(&bad[1], 1);
}

// torch's norm(p=2, dim=-1) of a 2-vector on the CPU, bit for bit (OP): sqrtf(fmaf(dy, dy, dx * dx))
__device__ __forceinline__ float dist2(float ax, float ay, float bx, float by)
{
    const float dx = ax - bx, dy = ay - by;
    return __builtin_sqrtf(fma_(dy, dy, dx * dx));
}

// OP: one wavefront per row.  STEP = 0: mask only; STEP = 1: move (tour length += leg), mark visited, then mask.
template <int STEP>
__device__ __forceinline__ void op_step_mask_row(uint8_t* visited, float* tour_len, float* prize_tot, const float* prize,
                                                     const float* locs, const float* maxlen, int64_t* cur, int64_t* istep,
                                                     uint8_t* mask, uint8_t* done, int M, int64_t rs, int64_t bi, int lane, int64_t act)
{
    const float* L = locs + bi * (int64_t)M * 2;
    const float* ml = maxlen + bi * (int64_t)M;
    uint8_t* vis = visited + rs * M;
    float tl = tour_len[rs];
    int64_t c = cur[rs];
    c = c < 0 ? 0 : (c > M - 1 ? M - 1 : c);
    int64_t a = -1;
    int v0 = vis[0] != 0;
    if (STEP) {
        a = act;
        a = a < 0 ? 0 : (a > M - 1 ? M - 1 : a);     // an out-of-range action must not become an out-of-bounds access
        tl = tl + dist2(L[2 * a], L[2 * a + 1], L[2 * c], L[2 * c + 1]);
        c = a;
        if (a == 0) v0 = 1;
    }
    const float cx = L[2 * c], cy = L[2 * c + 1];
    for (int n = 1 + lane; n < M; n += 64) {
        int v = vis[n] != 0;
        if (STEP && n == a) { v = 1; vis[n] = 1; }
        const int exceeds = (tl + dist2(L[2 * n], L[2 * n + 1], cx, cy)) > ml[n];
        mask[rs * M + n] = !(v | v0 | exceeds);
    }
    if (lane == 0) {
        mask[rs * M] = 1;                             // the depot can always be visited
        if (STEP) {
            if (a == 0) vis[0] = 1;
            tour_len[rs] = tl;
            if (prize_tot) prize_tot[rs] = prize_tot[rs] + prize[bi * M + a];
            const int64_t i = istep[rs];
            done[rs] = (a == 0 && i > 0) ? 1 : 0;
            cur[rs] = a;
            istep[rs] = i + 1;
        }
    }
}

template <int STEP>
__global__ __launch_bounds__(EB) void k_op_step_mask(uint8_t* visited, float* tour_len, float* prize_tot, const float* prize,
                                                     const float* locs, const float* maxlen, int64_t* cur, int64_t* istep,
                                                     const int64_t* action, uint8_t* mask, uint8_t* done, int64_t R,
                                                     int64_t B, int M)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= R) return;
    op_step_mask_row<STEP>(visited, tour_len, prize_tot, prize, locs, maxlen, cur, istep, mask, done, M, r, r % B, lane, STEP ? action[r] : (int64_t)-1);
}

// CVRPTW: CVRP transition and mask plus the clock (cvrptw/env.py:103-138).  One wavefront per row.
template <int STEP>
__device__ __forceinline__ void cvrptw_step_mask_row(uint8_t* visited, float* used, const float* vcap,
                                                         const float* demand, int64_t* cur, float* time, const float* locs,
                                                         const float* tw, const float* dur, uint8_t* mask, uint8_t* done, int N, int64_t rs, int64_t bi, int lane, int64_t act)
{
    const int M = N + 1;
    const float* dem = demand + bi * N;
    const float* L = locs + bi * (int64_t)M * 2;
    const float* W = tw + bi * (int64_t)M * 2;
    uint8_t* vis = visited + rs * M;
    float u = used[rs], now = time[rs];
    int64_t c = cur[rs];
    c = c < 0 ? 0 : (c > M - 1 ? M - 1 : c);
    int64_t a = -1;
    if (STEP) {
        a = act;
        a = a < 0 ? 0 : (a > M - 1 ? M - 1 : a);     // an out-of-range action must not become an out-of-bounds access
        const float arrive = now + dist2(L[2 * c], L[2 * c + 1], L[2 * a], L[2 * a + 1]);
        const float start = arrive > W[2 * a] ? arrive : W[2 * a];
        now = (a != 0 ? 1.0f : 0.0f) * (start + dur[bi * (int64_t)M + a]);
        int64_t di = a - 1;
        di = di < 0 ? 0 : (di > N - 1 ? N - 1 : di);
        u = (u + dem[di]) * (a != 0 ? 1.0f : 0.0f);
        c = a;
    }
    const float cx = L[2 * c], cy = L[2 * c + 1];
    const float lim = vcap[rs] + 1e-5f;
    int any_free = 0, all_vis = 1;
    for (int j = lane; j < N; j += 64) {
        int v = vis[j + 1] != 0;
        if (STEP && j + 1 == a) { v = 1; vis[j + 1] = 1; }
        const int blocked = v | ((dem[j] + u) > lim);
        const int in_time = (now + dist2(cx, cy, L[2 * (j + 1)], L[2 * (j + 1) + 1])) <= W[2 * (j + 1) + 1];
        mask[rs * M + 1 + j] = (!blocked) & in_time;
        any_free |= !blocked;                        // the depot rule looks at the CVRP mask only
        all_vis &= v;
    }
    const bool any = __ballot(any_free != 0) != 0ull;
    const bool allc = __ballot(all_vis == 0) == 0ull;
    if (lane == 0) {
        int v0 = vis[0] != 0;
        if (STEP && a == 0) { v0 = 1; vis[0] = 1; }
        const int in_time0 = (now + dist2(cx, cy, L[0], L[1])) <= W[1];
        mask[rs * M] = (!((c == 0) && any)) & in_time0;
        if (STEP) {
            used[rs] = u;
            time[rs] = now;
            cur[rs] = c;
            done[rs] = (allc && v0) ? 1 : 0;
        }
    }
}

template <int STEP>
__global__ __launch_bounds__(EB) void k_cvrptw_step_mask(uint8_t* visited, float* used, const float* vcap,
                                                         const float* demand, int64_t* cur, float* time, const float* locs,
                                                         const float* tw, const float* dur, const int64_t* action,
                                                         uint8_t* mask, uint8_t* done, int64_t R, int64_t B, int N)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= R) return;
    cvrptw_step_mask_row<STEP>(visited, used, vcap, demand, cur, time, locs, tw, dur, mask, done, N, r, r % B, lane, STEP ? action[r] : (int64_t)-1);
}

// the time-window replay of CVRPTWEnv.check_solution_validity (cvrptw/env.py:203-227): sequential, one thread per row
__global__ void k_check_cvrptw_time(const int64_t* actions, const float* locs, const float* tw, const float* dur, int64_t R,
                                    int64_t B, int M, int T, int32_t* bad)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float* L = locs + (r % B) * (int64_t)M * 2;
    const float* W = tw + (r % B) * (int64_t)M * 2;
    float curr = 0.0f;
    int64_t node = 0;
    int late = 0;
    for (int t = 0; t < T; ++t) {
        const int64_t nx = actions[r * T + t];
        if (nx < 0 || nx >= M) { late = 1; break; }
        int ct = (int)(curr + dist2(L[2 * node], L[2 * node + 1], L[2 * nx], L[2 * nx + 1]));
        const int ws = (int)W[2 * nx];
        ct = ws > ct ? ws : ct;
        if ((float)ct > W[2 * nx + 1]) late = 1;
        curr = (float)ct + dur[(r % B) * (int64_t)M + nx];
        node = nx;
        if (nx == 0) curr = 0.0f;
    }
    if (late) atomicAdd(&bad[0], 1);
}

// OPEnv._get_reward: lane tree over the steps of prize[a_t]
__global__ __launch_bounds__(EB) void k_op_reward(const float* prize, const int64_t* actions, float* reward, int64_t R,
                                                  int64_t B, int M, int T)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* pz = prize + (r % B) * (int64_t)M;
    const int64_t* act = actions + r * T;
    float total = 0.0f;
    for (int b0 = 0; b0 < T; b0 += 64) {
        const int t = b0 + lane;
        int64_t a = t < T ? act[t] : 0;
        a = a < 0 ? 0 : (a >= M ? M - 1 : a);
        const float s = wave_tree_sum(t < T ? pz[a] : 0.0f);
        total = (b0 == 0) ? s : total + s;
    }
    if (lane == 0) reward[r] = total;
}

// OPEnv.check_solution_validity (op/env.py:179-212): one wavefront per row
__global__ __launch_bounds__(EB) void k_check_op(const int64_t* actions, const float* locs, const float* maxlen, int64_t R,
                                                 int64_t B, int M, int T, int32_t* bad)
{
    __shared__ uint32_t seen_all[ROWS_PER_BLOCK][128];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wv;
    if (r >= R) return;
    uint32_t* seen = seen_all[wv];
    for (int i = lane; i < 128; i += 64) seen[i] = 0;
    __builtin_amdgcn_wave_barrier();
    const int64_t* act = actions + r * T;
    const float* L = locs + (r % B) * (int64_t)M * 2;
    int bad_lane = 0;
    float length = 0.0f;
    for (int b0 = 0; b0 < T; b0 += 64) {
        const int t = b0 + lane;
        float d = 0.0f;
        if (t < T) {
            int64_t a0 = act[t], a1 = act[(t + 1 == T) ? 0 : t + 1];
            if (a0 < 0 || a0 >= M) { bad_lane = 1; a0 = 0; }
            else if (a0 != 0) {
                const uint32_t bit = 1u << (a0 & 31);
                if (atomicOr(&seen[a0 >> 5], bit) & bit) bad_lane = 1;
            }
            a1 = a1 < 0 ? 0 : (a1 >= M ? M - 1 : a1);
            d = dist2(L[2 * a1], L[2 * a1 + 1], L[2 * a0], L[2 * a0 + 1]);
        }
        const float s = wave_tree_sum(d);
        length = (b0 == 0) ? s : length + s;
    }
    const bool invalid = __ballot(bad_lane != 0) != 0ull;
    int ex = 0;
    const float* ml = maxlen + (r % B) * (int64_t)M;
    for (int n = lane; n < M; n += 64) {
        const float lim = ((ml[n] + dist2(L[0], L[1], L[2 * n], L[2 * n + 1])) + 1e-6f) + 1e-5f;
        ex |= !(length <= lim);
    }
    const bool over = __ballot(ex != 0) != 0ull;
    if (lane != 0) return;
    if (invalid) { atomicAdd(&bad[0], 1); return; }
    if (over) atomicAdd(&bad[1], 1);
}

// PCTSP: one wavefront per row.  STEP = 0: mask only; STEP = 1: collect prize (and penalty), mark visited, then mask.
// prize / penalty [B][M] with a zero depot slot; pen_tot / penalty may be null together.
template <int STEP>
__device__ __forceinline__ void pctsp_step_mask_row(uint8_t* visited, float* prize_tot, float* pen_tot,
                                                        const float* prize, const float* penalty, int64_t* cur,
                                                        int64_t* istep, uint8_t* mask,
                                                        uint8_t* done, int M, int64_t rs, int64_t bi, int lane, int64_t act)
{
    uint8_t* vis = visited + rs * M;
    float pt = prize_tot[rs];
    int64_t a = -1;
    int v0 = vis[0] != 0;
    if (STEP) {
        a = act;
        a = a < 0 ? 0 : (a > M - 1 ? M - 1 : a);     // an out-of-range action must not become an out-of-bounds access
        pt = pt + prize[bi * M + a];
        if (a == 0) v0 = 1;
    }
    int unvisited = 0;
    for (int n = 1 + lane; n < M; n += 64) {
        int v = vis[n] != 0;
        if (STEP && n == a) { v = 1; vis[n] = 1; }
        mask[rs * M + n] = !(v | v0);
        unvisited |= !v;
    }
    const bool unv = __ballot(unvisited != 0) != 0ull;
    if (lane == 0) {
        mask[rs * M] = !((pt < 1.0f) && unv);
        if (STEP) {
            if (a == 0) vis[0] = 1;
            prize_tot[rs] = pt;
            if (pen_tot) pen_tot[rs] = pen_tot[rs] + penalty[bi * M + a];
            const int64_t i = istep[rs];
            done[rs] = (i > 0 && a == 0) ? 1 : 0;
            cur[rs] = a;
            istep[rs] = i + 1;
        }
    }
}

template <int STEP>
__global__ __launch_bounds__(EB) void k_pctsp_step_mask(uint8_t* visited, float* prize_tot, float* pen_tot,
                                                        const float* prize, const float* penalty, int64_t* cur,
                                                        int64_t* istep, const int64_t* action, uint8_t* mask,
                                                        uint8_t* done, int64_t R, int64_t B, int M)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= R) return;
    pctsp_step_mask_row<STEP>(visited, prize_tot, pen_tot, prize, penalty, cur, istep, mask, done, M, r, r % B, lane, STEP ? action[r] : (int64_t)-1);
}

// PCTSPEnv.check_solution_validity (pctsp/env.py:189-205): one wavefront per row.  bad[0] += rows with a customer
// visited twice (or an id out of range), bad[1] += rows that neither collect a prize >= 1 - 1e-5 nor visit everyone.
__global__ __launch_bounds__(EB) void k_check_pctsp(const int64_t* actions, const float* prize, int64_t R, int64_t B, int M,
                                                    int T, int32_t* bad)
{
    __shared__ uint32_t seen_all[ROWS_PER_BLOCK][128];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wv;
    if (r >= R) return;
    uint32_t* seen = seen_all[wv];
    for (int i = lane; i < 128; i += 64) seen[i] = 0;
    __builtin_amdgcn_wave_barrier();
    const int64_t* act = actions + r * T;
    int bad_lane = 0, cnt = 0;
    for (int t = lane; t < T; t += 64) {
        const int64_t a = act[t];
        if (a < 0 || a >= M) { bad_lane = 1; continue; }
        if (a == 0) continue;
        const uint32_t bit = 1u << (a & 31);
        if (atomicOr(&seen[a >> 5], bit) & bit) bad_lane = 1;
        else ++cnt;
    }
    const bool invalid = __ballot(bad_lane != 0) != 0ull;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane != 0) return;
    if (invalid) { atomicAdd(&bad[0], 1); return; }
    const float* pz = prize + (r % B) * M;
    float p = 0.0f;
    for (int t = 0; t < T; ++t) p = p + pz[act[t]];
    if (!((p >= (float)(1.0 - 1e-5)) || cnt == M - 1)) atomicAdd(&bad[1], 1);
}

// penalty != null (PCTSP, with_depot): reward = saved penalties - (length + all penalties), pctsp/env.py:165-187;
// the penalty sums go through the same lane tree as the legs.
__global__ __launch_bounds__(EB) void k_tour_length(const float* locs, const int64_t* actions, float* reward, int64_t R,
                                                    int64_t B, int M, int T, int with_depot, const float* penalty)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* L = locs + (r % B) * (int64_t)M * 2;
    const int64_t* act = actions + r * T;
    const int P = T + (with_depot ? 1 : 0);
    float total = 0.0f;
    for (int b0 = 0; b0 < P; b0 += 64) {
        const int t = b0 + lane;
        float d = 0.0f;
        if (t < P) {
            int64_t a0, a1;
            if (with_depot) {
                a0 = (t == 0) ? 0 : act[t - 1];
                a1 = (t + 1 == P) ? 0 : act[t];
            } else {
                a0 = act[t];
                a1 = act[(t + 1 == P) ? 0 : t + 1];
            }
            // caller-supplied tours: keep the gather inside the instance whatever the indices are
            a0 = a0 < 0 ? 0 : (a0 >= M ? M - 1 : a0);
            a1 = a1 < 0 ? 0 : (a1 >= M ? M - 1 : a1);
            const float2 p0 = *reinterpret_cast<const float2*>(L + 2 * a0);
            const float2 p1 = *reinterpret_cast<const float2*>(L + 2 * a1);
            const float dx = p1.x - p0.x, dy = p1.y - p0.y;
            d = __builtin_sqrtf(fma_(dy, dy, dx * dx));
        }
        const float s = wave_tree_sum(d);
        total = (b0 == 0) ? s : total + s;
    }
    if (penalty) {
        const float* pen = penalty + (r % B) * (int64_t)M;
        float saved = 0.0f, all = 0.0f;
        for (int b0 = 0; b0 < T; b0 += 64) {
            const int t = b0 + lane;
            int64_t a = t < T ? act[t] : 0;
            a = a < 0 ? 0 : (a >= M ? M - 1 : a);
            const float s = wave_tree_sum(t < T ? pen[a] : 0.0f);
            saved = (b0 == 0) ? s : saved + s;
        }
        for (int b0 = 0; b0 < M - 1; b0 += 64) {
            const int n = b0 + lane;
            const float s = wave_tree_sum(n < M - 1 ? pen[1 + n] : 0.0f);
            all = (b0 == 0) ? s : all + s;
        }
        if (lane == 0) reward[r] = saved - (total + all);
        return;
    }
    if (lane == 0) reward[r] = -total;
}

// 64 rows per 256-thread block; tiles of 64 steps are staged through LDS so that the global reads are
// row-contiguous while thread r still adds row r strictly in step order.
__global__ __launch_bounds__(256) void k_sum_logp(const float* logp, int64_t ld, float* out, int64_t R, int T)
{
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    float s = 0.0f;
    for (int t0 = 0; t0 < T; t0 += 64) {
        for (int i = tid; i < 64 * 64; i += 256) {
            const int rr = i >> 6, tt = i & 63;
            const int64_t r = r0 + rr;
            tile[rr][tt] = (r < R && t0 + tt < T) ? logp[r * ld + t0 + tt] : 0.0f;
        }
        __syncthreads();
        if (tid < 64) {
            const int tn = min(64, T - t0);
            for (int t = 0; t < tn; ++t) s = s + tile[tid][t];
        }
        __syncthreads();
    }
    if (tid < 64 && r0 + tid < R) out[r0 + tid] = s;
}

// BeamSearch._make_beam_step (rl4co/utils/decoding.py:573-608): per instance, the beam_width best of the
// beam_width * M candidates  logprobs[w*B + b][n] + parent[w*B + b]  (rows in "(w b)" order), descending; ties go to
// the lower flat index w*M + n.  One workgroup per instance: candidates in LDS, beam_width rounds of a block argmax.
// Output row k*B + b: node, parent beam, cumulative log-prob and the step log-prob of the chosen (beam, node).
__global__ __launch_bounds__(256) void k_beam_topk(const float* logprobs, const float* parent, int64_t B, int BW, int M,
                                                   int64_t* node, int32_t* beam, float* cum, float* step_lp)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* val = reinterpret_cast<float*>(smem);                 // [BW * M]
    float* redv = val + BW * M;                                   // [4]
    int* redi = reinterpret_cast<int*>(redv + 4);                 // [4]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t b = blockIdx.x;
    const int C = BW * M;
    for (int c = tid; c < C; c += 256) {
        const int w = c / M, n = c - w * M;
        const int64_t r = (int64_t)w * B + b;
        val[c] = logprobs[r * M + n] + parent[r];
    }
    __syncthreads();
    for (int k = 0; k < BW; ++k) {
        float best = -INFINITY;
        int besti = 0x7fffffff;
        for (int c = tid; c < C; c += 256) {                      // ascending c per thread: the first maximum is kept
            const float v = val[c];
            if (v == v && (besti == 0x7fffffff || v > best)) { best = v; besti = c; }      // NaN marks "taken"
        }
        wave_argmax(best, besti);
        if (lane == 0) { redv[wv] = best; redi[wv] = besti; }
        __syncthreads();
        if (tid == 0) {
            float bv = redv[0];
            int bi = redi[0];
            for (int i = 1; i < 4; ++i)
                if (redi[i] != 0x7fffffff && (bi == 0x7fffffff || redv[i] > bv || (redv[i] == bv && redi[i] < bi))) { bv = redv[i]; bi = redi[i]; }
            if (bi == 0x7fffffff) bi = 0;                          // unreachable for BW <= BW * M
            const int w = bi / M, n = bi - w * M;
            const int64_t r = (int64_t)k * B + b;
            node[r] = n;
            beam[r] = w;
            cum[r] = bv;
            step_lp[r] = logprobs[((int64_t)w * B + b) * M + n];
            val[bi] = __builtin_nanf("");                          // taken
        }
        __syncthreads();
    }
}

// One wavefront per row; "seen" bitmap in LDS (M <= 4096).
__global__ __launch_bounds__(EB) void k_check_solution(int env, const int64_t* actions, const float* demand,
                                                       const float* vcap, int64_t R, int64_t B, int N, int T,
                                                       int32_t* bad)
{
    __shared__ uint32_t seen_all[ROWS_PER_BLOCK][128];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wv;
    if (r >= R) return;
    uint32_t* seen = seen_all[wv];
    for (int i = lane; i < 128; i += 64) seen[i] = 0;
    __builtin_amdgcn_wave_barrier();
    const int64_t* act = actions + r * T;
    const int top = (env == EAMRL_ENV_TSP) ? N - 1 : N;  // highest legal node id
    int bad_lane = 0;
    for (int t = lane; t < T; t += 64) {
        const int64_t a = act[t];
        if (a < 0 || a > top) { bad_lane = 1; continue; }
        if (env == EAMRL_ENV_CVRP && a == 0) continue;
        const uint32_t bit = 1u << (a & 31);
        const uint32_t old = atomicOr(&seen[a >> 5], bit);
        if (old & bit) bad_lane = 1;  // visited twice
    }
    __builtin_amdgcn_wave_barrier();
    const int lo = (env == EAMRL_ENV_TSP) ? 0 : 1;
    for (int n = lo + lane; n <= top; n += 64)
        if (!((seen[n >> 5] >> (n & 31)) & 1u)) bad_lane = 1;  // never visited
    const bool invalid = __ballot(bad_lane != 0) != 0ull;
    if (lane != 0) return;
    if (invalid) { atomicAdd(&bad[0], 1); return; }
    if (env == EAMRL_ENV_CVRP) {
        // running load, depot resets it (cvrp/env.py:172-185)
        const float cap = vcap[r], lim = cap + 1e-5f;
        const float* dem = demand + (r % B) * N;
        float usedc = 0.0f;
        int over = 0;
        for (int t = 0; t < T; ++t) {
            const int64_t a = act[t];
            usedc = usedc + ((a == 0) ? -cap : dem[a - 1]);
            if (usedc < 0.0f) usedc = 0.0f;
            if (usedc > lim) over = 1;
        }
        if (over) atomicAdd(&bad[1], 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The env state BEFORE every decode step of given action rows, in one launch (depot envs; TSP is closed-form in the
// actions: k_tsp_mask_bits): a wavefront keeps its row's state in LDS and alternates "record" (mask as a 128-bit set,
// current node, state scalars) and the env's own transition -- the very row functions of the step kernels above.
// Replaces T x (pack bits, copy current node, state scalar, step) launches of the re-evaluation's set-up.
// ---------------------------------------------------------------------------------------------------------------------
struct ReplayArgs {
    const uint8_t* mask; const uint8_t* visited; const float* used; const float* vcap; const int64_t* cur; const int64_t* istep;
    const float* time; const float* demand; const float* locs; const float* tw; const float* dur;
    const int64_t* actions; uint32_t* bits; int32_t* idxA; float* sc;
    int64_t R, B; int M, T;
};

// SLOTS = 128: graphs up to 112 nodes, masks as one 128-bit word per (row, step).  SLOTS = 1024: larger graphs, masks in the layout
// of the key-chunked re-evaluation kernels -- bits [R][T][nkc][4], nkc = ceil(M / 112), bit i of chunk c = node 112 c + i.
template <int ENV, int SLOTS = 128>
__global__ __launch_bounds__(EB) void k_replay_states(ReplayArgs a)
{
    __shared__ uint8_t s_mask[ROWS_PER_BLOCK][SLOTS], s_vis[ROWS_PER_BLOCK][SLOTS], s_done[ROWS_PER_BLOCK][8];
    __shared__ float s_used[ROWS_PER_BLOCK], s_time[ROWS_PER_BLOCK];
    __shared__ int64_t s_cur[ROWS_PER_BLOCK], s_istep[ROWS_PER_BLOCK];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wv;
    if (r >= a.R) return;
    const int M = a.M, T = a.T;
    const int64_t bi = r % a.B;
    for (int n = lane; n < SLOTS; n += 64) {
        s_mask[wv][n] = n < M ? a.mask[r * M + n] : 0;
        s_vis[wv][n] = n < M ? a.visited[r * M + n] : 0;
    }
    const int nkc = (M + 111) / 112;
    if (lane == 0) {
        s_used[wv] = a.used[r];
        s_time[wv] = a.time ? a.time[r] : 0.0f;
        s_cur[wv] = a.cur[r];
        s_istep[wv] = a.istep ? a.istep[r] : 0;
        s_done[wv][0] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    const float vc = a.vcap[r];
    for (int t = 0; t < T; ++t) {
        // ---- record ------------------------------------------------------------------------------------------------------
        const unsigned long long b0 = __ballot(s_mask[wv][lane] != 0), b1 = __ballot(s_mask[wv][64 + lane] != 0);
        if (SLOTS > 128) {          // one 32-bit word of one chunk per lane
            for (int wi = lane; wi < 4 * nkc; wi += 64) {
                const int c = wi >> 2, w = wi & 3;
                uint32_t v = 0;
                for (int i = 0; i < 32; ++i) {
                    const int li = 32 * w + i, n = 112 * c + li;
                    if (li < 112 && n < M && s_mask[wv][n]) v |= 1u << i;
                }
                a.bits[((r * T + t) * nkc + c) * 4 + w] = v;
            }
        }
        if (lane == 0) {
            const int64_t q = r * T + t;
            if (SLOTS == 128)
                *reinterpret_cast<uint4*>(a.bits + q * 4) =
                    make_uint4((uint32_t)b0, (uint32_t)(b0 >> 32), (uint32_t)b1, (uint32_t)(b1 >> 32));
            a.idxA[q] = (int32_t)s_cur[wv];
            float free_ = vc - s_used[wv];            // free capacity / prize still to collect / length still allowed
            if (ENV == EAMRL_ENV_PCTSP) free_ = free_ < 0.0f ? 0.0f : free_;
            a.sc[q] = free_;
            if (ENV == EAMRL_ENV_CVRPTW) a.sc[a.R * (int64_t)T + q] = s_time[wv];
        }
        __builtin_amdgcn_wave_barrier();
        // ---- the env transition (+ next mask) ---------------------------------------------------------------------------
        const int64_t act = a.actions[r * T + t];
        if (ENV == EAMRL_ENV_CVRP)
            cvrp_step_mask_row<1>(s_vis[wv], &s_used[wv], &vc, a.demand, &s_cur[wv], s_mask[wv], s_done[wv], M - 1, 0, bi, lane, act);
        else if (ENV == EAMRL_ENV_CVRPTW)
            cvrptw_step_mask_row<1>(s_vis[wv], &s_used[wv], &vc, a.demand, &s_cur[wv], &s_time[wv], a.locs, a.tw, a.dur, s_mask[wv],
                                    s_done[wv], M - 1, 0, bi, lane, act);
        else if (ENV == EAMRL_ENV_PCTSP)
            pctsp_step_mask_row<1>(s_vis[wv], &s_used[wv], nullptr, a.demand, nullptr, &s_cur[wv], &s_istep[wv], s_mask[wv], s_done[wv],
                                   M, 0, bi, lane, act);
        else
            op_step_mask_row<1>(s_vis[wv], &s_used[wv], nullptr, nullptr, a.locs, a.demand, &s_cur[wv], &s_istep[wv], s_mask[wv],
                                s_done[wv], M, 0, bi, lane, act);
        __builtin_amdgcn_wave_barrier();
    }
}

// SDVRP: the same recording for SDVRPEnv (sdvrp/env.py:58-92,137-146) -- the state is the remaining-demand row itself, kept in
// registers (nodes lane and lane + 64); additionally every step's row goes to rem_out [R][T][128] (zero padded), which is what the
// re-evaluation's dynamic embedding multiplies.  Arithmetic of k_sdvrp_step_mask.
__global__ __launch_bounds__(EB) void k_replay_sdvrp(const float* __restrict__ rem, const float* __restrict__ used,
                                                     const float* __restrict__ vcap, const int64_t* __restrict__ cur,
                                                     const int64_t* __restrict__ actions, uint32_t* __restrict__ bits,
                                                     int32_t* __restrict__ idxA, float* __restrict__ sc, float* __restrict__ rem_out,
                                                     int64_t R, int M, int T)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= R) return;
    const int n0 = lane, n1 = lane + 64;
    float r0 = n0 < M ? rem[r * M + n0] : 0.0f, r1 = n1 < M ? rem[r * M + n1] : 0.0f;
    float u = used[r];
    const float cap = vcap[r];
    int c = (int)cur[r];
    for (int t = 0; t < T; ++t) {
        // ---- record: mask (get_action_mask), current node, free capacity, the remaining demands -------------------------------
        const bool full = u >= cap;
        const bool ok0 = n0 >= 1 && n0 < M && !((r0 == 0.0f) | full), ok1 = n1 < M && !((r1 == 0.0f) | full);
        unsigned long long b0 = __ballot(ok0), b1 = __ballot(ok1);
        const bool any_free = (b0 | b1) != 0ull;
        if (!((c == 0) && any_free)) b0 |= 1ull;                         // the depot
        const int64_t q = r * T + t;
        if (lane == 0) {
            *reinterpret_cast<uint4*>(bits + q * 4) = make_uint4((uint32_t)b0, (uint32_t)(b0 >> 32), (uint32_t)b1, (uint32_t)(b1 >> 32));
            idxA[q] = c;
            sc[q] = cap - u;
        }
        rem_out[q * 128 + n0] = r0;
        rem_out[q * 128 + n1] = r1;
        // ---- SDVRPEnv._step: deliver min(remaining demand, free capacity) ---------------------------------------------------------
        int a = __builtin_amdgcn_readfirstlane((int)actions[q]);
        a = a < 0 ? 0 : (a > M - 1 ? M - 1 : a);
        const float sel = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a < 64 ? r0 : r1), a & 63));
        const float free_cap = cap - u;
        const float delivered = sel < free_cap ? sel : free_cap;
        u = (u + delivered) * (a != 0 ? 1.0f : 0.0f);
        const float left = sel + (-delivered);
        if (n0 == a) r0 = left;
        if (n1 == a) r1 = left;
        c = a;
    }
}

// Graphs above 112 nodes (M <= 1024): the row in 16 registers per lane (node lane + 64 k), masks and remaining demands in the
// chunked layout of the key-chunked re-evaluation -- bits [R][T][nkc][4], rem_out [R][T][nkc][128] (ZERO-FILLED by the caller:
// only existing nodes are written), nkc = ceil(M / 112), node n at chunk n / 112, slot n % 112.
__global__ __launch_bounds__(EB) void k_replay_sdvrp_big(const float* __restrict__ rem, const float* __restrict__ used,
                                                         const float* __restrict__ vcap, const int64_t* __restrict__ cur,
                                                         const int64_t* __restrict__ actions, uint32_t* __restrict__ bits,
                                                         int32_t* __restrict__ idxA, float* __restrict__ sc, float* __restrict__ rem_out,
                                                         int64_t R, int M, int T)
{
    __shared__ uint8_t s_ok[ROWS_PER_BLOCK][1024];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wv;
    if (r >= R) return;
    const int nkc = (M + 111) / 112;
    float rr[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) rr[k] = (lane + 64 * k) < M ? rem[r * M + lane + 64 * k] : 0.0f;
    float u = used[r];
    const float cap = vcap[r];
    int c = (int)cur[r];
    for (int t = 0; t < T; ++t) {
        const int64_t q = r * T + t;
        const bool full = u >= cap;
        bool any_free = false;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int n = lane + 64 * k;
            const bool ok = n >= 1 && n < M && !((rr[k] == 0.0f) | full);
            s_ok[wv][n] = ok;
            any_free |= __ballot(ok) != 0ull;
            if (n < M) rem_out[(q * nkc + n / 112) * 128 + n % 112] = rr[k];
        }
        if (lane == 0) {
            s_ok[wv][0] = !((c == 0) && any_free);                         // the depot
            idxA[q] = c;
            sc[q] = cap - u;
        }
        __builtin_amdgcn_wave_barrier();
        for (int wi = lane; wi < 4 * nkc; wi += 64) {                      // one 32-bit word of one chunk per lane
            const int ch = wi >> 2, w = wi & 3;
            uint32_t v = 0;
            for (int i = 0; i < 32; ++i) {
                const int li = 32 * w + i, n = 112 * ch + li;
                if (li < 112 && n < M && s_ok[wv][n]) v |= 1u << i;
            }
            bits[(q * nkc + ch) * 4 + w] = v;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- SDVRPEnv._step ------------------------------------------------------------------------------------------------------
        int a = __builtin_amdgcn_readfirstlane((int)actions[q]);
        a = a < 0 ? 0 : (a > M - 1 ? M - 1 : a);
        float selv = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if ((a >> 6) == k) selv = rr[k];
        const float sel = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, selv), a & 63));
        const float free_cap = cap - u;
        const float delivered = sel < free_cap ? sel : free_cap;
        u = (u + delivered) * (a != 0 ? 1.0f : 0.0f);
        const float left = sel + (-delivered);
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (lane + 64 * k == a) rr[k] = left;
        c = a;
    }
}

int launch_replay_sdvrp(const float* rem, const float* used, const float* vcap, const int64_t* cur, const int64_t* actions,
                        uint32_t* bits, int32_t* idxA, float* sc, float* rem_out, int64_t R, int M, int T, hipStream_t st)
{
    if (M > 112) {
        hipLaunchKernelGGL(k_replay_sdvrp_big, dim3((unsigned)((R + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(EB), 0, st, rem, used,
                           vcap, cur, actions, bits, idxA, sc, rem_out, R, M, T);
        return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
    }
    hipLaunchKernelGGL(k_replay_sdvrp, dim3((unsigned)((R + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(EB), 0, st, rem, used, vcap, cur, actions, bits, idxA, sc, rem_out,
                       R, M, T);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------------
// Rollout epilogue in one launch (TSP / CVRP): reward (k_tour_length), validity (k_check_solution) and the log-likelihood
// (k_sum_logp) of a row by one wavefront, in exactly the orders of the three kernels it replaces.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EB) void k_rollout_finish(int env, const float* locs, const int64_t* actions, const float* logp,
                                                       int64_t ld, const float* demand, const float* vcap, float* reward,
                                                       float* ll, int32_t* bad, int64_t R, int64_t B, int M, int T)
{
    __shared__ uint32_t seen_all[ROWS_PER_BLOCK][128];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wv;
    if (r >= R) return;
    const int with_depot = env != EAMRL_ENV_TSP;
    const float* L = locs + (r % B) * (int64_t)M * 2;
    const int64_t* act = actions + r * T;
    // ---- tour length: lane tree over the legs --------------------------------------------------------------------------
    if (reward) {
        const int P = T + with_depot;
        float total = 0.0f;
        for (int b0 = 0; b0 < P; b0 += 64) {
            const int t = b0 + lane;
            float d = 0.0f;
            if (t < P) {
                int64_t a0, a1;
                if (with_depot) {
                    a0 = (t == 0) ? 0 : act[t - 1];
                    a1 = (t + 1 == P) ? 0 : act[t];
                } else {
                    a0 = act[t];
                    a1 = act[(t + 1 == P) ? 0 : t + 1];
                }
                a0 = a0 < 0 ? 0 : (a0 >= M ? M - 1 : a0);
                a1 = a1 < 0 ? 0 : (a1 >= M ? M - 1 : a1);
                const float2 p0 = *reinterpret_cast<const float2*>(L + 2 * a0);
                const float2 p1 = *reinterpret_cast<const float2*>(L + 2 * a1);
                const float dx = p1.x - p0.x, dy = p1.y - p0.y;
                d = __builtin_sqrtf(fma_(dy, dy, dx * dx));
            }
            const float s = wave_tree_sum(d);
            total = (b0 == 0) ? s : total + s;
        }
        if (lane == 0) reward[r] = -total;
    }
    // ---- log-likelihood: strictly sequential over the steps --------------------------------------------------------------
    if (ll) {
        float s = 0.0f;
        for (int t0 = 0; t0 < T; t0 += 64) {
            const float v = (t0 + lane < T) ? logp[r * ld + t0 + lane] : 0.0f;
            const int tn = T - t0 < 64 ? T - t0 : 64;
#pragma unroll
            for (int t = 0; t < 64; ++t)
                if (t < tn) s = s + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), t));
        }
        if (lane == 0) ll[r] = s;
    }
    // ---- validity ---------------------------------------------------------------------------------------------------------
    if (!bad) return;
    uint32_t* seen = seen_all[wv];
    for (int i = lane; i < 128; i += 64) seen[i] = 0;
    __builtin_amdgcn_wave_barrier();
    const int N = with_depot ? M - 1 : M;
    const int top = with_depot ? N : N - 1;
    int bad_lane = 0;
    for (int t = lane; t < T; t += 64) {
        const int64_t a = act[t];
        if (a < 0 || a > top) { bad_lane = 1; continue; }
        if (with_depot && a == 0) continue;
        const uint32_t bit = 1u << (a & 31);
        const uint32_t old = atomicOr(&seen[a >> 5], bit);
        if (old & bit) bad_lane = 1;
    }
    __builtin_amdgcn_wave_barrier();
    for (int n = with_depot + lane; n <= top; n += 64)
        if (!((seen[n >> 5] >> (n & 31)) & 1u)) bad_lane = 1;
    const bool invalid = __ballot(bad_lane != 0) != 0ull;
    if (invalid) {
        if (lane == 0) atomicAdd(&bad[0], 1);
        return;
    }
    if (with_depot) {
        // running load, depot resets it (cvrp/env.py:172-185): strictly sequential over the steps; the 64 load changes of
        // a block are fetched by the lanes at once and consumed in step order through v_readlane (every lane the same chain)
        const float cap = vcap[r], lim = cap + 1e-5f;
        const float* dem = demand + (r % B) * N;
        float usedc = 0.0f;
        int over = 0;
        for (int t0 = 0; t0 < T; t0 += 64) {
            float delta = 0.0f;
            if (t0 + lane < T) {
                const int64_t a = act[t0 + lane];
                delta = (a == 0) ? -cap : dem[a - 1];
            }
            const int tn = T - t0 < 64 ? T - t0 : 64;
#pragma unroll
            for (int t = 0; t < 64; ++t)
                if (t < tn) {
                    usedc = usedc + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, delta), t));
                    if (usedc < 0.0f) usedc = 0.0f;
                    if (usedc > lim) over = 1;
                }
        }
        if (over && lane == 0) atomicAdd(&bad[1], 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Several small device copies / zero fills in one launch (state clones and output buffers of a rollout, the input
// refresh of a replayed graph): blockIdx.y = segment, 16-byte vectors where both ends are aligned.
// ---------------------------------------------------------------------------------------------------------------------
struct MultiCopyArgs { const char* src[EAMRL_MULTI_COPY_MAX]; char* dst[EAMRL_MULTI_COPY_MAX]; int64_t bytes[EAMRL_MULTI_COPY_MAX]; };

__global__ __launch_bounds__(256) void k_multi_copy(MultiCopyArgs a)
{
    const int seg = blockIdx.y;
    const char* src = a.src[seg];
    char* dst = a.dst[seg];
    const int64_t n = a.bytes[seg];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool vec = (((uintptr_t)dst | (uintptr_t)src) & 15) == 0;
    const int64_t nv = vec ? n >> 4 : 0;
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (int64_t i = i0; i < nv; i += stride)
        reinterpret_cast<uint4*>(dst)[i] = src ? reinterpret_cast<const uint4*>(src)[i] : z;
    for (int64_t i = (nv << 4) + i0; i < n; i += stride) dst[i] = src ? src[i] : (char)0;
}

static inline unsigned row_blocks(int64_t R) { return (unsigned)((R + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK); }

int launch_tsp_step(uint8_t* mask, int64_t* first, int64_t* cur, int64_t* istep, const int64_t* action,
                    uint8_t* done, int64_t R, int N, hipStream_t st)
{
    hipLaunchKernelGGL(k_tsp_step, dim3(row_blocks(R)), dim3(EB), 0, st, mask, first, cur, istep, action, done, R, N);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_cvrp(int step, uint8_t* visited, float* used, const float* vcap, const float* demand, int64_t* cur,
                const int64_t* action, uint8_t* mask, uint8_t* done, int64_t R, int64_t B, int N, hipStream_t st)
{
    if (step)
        hipLaunchKernelGGL(k_cvrp_step_mask<1>, dim3(row_blocks(R)), dim3(EB), 0, st, visited, used, vcap, demand, cur,
                           action, mask, done, R, B, N);
    else
        hipLaunchKernelGGL(k_cvrp_step_mask<0>, dim3(row_blocks(R)), dim3(EB), 0, st, visited, used, vcap, demand, cur,
                           action, mask, done, R, B, N);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_sdvrp(float* rem, float* used, const float* vcap, int64_t* cur, const int64_t* action, uint8_t* mask,
                 uint8_t* done, int64_t R, int M, hipStream_t st)
{
    if (action)
        hipLaunchKernelGGL(k_sdvrp_step_mask<1>, dim3(row_blocks(R)), dim3(EB), 0, st, rem, used, vcap, cur, action, mask,
                           done, R, M);
    else
        hipLaunchKernelGGL(k_sdvrp_step_mask<0>, dim3(row_blocks(R)), dim3(EB), 0, st, rem, used, vcap, cur, action, mask,
                           done, R, M);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_cvrptw(uint8_t* visited, float* used, const float* vcap, const float* demand, int64_t* cur, float* time,
                  const float* locs, const float* tw, const float* dur, const int64_t* action, uint8_t* mask, uint8_t* done,
                  int64_t R, int64_t B, int N, hipStream_t st)
{
    if (action)
        hipLaunchKernelGGL(k_cvrptw_step_mask<1>, dim3(row_blocks(R)), dim3(EB), 0, st, visited, used, vcap, demand, cur, time,
                           locs, tw, dur, action, mask, done, R, B, N);
    else
        hipLaunchKernelGGL(k_cvrptw_step_mask<0>, dim3(row_blocks(R)), dim3(EB), 0, st, visited, used, vcap, demand, cur, time,
                           locs, tw, dur, action, mask, done, R, B, N);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_cvrptw_check(const int64_t* actions, const float* locs, const float* tw, const float* dur, int64_t R, int64_t B,
                        int M, int T, int32_t* bad, hipStream_t st)
{
    hipLaunchKernelGGL(k_check_cvrptw_time, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, actions, locs, tw, dur, R, B,
                       M, T, bad);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_op(uint8_t* visited, float* tour_len, float* prize_tot, const float* prize, const float* locs, const float* maxlen,
              int64_t* cur, int64_t* istep, const int64_t* action, uint8_t* mask, uint8_t* done, int64_t R, int64_t B, int M,
              hipStream_t st)
{
    if (action)
        hipLaunchKernelGGL(k_op_step_mask<1>, dim3(row_blocks(R)), dim3(EB), 0, st, visited, tour_len, prize_tot, prize, locs,
                           maxlen, cur, istep, action, mask, done, R, B, M);
    else
        hipLaunchKernelGGL(k_op_step_mask<0>, dim3(row_blocks(R)), dim3(EB), 0, st, visited, tour_len, prize_tot, prize, locs,
                           maxlen, cur, istep, action, mask, done, R, B, M);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_op_reward(const float* prize, const int64_t* actions, float* reward, int64_t R, int64_t B, int M, int T,
                     hipStream_t st)
{
    hipLaunchKernelGGL(k_op_reward, dim3(row_blocks(R)), dim3(EB), 0, st, prize, actions, reward, R, B, M, T);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_op_check(const int64_t* actions, const float* locs, const float* maxlen, int64_t R, int64_t B, int M, int T,
                    int32_t* bad, hipStream_t st)
{
    hipLaunchKernelGGL(k_check_op, dim3(row_blocks(R)), dim3(EB), 0, st, actions, locs, maxlen, R, B, M, T, bad);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_pctsp(uint8_t* visited, float* prize_tot, float* pen_tot, const float* prize, const float* penalty, int64_t* cur,
                 int64_t* istep, const int64_t* action, uint8_t* mask, uint8_t* done, int64_t R, int64_t B, int M,
                 hipStream_t st)
{
    if (action)
        hipLaunchKernelGGL(k_pctsp_step_mask<1>, dim3(row_blocks(R)), dim3(EB), 0, st, visited, prize_tot, pen_tot, prize,
                           penalty, cur, istep, action, mask, done, R, B, M);
    else
        hipLaunchKernelGGL(k_pctsp_step_mask<0>, dim3(row_blocks(R)), dim3(EB), 0, st, visited, prize_tot, pen_tot, prize,
                           penalty, cur, istep, action, mask, done, R, B, M);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_tour_length(const float* locs, const int64_t* actions, float* reward, int64_t R, int64_t B, int M, int T,
                       int with_depot, hipStream_t st, const float* penalty)
{
    hipLaunchKernelGGL(k_tour_length, dim3(row_blocks(R)), dim3(EB), 0, st, locs, actions, reward, R, B, M, T, with_depot,
                       penalty);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_sum_logp(const float* logp, int64_t ld, float* out, int64_t R, int T, hipStream_t st)
{
    hipLaunchKernelGGL(k_sum_logp, dim3((unsigned)((R + 63) / 64)), dim3(256), 0, st, logp, ld, out, R, T);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_replay_states(int env, const uint8_t* mask, const uint8_t* visited, const float* used, const float* vcap,
                         const int64_t* cur, const int64_t* istep, const float* time, const float* demand, const float* locs,
                         const float* tw, const float* dur, const int64_t* actions, uint32_t* bits, int32_t* idxA, float* sc,
                         int64_t R, int64_t B, int M, int T, hipStream_t st)
{
    ReplayArgs a{mask, visited, used, vcap, cur, istep, time, demand, locs, tw, dur, actions, bits, idxA, sc, R, B, M, T};
    const dim3 grid(row_blocks(R)), block(EB);
    if (M > 112) {              // the chunked mask layout (graphs above 112 nodes)
        if (env == EAMRL_ENV_CVRP) hipLaunchKernelGGL((k_replay_states<EAMRL_ENV_CVRP, 1024>), grid, block, 0, st, a);
        else if (env == EAMRL_ENV_CVRPTW) hipLaunchKernelGGL((k_replay_states<EAMRL_ENV_CVRPTW, 1024>), grid, block, 0, st, a);
        else if (env == EAMRL_ENV_PCTSP) hipLaunchKernelGGL((k_replay_states<EAMRL_ENV_PCTSP, 1024>), grid, block, 0, st, a);
        else if (env == EAMRL_ENV_OP) hipLaunchKernelGGL((k_replay_states<EAMRL_ENV_OP, 1024>), grid, block, 0, st, a);
    } else if (env == EAMRL_ENV_CVRP) hipLaunchKernelGGL(k_replay_states<EAMRL_ENV_CVRP>, grid, block, 0, st, a);
    else if (env == EAMRL_ENV_CVRPTW) hipLaunchKernelGGL(k_replay_states<EAMRL_ENV_CVRPTW>, grid, block, 0, st, a);
    else if (env == EAMRL_ENV_PCTSP) hipLaunchKernelGGL(k_replay_states<EAMRL_ENV_PCTSP>, grid, block, 0, st, a);
    else if (env == EAMRL_ENV_OP) hipLaunchKernelGGL(k_replay_states<EAMRL_ENV_OP>, grid, block, 0, st, a);
    else return EAMRL_E_ARG;
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_rollout_finish(int env, const float* locs, const int64_t* actions, const float* logp, int64_t ld, const float* demand,
                          const float* vcap, float* reward, float* ll, int32_t* bad, int64_t R, int64_t B, int M, int T,
                          hipStream_t st)
{
    hipLaunchKernelGGL(k_rollout_finish, dim3(row_blocks(R)), dim3(EB), 0, st, env, locs, actions, logp, ld, demand, vcap, reward,
                       ll, bad, R, B, M, T);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_multi_copy(int n, const void* const* src, void* const* dst, const int64_t* bytes, hipStream_t st)
{
    MultiCopyArgs a;
    int64_t mx = 0;
    int m = 0;
    for (int i = 0; i < n; ++i) {
        if (bytes[i] <= 0) continue;
        a.src[m] = (const char*)src[i]; a.dst[m] = (char*)dst[i]; a.bytes[m] = bytes[i];
        mx = bytes[i] > mx ? bytes[i] : mx;
        ++m;
    }
    if (m == 0) return 0;
    int64_t bx = (mx / 16 + 255) / 256;
    bx = bx < 1 ? 1 : (bx > 2048 ? 2048 : bx);
    hipLaunchKernelGGL(k_multi_copy, dim3((unsigned)bx, (unsigned)m), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_beam_topk(const float* logprobs, const float* parent, int64_t B, int BW, int M, int64_t* node, int32_t* beam,
                     float* cum, float* step_lp, hipStream_t st)
{
    const size_t lds = ((size_t)BW * M + 8) * sizeof(float);
    if (lds > 150 * 1024) return EAMRL_E_ARG;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(k_beam_topk),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return EAMRL_E_LAUNCH;
    hipLaunchKernelGGL(k_beam_topk, dim3((unsigned)B), dim3(256), lds, st, logprobs, parent, B, BW, M, node, beam, cum, step_lp);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

int launch_check_solution(int env, const int64_t* actions, const float* demand, const float* vcap, int64_t R, int64_t B,
                          int N, int T, int32_t* bad, hipStream_t st)
{
    if (N + 1 > 4096) return EAMRL_E_ARG;
    if (env == EAMRL_ENV_PCTSP) {       // demand = real_prize [B][N+1]
        hipLaunchKernelGGL(k_check_pctsp, dim3(row_blocks(R)), dim3(EB), 0, st, actions, demand, R, B, N + 1, T, bad);
        return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
    }
    if (env == EAMRL_ENV_SDVRP) {
        const size_t lds = (size_t)ROWS_PER_BLOCK * (N + 1) * sizeof(float);
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(k_check_sdvrp),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return EAMRL_E_LAUNCH;
        hipLaunchKernelGGL(k_check_sdvrp, dim3(row_blocks(R)), dim3(EB), lds, st, actions, demand, vcap, R, B, N, T, bad);
        return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
    }
    hipLaunchKernelGGL(k_check_solution, dim3(row_blocks(R)), dim3(EB), 0, st, env, actions, demand, vcap, R, B, N, T, bad);
    return hipGetLastError() == hipSuccess ? 0 : EAMRL_E_LAUNCH;
}

}  // namespace eamrl
