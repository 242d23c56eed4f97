"""CPU restatement of the fork's evolutionary operators for TSP -- TEST INFRASTRUCTURE ONLY (imported by tests/,
never by the product; see oracle/oracle.py).

Follows rl4co/models/zoo/earl/evolution.py of the reference:
  tsp_fitness ............ calculate_fitness_tsp          evolution.py:356-362  (fitness = f32(1.5*N) - cost)
  elitism_selection ...... elitism_selection              evolution.py:1103-1108
  order_crossover_tsp .... order_crossover_tsp            evolution.py:392-488
  inverse_mutate_tsp ..... inverse_mutate_tsp             evolution.py:490-517
  merge_* ................ EA.run, replacement step       evolution.py:300-350
  ea_run_tsp ............. EA.run                         evolution.py:252-354
  generate_population .... generate_population (tsp)      evolution.py:1574-1601
  worker_layout .......... evolution_worker               evolution.py:28-123  (reshapes only)

The reference draws its random numbers from numba's per-thread np.random inside the operators, so its results
are not reproducible run to run.  Here (and in the HIP kernels) every draw is an INPUT: per generation one
uniform per crossover pair and per offspring, and two cut indices each (used only where the uniform fires).
Parity status: pinned by tests/golden/ea_*.npz, which were produced by running the reference's own functions
with their draws recorded (tests/golden/make_golden_ea.py).

Defined choices where the reference leaves the order to numpy's unstable argsort: sorting is STABLE ascending
(ties keep index order); the reversed order used by the top-k replacement is that order reversed.
The cost is the canonical tour length of oracle/eamrl_oracle.c (orc_tour_length), as in the rollout path.
"""
from __future__ import annotations

import numpy as np

from . import oracle as orc


def tsp_cost(locs_b, pop):
    """cost = -reward of closed tours `pop` [n, N] on one instance locs_b [N, 2] (EA.get_cost, evolution.py:175-198)."""
    return -orc.tour_length_reward(locs_b[None], np.ascontiguousarray(pop, dtype=np.int64), with_depot=False)


def tsp_fitness(costs, problem_size):
    return (np.float32(1.5 * problem_size) - costs.astype(np.float32)).astype(np.float32)


def num_elites(selection_rate, pop_size):
    """EA.run + elitism_selection: populations of <= 2 are used whole; `idx[-0:]` is the whole array."""
    if pop_size <= 2:
        return pop_size
    ne = int(selection_rate * pop_size)
    return pop_size if ne == 0 else ne


def elitism_selection(pop, fitness, selection_rate):
    ne = num_elites(selection_rate, pop.shape[0])
    if pop.shape[0] <= 2:
        return pop.copy()
    idx = np.argsort(fitness, kind="stable")
    return pop[idx[-ne:]]


def adjusted_rate(num_pairs, rate):
    if num_pairs > 1:
        return max(0.0, min(1.0, (num_pairs * rate - 1.0) / (num_pairs - 1)))
    return rate


def order_crossover_tsp(parents, crossover_rate, cross_rand, cross_idx):
    """parents [n, N]; cross_rand [n//2] uniforms; cross_idx [n//2, 2] cut points in [1, N).  The first pair always
    crosses (its uniform is replaced by 0), the others with the adjusted rate."""
    n, N = parents.shape
    n -= n % 2
    P = n // 2
    rate = float(np.float32(crossover_rate))          # the reference signature takes a float32 rate
    off = np.zeros((n, N), dtype=np.int64)
    for p in range(P):
        pa, pb = parents[2 * p], parents[2 * p + 1]
        r = 0.0 if p == 0 else float(cross_rand[p])
        if not (r < (rate if p == 0 else adjusted_rate(P, rate))):
            off[2 * p], off[2 * p + 1] = pa, pb
            continue
        i1, i2 = int(cross_idx[p, 0]), int(cross_idx[p, 1])
        start, end = min(i1, i2), max(i1, i2)
        for own, other, row in ((pa, pb, 2 * p), (pb, pa, 2 * p + 1)):
            o = np.full(N, -1, dtype=np.int64)
            used = np.zeros(2 * N, dtype=bool)
            o[0] = own[0]
            used[own[0]] = True
            o[start:end] = own[start:end]
            used[own[start:end]] = True
            pos = end % N
            for _ in range(N):
                if pos != 0 and o[pos] == -1:
                    for node in other:
                        if not used[node]:
                            o[pos] = node
                            used[node] = True
                            break
                pos = (pos + 1) % N
            off[row] = o
    return off


def inverse_mutate_tsp(pop, mutation_rate, mut_rand, mut_idx):
    out = pop.copy()
    n, N = pop.shape
    for i in range(n):
        if float(mut_rand[i]) < mutation_rate:
            i1, i2 = int(mut_idx[i, 0]), int(mut_idx[i, 1])
            start, end = min(i1, i2), max(i1, i2)
            if start < end:
                out[i, start:end] = out[i, start:end][::-1].copy()
            elif start < N - 1:
                out[i, start], out[i, start + 1] = out[i, start + 1], out[i, start]
    return out


def merge_by_first_node(pop, fitness, offspring, off_fitness, first_nodes):
    """Position s keeps the best individual among pop[s] and the offspring that start at first_nodes[s]; on ties the
    earliest of (pop, offspring in order) wins (stable descending sort of the reference)."""
    new_pop, new_fit = pop.copy(), fitness.copy()
    for s in range(pop.shape[0]):
        for t in range(offspring.shape[0]):
            if offspring[t, 0] == first_nodes[s] and off_fitness[t] > new_fit[s]:
                new_pop[s], new_fit[s] = offspring[t], off_fitness[t]
    return new_pop, new_fit


def merge_top_k(pop, fitness, offspring, off_fitness):
    comb = np.vstack([pop, offspring])
    cf = np.concatenate([fitness, off_fitness])
    order = np.argsort(cf, kind="stable")[::-1][:pop.shape[0]]
    return comb[order], cf[order]


def ea_run_tsp(locs_b, init_pop, num_generations, mutation_rate, crossover_rate, selection_rate,
               cross_rand, cross_idx, mut_rand, mut_idx):
    """One instance.  init_pop [S, N]; draws [G, P], [G, P, 2], [G, O], [G, O, 2] with P = elites // 2, O = 2 P.
    -> (pop [S, N], fitness [S])."""
    S, N = init_pop.shape
    pop = init_pop.copy()
    fit = tsp_fitness(tsp_cost(locs_b, pop), N)
    first_nodes = init_pop[:, 0].copy()
    unique_first = len(np.unique(first_nodes)) == S
    for g in range(num_generations):
        sel = elitism_selection(pop, fit, selection_rate)
        off = order_crossover_tsp(sel, crossover_rate, cross_rand[g], cross_idx[g])
        off = inverse_mutate_tsp(off, mutation_rate, mut_rand[g], mut_idx[g])
        if len(off) == 0:
            continue
        off_fit = tsp_fitness(tsp_cost(locs_b, off), N)
        if unique_first:
            pop, fit = merge_by_first_node(pop, fit, off, off_fit, first_nodes)
        else:
            pop, fit = merge_top_k(pop, fit, off, off_fit)
    return pop, fit


def generate_population_tsp(route, pop_size):
    """Rotations of one tour (single-start augmentation, evolution.py:1574-1595)."""
    n1 = len(route)
    pop = np.zeros((pop_size, n1), dtype=np.int64)
    pop[0] = route
    for i in range(1, pop_size):
        start = i % n1
        if start == 0:
            start = 1
        pop[i] = np.roll(route, -start)
    return pop


# ------------------------------------------------------------------------------------------------------------
# CVRP operators (evolution.py:519-553 inverse_mutate_cvrp, :585-788 order_crossover_cvrp, :364-370 fitness)
# ------------------------------------------------------------------------------------------------------------
class StructuredDraws:
    """Integer draws from per-slot uniforms: randint(lo, hi) = lo + min(floor(u * (hi - lo)), hi - lo - 1), the rule of
    the HIP kernel.  `u` maps a slot key to a float in [0, 1)."""

    def __init__(self, u):
        self.u = u

    def __call__(self, lo, hi, key):
        n = hi - lo
        return lo + min(int(float(self.u[key]) * n), n - 1)


def cvrp_cost(locs_b, pop):
    """cost = -reward: closed tour depot -> actions -> depot (cvrp/env.py:146-155), locs_b [M, 2] with the depot first."""
    return -orc.tour_length_reward(locs_b[None], np.ascontiguousarray(pop, dtype=np.int64), with_depot=True)


def cvrp_fitness(costs, problem_size):
    return (np.float32(2.5 * problem_size) - costs.astype(np.float32)).astype(np.float32)


def inverse_mutate_cvrp(pop, mutation_rate, mut_rand, rint, slot):
    """Reverse a random segment inside one route.  rint(lo, hi, key) supplies randint(lo, hi); keys are
    (slot, individual, 0|1|2) for the route, the segment start and the segment end."""
    out = pop.copy()
    n, L = pop.shape
    for i in range(n):
        if not (float(mut_rand[i]) < mutation_rate):
            continue
        depots = np.flatnonzero(out[i] == 0)
        if len(depots) > 1:
            r = rint(0, len(depots) - 1, (slot, i, 0))
            start, end = depots[r] + 1, depots[r + 1] - 1
            if end - start > 1:
                s0 = rint(start, end, (slot, i, 1))
                s1 = rint(s0 + 1, end + 1, (slot, i, 2))
                if s0 < s1:
                    out[i, s0:s1] = out[i, s0:s1][::-1].copy()
    return out


def _cvrp_child(parent, end, full_demand, vehicle_capacity, num_customers):
    """One child of order_crossover_cvrp: the parent's first `end` routes, then every customer not yet placed in
    ascending index order, a new route whenever the (float64) load would exceed the capacity."""
    L = len(parent)
    end_idx = int(np.flatnonzero(parent == 0)[end]) if end > 0 else 0
    o = np.full(2 * L, -1, dtype=np.int64)
    o[:end_idx] = parent[:end_idx]
    used = np.zeros(num_customers + 1, dtype=bool)
    for j in range(end_idx):
        if o[j] > 0:
            used[o[j]] = True
    pos = end_idx
    if pos > 0 and o[pos - 1] != 0:
        o[pos] = 0
        pos += 1
    load = 0.0
    remaining = [i for i in range(1, num_customers + 1) if not used[i]]
    count = len(remaining)
    for i, node in enumerate(remaining):
        if pos >= 2 * L - 1:
            break
        if load + float(full_demand[node]) > vehicle_capacity:
            if pos > 0 and o[pos - 1] == 0 and i < count - 1:
                continue
            o[pos] = 0
            pos += 1
            load = 0.0
            if pos >= 2 * L - 1:
                break
        o[pos] = node
        load += float(full_demand[node])
        pos += 1
    if pos < 2 * L and o[pos - 1] != 0:
        all_visited = True
        for i in range(1, num_customers + 1):
            if not used[i] and i < count and remaining[i] > 0:
                all_visited = False
                break
        if all_visited:
            o[pos] = 0
            pos += 1
    invalid = False
    for j in range(1, pos):
        if o[j] == 0 and o[j - 1] == 0:
            if count > 0:          # some remaining node is by construction never marked used
                invalid = True
            break
    if invalid:
        return parent.copy()
    last_valid = int(np.max(np.flatnonzero(o != -1)))
    if last_valid >= L:
        return parent.copy()
    child = o[:L].copy()
    child[child == -1] = 0
    return child


def order_crossover_cvrp(parents, crossover_rate, demand, vehicle_capacity, cross_rand, rint, slot):
    n, L = parents.shape
    n -= n % 2
    P = n // 2
    num_customers = len(demand)
    full_demand = np.concatenate([[np.float32(0)], demand.astype(np.float32)])
    off = np.zeros((n, L), dtype=np.int64)
    for p in range(P):
        pa, pb = parents[2 * p], parents[2 * p + 1]
        r = 0.0 if p == 0 else float(cross_rand[p])
        if not (r < (crossover_rate if p == 0 else adjusted_rate(P, crossover_rate))):
            off[2 * p], off[2 * p + 1] = pa, pb
            continue
        routes = []
        for par in (pa, pb):
            valid_end = int(np.max(np.flatnonzero(par != 0))) + 1 if np.any(par != 0) else 1
            routes.append(int(np.count_nonzero(par[:valid_end] == 0)))
        m = min(routes)
        end = rint(1, m, (slot, p, 0)) if m > 1 else 0
        off[2 * p] = _cvrp_child(pa, end, full_demand, vehicle_capacity, num_customers)
        off[2 * p + 1] = _cvrp_child(pb, end, full_demand, vehicle_capacity, num_customers)
    return off


def ea_run_cvrp(locs_b, demand_b, vehicle_capacity, init_pop, num_generations, mutation_rate, crossover_rate,
                selection_rate, init_mut_rand, cross_rand, mut_rand, rint, top_k=False):
    """EA.run for one CVRP instance (evolution.py:252-354): an initial mutation pass, then generations.
    init_mut_rand [S]; cross_rand [G, P]; mut_rand [G, O]; rint as in inverse_mutate_cvrp with slots
    ("init",), ("cross", g), ("mut", g).  top_k = the `method == "am"` replacement."""
    S, L = init_pop.shape
    pop = inverse_mutate_cvrp(init_pop, mutation_rate, init_mut_rand, rint, ("init",))
    fit = cvrp_fitness(cvrp_cost(locs_b, pop), L)
    first_nodes = init_pop[:, 0].copy()
    unique_first = len(np.unique(first_nodes)) == S
    for g in range(num_generations):
        sel = elitism_selection(pop, fit, selection_rate)
        off = order_crossover_cvrp(sel, crossover_rate, demand_b, vehicle_capacity, cross_rand[g], rint, ("cross", g))
        off = inverse_mutate_cvrp(off, mutation_rate, mut_rand[g], rint, ("mut", g))
        if len(off) == 0:
            continue
        off_fit = cvrp_fitness(cvrp_cost(locs_b, off), L)
        if unique_first and not top_k:
            pop, fit = merge_by_first_node(pop, fit, off, off_fit, first_nodes)
        else:
            pop, fit = merge_top_k(pop, fit, off, off_fit)
    return pop, fit


# ------------------------------------------------------------------------------------------------------------
# PCTSP operators (evolution.py:555-583 inverse_mutate_pctsp, :905-1101 cycle_crossover_pctsp; fitness = the CVRP one)
# ------------------------------------------------------------------------------------------------------------
# numba typing restated: float32 array elements added to a `0.0` accumulator are summed in float64; `prize_ratios` is a
# float32 array, so each ratio is computed in float64 and ROUNDED to float32 when stored.
# Defined choice: `next(iter(remaining_nodes))` of the reference iterates a hash set; for node ids below the table size
# (any graph here) both CPython's and numba's open-addressing tables hold small ints at slot == value, i.e. the
# iteration starts at the SMALLEST remaining node.  That is the rule here and in the kernel.
def pctsp_cost(locs_b, penalty_b, pop):
    """cost = -reward = (length + penalties of everything) - saved penalties (pctsp/env.py:158-178); locs_b [M, 2] and
    penalty_b [M] with the depot first."""
    return -orc.pctsp_reward(locs_b[None], penalty_b[None], np.ascontiguousarray(pop, dtype=np.int64))


def inverse_mutate_pctsp(pop, mutation_rate, mut_rand, rint, slot):
    """Reverse [start, end) with both ends drawn from [1, index of the last non-zero entry); equal ends swap two
    neighbours.  Rows with fewer than two entries before the last non-zero one are left alone (the reference's
    randint(1, 1) raises there)."""
    out = pop.copy()
    n, L = pop.shape
    for i in range(n):
        if not (float(mut_rand[i]) < mutation_rate):
            continue
        nz = np.flatnonzero(out[i] != 0)
        if len(nz) == 0 or nz[-1] < 2:
            continue
        valid_len = int(nz[-1])
        i1 = rint(1, valid_len, (slot, i, 0))
        i2 = rint(1, valid_len, (slot, i, 1))
        start, end = min(i1, i2), max(i1, i2)
        if start < end:
            out[i, start:end] = out[i, start:end][::-1].copy()
        elif start < L - 1:
            out[i, start], out[i, start + 1] = out[i, start + 1], out[i, start]
    return out


def _valid_end_from_zero(par):
    """index after the last non-zero entry, scanning down to index 0 (0 when the row is all zeros)."""
    nz = np.flatnonzero(par != 0)
    return int(nz[-1]) + 1 if len(nz) else 0


def _pctsp_fill(nodes, prize, penalty, num_customers):
    """Keep first occurrences, then append unused customers by descending float32 prize / penalty ratio (first index
    wins ties) until the float64 prize sum reaches 1 - 1e-5."""
    used = np.zeros(num_customers + 1, dtype=bool)
    total = 0.0
    out = []
    for node in nodes:
        if node > 0 and not used[node]:
            out.append(int(node)); used[node] = True
            total += float(prize[node])
    if total < 1.0 - 1e-5:
        ratios = np.zeros(num_customers + 1, dtype=np.float32)
        for i in range(1, num_customers + 1):
            if not used[i]:
                ratios[i] = np.float32(float(prize[i]) / (float(penalty[i]) + 1e-10))
        while total < 1.0 - 1e-5:
            best, best_ratio = 0, -1.0
            for i in range(1, num_customers + 1):
                if not used[i] and float(ratios[i]) > best_ratio:
                    best_ratio, best = float(ratios[i]), i
            if best == 0:
                break
            out.append(best); used[best] = True
            total += float(prize[best])
    return out


def cycle_crossover_pctsp(parents, crossover_rate, prize, penalty, cross_rand):
    """parents [n, L]; prize / penalty [N + 1] float32 with the depot first.  No integer draws."""
    n, L = parents.shape
    n -= n % 2
    P = n // 2
    num_customers = len(prize) - 1
    off = np.zeros((n, L), dtype=np.int64)
    for p in range(P):
        pa, pb = parents[2 * p], parents[2 * p + 1]
        r = 0.0 if p == 0 else float(cross_rand[p])
        if not (r < (crossover_rate if p == 0 else adjusted_rate(P, crossover_rate))):
            off[2 * p], off[2 * p + 1] = pa, pb
            continue
        p1 = [int(x) for x in pa[:_valid_end_from_zero(pa)] if x > 0]
        p2 = [int(x) for x in pb[:_valid_end_from_zero(pb)] if x > 0]
        pos1 = {node: i for i, node in enumerate(p1)}
        pos2 = {node: i for i, node in enumerate(p2)}
        remaining = set(p1)
        cycles = []
        while remaining:
            start = min(remaining)
            node, cycle = start, []
            while True:
                cycle.append(node)
                remaining.remove(node)
                if node not in pos2:
                    break
                k = pos2[node]
                if k >= len(p1):
                    break
                node = p1[k]
                if node == start or node not in remaining:
                    break
            cycles.append(cycle)
        o1, o2 = [], []
        for i, cycle in enumerate(cycles):
            if i % 2 == 0:
                o1.extend(cycle)
                o2.extend(p2[pos2[node]] for node in cycle if node in pos2 and pos2[node] < len(p2))
            else:
                o2.extend(cycle)
                o1.extend(p1[pos1[node]] for node in cycle if node in pos1 and pos1[node] < len(p1))
        for row, nodes in ((2 * p, o1), (2 * p + 1, o2)):
            child = _pctsp_fill(nodes, prize, penalty, num_customers)[:L]
            off[row, :len(child)] = child
    return off


def ea_run_pctsp(locs_b, prize_b, penalty_b, init_pop, num_generations, mutation_rate, crossover_rate, selection_rate,
                 init_mut_rand, cross_rand, mut_rand, rint, top_k=False):
    """EA.run for one PCTSP instance: initial mutation pass, then generations (evolution.py:252-354).  prize_b is
    td["real_prize"], penalty_b td["penalty"], both [N + 1] with the depot first.  Slots as in ea_run_cvrp."""
    S, L = init_pop.shape
    pop = inverse_mutate_pctsp(init_pop, mutation_rate, init_mut_rand, rint, ("init",))
    fit = cvrp_fitness(pctsp_cost(locs_b, penalty_b, pop), L)
    first_nodes = init_pop[:, 0].copy()
    unique_first = len(np.unique(first_nodes)) == S
    for g in range(num_generations):
        sel = elitism_selection(pop, fit, selection_rate)
        off = cycle_crossover_pctsp(sel, crossover_rate, prize_b, penalty_b, cross_rand[g])
        off = inverse_mutate_pctsp(off, mutation_rate, mut_rand[g], rint, ("mut", g))
        if len(off) == 0:
            continue
        off_fit = cvrp_fitness(pctsp_cost(locs_b, penalty_b, off), L)
        if unique_first and not top_k:
            pop, fit = merge_by_first_node(pop, fit, off, off_fit, first_nodes)
        else:
            pop, fit = merge_top_k(pop, fit, off, off_fit)
    return pop, fit


# ------------------------------------------------------------------------------------------------------------
# OP operators (evolution.py:1110-1346 order_crossover_op, :1468-1572 inverse_mutate_op, :372-378 fitness)
# ------------------------------------------------------------------------------------------------------------
# The distance matrix is the float32 one of EA.run's calculate_distance_matrix (sqrt(dx*dx + dy*dy), every operation
# rounded to float32, no fused multiply-add); route lengths are float64 sums of those float32 entries (numba typing).
def op_dist_matrix(locs_b):
    locs = np.asarray(locs_b, dtype=np.float32)
    diff = locs[:, None, :] - locs[None, :, :]
    sq = diff * diff
    return np.sqrt(sq[..., 0] + sq[..., 1]).astype(np.float32)


def op_cost(prize_b, pop):
    """cost = -reward = -(sum of the collected prizes) (op/env.py:167-177); prize_b [M] with the depot (0) first."""
    return -orc.op_reward(prize_b[None], np.ascontiguousarray(pop, dtype=np.int64))


def op_fitness(costs):
    return (np.float32(0.0) - costs.astype(np.float32)).astype(np.float32)


def _valid_end_from_one(row):
    """index after the last non-zero entry at index >= 1; the row length when there is none (the reference's scan stops at 1)."""
    L = len(row)
    for j in range(L - 1, 0, -1):
        if row[j] != 0:
            return j + 1
    return L


def _op_checked(cand, dist, num_nodes, global_max):
    """The post-check of a candidate row: close it with a depot visit, length of legs 1.. (the leg from the depot to the
    first node is NOT counted there, as in the reference) <= max - 1e-5 and no customer twice (the scan stops at the first repeat)."""
    L = len(cand)
    t = cand.copy()
    ve = _valid_end_from_one(t)
    if t[ve - 1] != 0:
        if ve < L:
            t[ve] = 0
            ve += 1
        else:
            t[ve - 1] = 0
    total = 0.0
    dup = False
    visited = np.zeros(num_nodes, dtype=bool)
    for j in range(1, ve):
        total += float(dist[t[j - 1], t[j]])
        if t[j] != 0:
            if visited[t[j]]:
                dup = True
                break
            visited[t[j]] = True
    return t, (total <= float(global_max) - 1e-5) and not dup


def _op_child(parent, end, prize, dist, global_max):
    """One child of order_crossover_op: the parent's first `end` entries, then the customers 1..L (L = chromosome
    length; ids >= the node count do not exist and are skipped -- the reference reads out of bounds there) not yet used,
    in ascending order, each appended when the route so far + the leg to it + its way back to the depot fits max - 0.1."""
    L = len(parent)
    num_nodes = len(prize)
    safe = float(global_max) - 0.1
    o = np.full(2 * L, -1, dtype=np.int64)
    o[:end] = parent[:end]
    used = np.zeros(num_nodes, dtype=bool)
    for j in range(end):
        if o[j] != 0:
            used[o[j]] = True
    cur = 0.0
    for j in range(1, end):
        cur += float(dist[o[j - 1], o[j]])
    cur += float(dist[0, o[0]])
    pos = end
    for node in range(1, L + 1):
        if node >= num_nodes or used[node]:
            continue
        nxt = float(dist[o[pos - 1], node])
        back = float(dist[node, 0])
        if cur + nxt + back <= safe:
            o[pos] = node
            cur += nxt
            used[node] = True
            pos += 1
        if pos >= 2 * L - 2:
            break
    o[pos] = 0
    pos += 1
    last_valid = int(np.max(np.flatnonzero(o != -1)))
    if last_valid >= L:
        return parent.copy()
    cand = np.where(o[:L] == -1, 0, o[:L])
    t, ok = _op_checked(cand, dist, num_nodes, global_max)
    return t if ok else parent.copy()


def order_crossover_op(parents, crossover_rate, prize, dist, max_distances, cross_rand, rint, slot):
    n, L = parents.shape
    n -= n % 2
    P = n // 2
    global_max = max_distances[0]
    off = np.zeros((n, L), dtype=np.int64)
    for p in range(P):
        pa, pb = parents[2 * p].copy(), parents[2 * p + 1].copy()
        off[2 * p], off[2 * p + 1] = pa, pb
        r = 0.0 if p == 0 else float(cross_rand[p])
        if not (r < (crossover_rate if p == 0 else adjusted_rate(P, crossover_rate))):
            continue
        e1, e2 = _valid_end_from_one(pa), _valid_end_from_one(pb)
        if pa[e1 - 1] != 0 or pb[e2 - 1] != 0:
            continue
        max_cross = min(e1 - 1, e2 - 1, L - 1)
        if max_cross <= 1:
            continue
        end = rint(1, max_cross, (slot, p, 0))
        off[2 * p] = _op_child(pa, end, prize, dist, global_max)
        off[2 * p + 1] = _op_child(pb, end, prize, dist, global_max)
    return off


def inverse_mutate_op(pop, mutation_rate, prize, dist, max_distances, mut_rand, rint, slot):
    """Reverse [start, end] inside the route when the reversed route stays within max - 1e-5 and has no repeated customer;
    otherwise the row is restored (including the closing depot the operator may have written)."""
    out = pop.copy()
    n, L = pop.shape
    num_nodes = len(prize)
    safe = float(max_distances[0]) - 1e-5
    for i in range(n):
        if not (float(mut_rand[i]) < mutation_rate):
            continue
        ve = _valid_end_from_one(pop[i])
        if ve <= 3:
            continue
        if pop[i, ve - 1] != 0:
            if ve < L:
                out[i, ve] = 0
                ve += 1
            else:
                out[i, ve - 1] = 0
        row = out[i]
        cur = float(dist[0, row[0]])
        for j in range(1, ve):
            cur += float(dist[row[j - 1], row[j]])
        s = rint(1, ve - 2, (slot, i, 0))
        e = rint(s + 1, ve - 1, (slot, i, 1))
        success = False
        if s < e:
            old_sub = 0.0
            for j in range(s, e):
                old_sub += float(dist[row[j], row[j + 1]])
            old_conn = 0.0
            if s > 0:
                old_conn += float(dist[row[s - 1], row[s]])
            if e < ve - 1:
                old_conn += float(dist[row[e], row[e + 1]])
            t = row.copy()
            t[s:e + 1] = row[s:e + 1][::-1]
            new_sub = 0.0
            for j in range(s, e):
                new_sub += float(dist[t[j], t[j + 1]])
            new_conn = 0.0
            if s > 0:
                new_conn += float(dist[t[s - 1], t[s]])
            if e < ve - 1:
                new_conn += float(dist[t[e], t[e + 1]])
            change = (new_sub + new_conn) - (old_sub + old_conn)
            if cur + change <= safe:
                dup = False
                visited = np.zeros(num_nodes, dtype=bool)
                for j in range(ve):
                    if t[j] != 0:
                        if visited[t[j]]:
                            dup = True
                            break
                        visited[t[j]] = True
                total = float(dist[0, t[0]])
                for j in range(1, ve):
                    total += float(dist[t[j - 1], t[j]])
                if total <= safe and not dup:
                    out[i, s:e + 1] = t[s:e + 1]
                    success = True
        if not success:
            out[i] = pop[i]
    return out


def ea_run_op(locs_b, prize_b, max_length_b, init_pop, num_generations, mutation_rate, crossover_rate, selection_rate,
              init_mut_rand, cross_rand, mut_rand, rint, top_k=False):
    """EA.run for one OP instance.  locs_b [M, 2], prize_b [M], max_length_b [M] (td["max_length"]: only entry 0, the
    depot's, is used), all with the depot first."""
    S, L = init_pop.shape
    dist = op_dist_matrix(locs_b)
    pop = inverse_mutate_op(init_pop, mutation_rate, prize_b, dist, max_length_b, init_mut_rand, rint, ("init",))
    fit = op_fitness(op_cost(prize_b, pop))
    first_nodes = init_pop[:, 0].copy()
    unique_first = len(np.unique(first_nodes)) == S
    for g in range(num_generations):
        sel = elitism_selection(pop, fit, selection_rate)
        off = order_crossover_op(sel, crossover_rate, prize_b, dist, max_length_b, cross_rand[g], rint, ("cross", g))
        off = inverse_mutate_op(off, mutation_rate, prize_b, dist, max_length_b, mut_rand[g], rint, ("mut", g))
        if len(off) == 0:
            continue
        off_fit = op_fitness(op_cost(prize_b, off))
        if unique_first and not top_k:
            pop, fit = merge_by_first_node(pop, fit, off, off_fit, first_nodes)
        else:
            pop, fit = merge_top_k(pop, fit, off, off_fit)
    return pop, fit
