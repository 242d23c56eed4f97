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
