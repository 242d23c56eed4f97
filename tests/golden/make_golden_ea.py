#!/usr/bin/env python3
"""Golden vectors of the fork's evolutionary operators (rl4co/models/zoo/earl/evolution.py), produced by RUNNING
THE REFERENCE's own functions (EA.run, order_crossover_*, inverse_mutate_*, elitism_selection) in the build
container:

    python tests/golden/make_golden_ea.py

`numba` is absent, so `_refshim.install_numba()` turns `njit` into the identity and `prange` into `range`: the
operators then execute as the Python/numpy they are written in.  They draw from `np.random`; the draws are
recorded here and stored next to the results in the structured form the native kernels take them
(one uniform + two indices per crossover pair and per mutated individual, per generation), so a deterministic
implementation can replay them.  `EA.run` is called instance by instance (the reference spreads instances over
a thread pool, which only interleaves the draws).
"""
from __future__ import annotations

import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import _refshim  # noqa: E402

_refshim.install()
_refshim.install_numba()

import torch  # noqa: E402
from tensordict import TensorDict  # noqa: E402  (the stand-in)

from rl4co.envs.routing.cvrp.env import CVRPEnv  # noqa: E402
from rl4co.envs.routing.tsp.env import TSPEnv  # noqa: E402

ev = importlib.import_module("rl4co.models.zoo.earl.evolution")
torch.set_num_threads(1)


class DrawLog:
    """Wraps np.random.random / np.random.randint and keeps what they returned, in program order."""

    def __init__(self):
        self.events = []

    def __enter__(self):
        self._random, self._randint = np.random.random, np.random.randint

        def random(n=None):
            v = self._random(n)
            self.events.append(("random", np.array(v, dtype=np.float64).copy()))
            return v

        def randint(lo, hi=None):
            v = self._randint(lo, hi)
            self.events.append(("randint", (int(lo), int(hi), int(v))))
            return v

        np.random.random, np.random.randint = random, randint
        return self

    def __exit__(self, *exc):
        np.random.random, np.random.randint = self._random, self._randint


def adjusted_rate(num_pairs, rate):
    if num_pairs > 1:
        return max(0.0, min(1.0, (num_pairs * rate - 1.0) / (num_pairs - 1)))
    return rate


def structure_tsp_draws(events, G, P, O, crossover_rate, mutation_rate):
    """Flat draw log of one EA.run (TSP) -> cross_rand [G,P], cross_idx [G,P,2], mut_rand [G,O], mut_idx [G,O,2]."""
    cross_rand = np.zeros((G, P)); cross_idx = -np.ones((G, P, 2), dtype=np.int32)
    mut_rand = np.zeros((G, O)); mut_idx = -np.ones((G, O, 2), dtype=np.int32)
    it = iter(events)
    rate32 = float(np.float32(crossover_rate))       # order_crossover_tsp takes the rate as float32
    for g in range(G):
        kind, v = next(it); assert kind == "random" and v.shape == (P,)
        cross_rand[g] = v
        for p in range(P):
            r = 0.0 if p == 0 else v[p]                # cross_rand[0] = 0.0 in the reference
            if r < (rate32 if p == 0 else adjusted_rate(P, rate32)):
                for k in range(2):
                    kind, (lo, hi, x) = next(it); assert kind == "randint"
                    cross_idx[g, p, k] = x
        kind, v = next(it); assert kind == "random" and v.shape == (O,)
        mut_rand[g] = v
        for o in range(O):
            if v[o] < mutation_rate:
                for k in range(2):
                    kind, (lo, hi, x) = next(it); assert kind == "randint"
                    mut_idx[g, o, k] = x
    assert next(it, None) is None, "unconsumed draws"
    return cross_rand, cross_idx, mut_rand, mut_idx


def tsp_case(name, N, B, S, G, mutation_rate, crossover_rate, selection_rate, seed, duplicate_first=False):
    torch.manual_seed(seed)
    np.random.seed(seed)
    env = TSPEnv(generator_params=dict(num_loc=N))
    td = env.reset(batch_size=[B])
    locs = td["locs"].numpy().copy()
    ea = ev.EA(env, dict(num_generations=G, mutation_rate=mutation_rate, crossover_rate=crossover_rate,
                         selection_rate=selection_rate))
    init = np.zeros((B, S, N), dtype=np.int64)
    for b in range(B):
        for s in range(S):
            first = s % N if not duplicate_first else (s // 2) % N
            rest = np.random.permutation([x for x in range(N) if x != first])
            init[b, s] = np.concatenate([[first], rest])
    ne = int(selection_rate * S) if S > 2 else S
    ne = S if ne == 0 else ne
    P = (ne - ne % 2) // 2
    O = 2 * P
    out_pop = np.zeros_like(init); out_fit = np.zeros((B, S), dtype=np.float32)
    cr = np.zeros((G, B, P)); ci = np.zeros((G, B, P, 2), dtype=np.int32)
    mr = np.zeros((G, B, O)); mi = np.zeros((G, B, O, 2), dtype=np.int32)
    for b in range(B):
        env_td = TensorDict({"locs": td["locs"][b:b + 1]}, batch_size=[1])
        with DrawLog() as log:
            pop, fit = ea.run(init[b], env_td)
        out_pop[b], out_fit[b] = pop, fit
        a, i1, m, i2 = structure_tsp_draws(log.events, G, P, O, crossover_rate, mutation_rate)
        cr[:, b], ci[:, b], mr[:, b], mi[:, b] = a, i1, m, i2
    np.savez_compressed(os.path.join(HERE, name + ".npz"), env_name="tsp", locs=locs, init_pop=init,
                        num_generations=G, mutation_rate=mutation_rate, crossover_rate=crossover_rate,
                        selection_rate=selection_rate, cross_rand=cr, cross_idx=ci, mut_rand=mr, mut_idx=mi,
                        pop=out_pop, fitness=out_fit, torch_version=torch.__version__)
    print(f"{name}: B={B} S={S} N={N} G={G} elites={ne} pairs={P}; mean cost {np.mean(1.5 * N - out_fit):.4f}")


def operator_cases():
    """Single calls of the TSP operators with their draws (no fitness involved): exact integer fixtures."""
    np.random.seed(11)
    N, S = 13, 9            # odd population: the last parent is dropped by the crossover
    parents = np.stack([np.concatenate([[s], np.random.permutation([x for x in range(N) if x != s])])
                        for s in range(S)]).astype(np.int64)
    with DrawLog() as log:
        off = ev.order_crossover_tsp(parents, np.float32(0.9))
    P = S // 2
    cr, ci, _, _ = structure_tsp_draws(log.events + [("random", np.zeros(2 * P))], 1, P, 2 * P, 0.9, -1.0)
    with DrawLog() as log:
        mut = ev.inverse_mutate_tsp(off, 0.7)
    O = off.shape[0]
    mr = log.events[0][1]
    mi = -np.ones((O, 2), dtype=np.int32)
    it = iter(log.events[1:])
    for o in range(O):
        if mr[o] < 0.7:
            mi[o, 0] = next(it)[1][2]; mi[o, 1] = next(it)[1][2]
    fitness = np.random.random(S).astype(np.float32)
    sel, sel_fit = ev.elitism_selection(parents, fitness, 0.5)
    np.savez_compressed(os.path.join(HERE, "ea_tsp_operators.npz"), parents=parents, crossover_rate=0.9,
                        cross_rand=cr[0], cross_idx=ci[0], offspring=off, mutation_rate=0.7, mut_rand=mr,
                        mut_idx=mi, mutated=mut, fitness=fitness, selection_rate=0.5, selected=sel,
                        selected_fitness=sel_fit)
    print("ea_tsp_operators: crossed pairs", int((ci[0, :, 0] >= 0).sum()), "mutated", int((mi[:, 0] >= 0).sum()))


def main():
    operator_cases()
    tsp_case("ea_tsp20_default", N=20, B=3, S=20, G=3, mutation_rate=0.1, crossover_rate=0.6, selection_rate=0.2, seed=5)
    tsp_case("ea_tsp20_busy", N=20, B=3, S=20, G=4, mutation_rate=0.6, crossover_rate=0.9, selection_rate=0.7, seed=6)
    tsp_case("ea_tsp50_busy", N=50, B=2, S=50, G=3, mutation_rate=0.5, crossover_rate=0.8, selection_rate=0.5, seed=7)


if __name__ == "__main__":
    main()
