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


class _NumbaTypedNumpy:
    """numba unifies `load = 0.0; load += full_demand[node]` (float32 array element) to a float64 accumulator.  Plain
    Python with NumPy >= 2 would keep that sum in float32 (a Python float is a weak scalar), i.e. compute something
    the real reference never computes.  Under this proxy the operators' float32 work arrays are allocated as
    float64 -- they only ever receive float32 values, so their contents are unchanged -- and the sums come out
    in float64 as under numba.  Everything else is numpy itself."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def zeros(shape, dtype=float, **kw):
        if np.dtype(dtype) == np.float32:
            dtype = np.float64
        return np.zeros(shape, dtype=dtype, **kw)


ev.np = _NumbaTypedNumpy()


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


class FlatConsumer:
    """rint for oracle/ea_oracle.py that replays the reference's recorded draws in program order and writes down,
    per slot, the uniform that reproduces each integer under the kernels' rule lo + floor(u * (hi - lo))."""

    def __init__(self, events):
        self.it = iter(events)
        self.u = {}

    def vector(self, n):
        kind, v = next(self.it)
        assert kind == "random" and v.shape == (n,), (kind, getattr(v, "shape", None), n)
        return v

    def __call__(self, lo, hi, key):
        kind, (l, h, x) = next(self.it)
        assert kind == "randint" and (l, h) == (lo, hi), ((l, h), (lo, hi), key)
        self.u[key] = (x - lo + 0.5) / (hi - lo)
        return x

    def done(self):
        assert next(self.it, None) is None, "unconsumed draws"


def cvrp_case(name, N, B, S, G, mutation_rate, crossover_rate, selection_rate, seed, method=None):
    """EA.run of the reference on CVRP populations taken from random feasible tours; the flat draw log is turned
    into per-slot uniforms by replaying it through the oracle restatement, whose result must equal the reference's."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import ea_oracle as eo

    torch.manual_seed(seed)
    np.random.seed(seed)
    env = CVRPEnv(generator_params=dict(num_loc=N))
    td = env.reset(batch_size=[B])
    locs = td["locs"].numpy().copy()                 # [B, N+1, 2], depot first
    demand = td["demand"].numpy().copy()             # [B, N], normalised by the capacity
    vcap = float(td["vehicle_capacity"].reshape(-1)[0])
    ea = ev.EA(env, dict(num_generations=G, mutation_rate=mutation_rate, crossover_rate=crossover_rate,
                         selection_rate=selection_rate, method=method))
    # populations: S random feasible tours per instance, tour s starts at customer s + 1 (POMO layout)
    tours = []
    for b in range(B):
        rows = []
        for s in range(S):
            order = [s % N + 1] + [int(x) for x in np.random.permutation([c for c in range(1, N + 1) if c != s % N + 1])]
            row, load = [], 0.0
            for c in order:
                if load + float(demand[b, c - 1]) > vcap + 1e-6:
                    row.append(0); load = 0.0
                row.append(c); load += float(demand[b, c - 1])
            row.append(0)
            rows.append(row)
        tours.append(rows)
    L = max(len(r) for rows in tours for r in rows) + 2
    init = np.zeros((B, S, L), dtype=np.int64)
    for b in range(B):
        for s in range(S):
            init[b, s, :len(tours[b][s])] = tours[b][s]
    ne = int(selection_rate * S) if S > 2 else S
    ne = S if ne == 0 else ne
    P = (ne - ne % 2) // 2
    O = 2 * P
    out_pop = np.zeros_like(init); out_fit = np.zeros((B, S), dtype=np.float32)
    imr = np.zeros((B, S)); imu = np.zeros((B, S, 3))
    cr = np.zeros((G, B, P)); cu = np.zeros((G, B, P))
    mr = np.zeros((G, B, O)); mu = np.zeros((G, B, O, 3))
    for b in range(B):
        env_td = TensorDict({k: td[k][b:b + 1] for k in ("locs", "demand", "vehicle_capacity")}, batch_size=[1])
        with DrawLog() as log:
            pop, fit = ea.run(init[b], env_td)
        out_pop[b], out_fit[b] = pop, fit
        # replay through the restatement: vectors first (they are drawn before the integers of the same operator)
        fc = FlatConsumer(log.events)

        class Vec:      # hands the recorded uniform vectors to ea_run_cvrp in program order
            pass
        # the oracle takes the vectors as arrays, so peel them off the log in the order EA.run draws them
        events = log.events
        vec_iter = (e[1] for e in events if e[0] == "random")
        imr[b] = next(vec_iter)
        for g in range(G):
            cr[g, b] = next(vec_iter); mr[g, b] = next(vec_iter)
        fc_ints = FlatConsumer([e for e in events if e[0] == "randint"])
        opop, ofit = eo.ea_run_cvrp(locs[b], demand[b], vcap, init[b], G, mutation_rate, crossover_rate, selection_rate,
                                    imr[b], cr[:, b], mr[:, b], fc_ints, top_k=(method == "am"))
        fc_ints.done()
        assert np.array_equal(opop, pop), f"{name}: restatement differs from the reference on instance {b}"
        for key, u in fc_ints.u.items():
            if key[0] == ("init",):
                imu[b, key[1], key[2]] = u
            elif key[0][0] == "cross":
                cu[key[0][1], b, key[1]] = u
            else:
                mu[key[0][1], b, key[1], key[2]] = u
    np.savez_compressed(os.path.join(HERE, name + ".npz"), env_name="cvrp", locs=locs, demand=demand,
                        vehicle_capacity=vcap, init_pop=init, num_generations=G, mutation_rate=mutation_rate,
                        crossover_rate=crossover_rate, selection_rate=selection_rate, top_k=int(method == "am"),
                        init_mut_rand=imr, init_mut_u=imu, cross_rand=cr, cross_u=cu, mut_rand=mr, mut_u=mu,
                        pop=out_pop, fitness=out_fit, torch_version=torch.__version__)
    changed = int((out_pop != init).any(-1).sum())
    print(f"{name}: B={B} S={S} N={N} L={L} G={G} pairs={P}; {changed} of {B * S} individuals changed; "
          f"mean cost {np.mean(2.5 * L - out_fit):.4f}")


def main():
    operator_cases()
    tsp_case("ea_tsp20_default", N=20, B=3, S=20, G=3, mutation_rate=0.1, crossover_rate=0.6, selection_rate=0.2, seed=5)
    tsp_case("ea_tsp20_busy", N=20, B=3, S=20, G=4, mutation_rate=0.6, crossover_rate=0.9, selection_rate=0.7, seed=6)
    tsp_case("ea_tsp50_busy", N=50, B=2, S=50, G=3, mutation_rate=0.5, crossover_rate=0.8, selection_rate=0.5, seed=7)
    cvrp_case("ea_cvrp20_default", N=20, B=3, S=20, G=3, mutation_rate=0.1, crossover_rate=0.6, selection_rate=0.2, seed=8)
    cvrp_case("ea_cvrp20_busy", N=20, B=3, S=20, G=4, mutation_rate=0.7, crossover_rate=0.9, selection_rate=0.7, seed=9)
    cvrp_case("ea_cvrp50_am", N=50, B=2, S=30, G=3, mutation_rate=0.5, crossover_rate=0.8, selection_rate=0.5, seed=10,
              method="am")


if __name__ == "__main__":
    main()
