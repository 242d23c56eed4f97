"""Host side of the fork's evolutionary improvement step, mirroring
rl4co/models/zoo/earl/evolution.py: `EA` (:125-354) and `evolution_worker` (:28-123).

The reference moves the sampled tours to the CPU, spreads the instances over a thread pool and runs numba
operators per instance, every training step.  Here the whole batch is one launch of `eamrl_ea_tsp_run`
(csrc/evolution.hip): one workgroup per instance keeps its population in LDS for all generations, and the
tours never leave the GPU.

Randomness: the reference draws inside the operators from numba's per-thread `np.random`, so it is not
reproducible; here the draws are explicit tensors (`EADraws`), generated on the device from a `torch.Generator`
or supplied by the caller -- the form in which the operators are tested against the reference's own outputs
(tests/golden/ea_*.npz).

Built: TSP (order crossover, inversion mutation, elitism, per-start-node / top-k replacement, single-start
rotation population), CVRP (route-prefix crossover with capacity repair, in-route inversion, initial
mutation pass), PCTSP (cycle crossover with prize top-up, prefix inversion) and OP (budget-checked rebuild and
inversion) -- csrc/evolution_prize.hip for the last two.  FFSP (not a routing env of this path) raises
NotImplementedError.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from . import ops
from .utils import unbatchify

SINGLE_START_POP_SIZE = 50       # generate_batch_population: pop_size = nb.int64(50)  (evolution.py:1614)


@dataclass
class EADraws:
    """Random inputs of one EA.run over a batch: see include/eamrl.h (eamrl_ea_tsp_run)."""
    cross_rand: torch.Tensor     # [G, B, P]    float64 uniforms
    cross_idx: torch.Tensor      # [G, B, P, 2] int32 in [1, N)
    mut_rand: torch.Tensor       # [G, B, O]    float64 uniforms
    mut_idx: torch.Tensor        # [G, B, O, 2] int32 in [1, N)

    @staticmethod
    def sample(G, B, S, N, selection_rate, device, generator=None):
        P = ops.ea_num_pairs(selection_rate, S)
        kw = dict(device=device, generator=generator)
        return EADraws(
            torch.rand(G, B, P, dtype=torch.float64, **kw),
            torch.randint(1, max(N, 2), (G, B, P, 2), dtype=torch.int32, **kw),
            torch.rand(G, B, 2 * P, dtype=torch.float64, **kw),
            torch.randint(1, max(N, 2), (G, B, 2 * P, 2), dtype=torch.int32, **kw))

    def to(self, device):
        return EADraws(*(t.to(device).contiguous() for t in (self.cross_rand, self.cross_idx, self.mut_rand, self.mut_idx)))


@dataclass
class EACvrpDraws:
    """Random inputs of one CVRP EA.run over a batch (eamrl_ea_cvrp_run): uniforms in [0, 1) everywhere; integer
    choices are lo + floor(u * (hi - lo)) inside the kernel because their ranges depend on the evolving tours."""
    init_mut_rand: torch.Tensor  # [B, S]
    init_mut_u: torch.Tensor     # [B, S, 3]
    cross_rand: torch.Tensor     # [G, B, P]
    cross_u: torch.Tensor        # [G, B, P]
    mut_rand: torch.Tensor       # [G, B, O]
    mut_u: torch.Tensor          # [G, B, O, 3]

    @staticmethod
    def sample(G, B, S, selection_rate, device, generator=None):
        P = ops.ea_num_pairs(selection_rate, S)
        r = lambda *shape: torch.rand(*shape, dtype=torch.float64, device=device, generator=generator)
        return EACvrpDraws(r(B, S), r(B, S, 3), r(G, B, P), r(G, B, P), r(G, B, 2 * P), r(G, B, 2 * P, 3))

    def to(self, device):
        return EACvrpDraws(*(t.to(device).contiguous() for t in (self.init_mut_rand, self.init_mut_u, self.cross_rand,
                                                                 self.cross_u, self.mut_rand, self.mut_u)))


@dataclass
class EAPrizeDraws:
    """Random inputs of one PCTSP / OP EA.run over a batch (eamrl_ea_prize_run): uniforms in [0, 1)."""
    init_mut_rand: torch.Tensor  # [B, S]
    init_mut_u: torch.Tensor     # [B, S, 2]
    cross_rand: torch.Tensor     # [G, B, P]
    cross_u: torch.Tensor        # [G, B, P]   (OP: the cut point; PCTSP: unused)
    mut_rand: torch.Tensor       # [G, B, O]
    mut_u: torch.Tensor          # [G, B, O, 2]

    @staticmethod
    def sample(G, B, S, selection_rate, device, generator=None):
        P = ops.ea_num_pairs(selection_rate, S)
        r = lambda *shape: torch.rand(*shape, dtype=torch.float64, device=device, generator=generator)
        return EAPrizeDraws(r(B, S), r(B, S, 2), r(G, B, P), r(G, B, P), r(G, B, 2 * P), r(G, B, 2 * P, 2))

    def to(self, device):
        return EAPrizeDraws(*(t.to(device).contiguous() for t in (self.init_mut_rand, self.init_mut_u, self.cross_rand,
                                                                  self.cross_u, self.mut_rand, self.mut_u)))


def generate_batch_population(routes: torch.Tensor, env_code: int = 1, pop_size: int = SINGLE_START_POP_SIZE):
    """[B, N] single tours -> [B, pop_size, N] (evolution.py:1574-1626): TSP (env_code 1) rotations of the tour --
    member 0 is the tour, member i starts at position i % N (position 1 where that is 0); other codes: copies."""
    B, N = routes.shape
    if env_code != 1:
        return routes[:, None, :].expand(B, pop_size, N).contiguous()
    start = torch.arange(pop_size, device=routes.device) % N
    start = torch.where((start == 0) & (torch.arange(pop_size, device=routes.device) > 0), torch.ones_like(start), start)
    idx = (start[:, None] + torch.arange(N, device=routes.device)[None, :]) % N          # [pop, N]
    return routes[:, idx]


class EA:
    """Evolutionary algorithm runner with the reference's constructor contract (evolution.py:129-173)."""

    def __init__(self, env, kwargs: dict):
        self.env = env
        self.num_generations = kwargs.get("num_generations")
        self.mutation_rate = kwargs.get("mutation_rate")
        self.crossover_rate = kwargs.get("crossover_rate")
        self.selection_rate = kwargs.get("selection_rate")
        self.method = kwargs.get("method", None)
        self.env_name = env.name
        if self.env_name not in ("tsp", "cvrp", "pctsp", "op"):
            if self.env_name == "ffsp":
                raise NotImplementedError("EA operators for ffsp are not built (TSP, CVRP, PCTSP and OP only)")
            raise ValueError(f"Unsupported env for EA operators: {self.env_name}")
        assert self.num_generations is not None, "Number of generations must be specified"
        assert self.mutation_rate is not None, "Mutation rate must be specified"
        assert self.crossover_rate is not None, "Crossover rate must be specified"
        assert self.selection_rate is not None, "Selection rate must be specified"

    # -- fitness ----------------------------------------------------------------------------------------------
    def get_cost(self, pop: torch.Tensor, td):
        """cost = -reward of tours pop [B, S, N] (or [S, N] for a batch of one) on td's instances."""
        if pop.dim() == 2:
            pop = pop[None]
        B, S, N = pop.shape
        rows = pop.permute(1, 0, 2).reshape(S * B, N).contiguous()      # (s b) order: row r reads instance r % B
        if self.env_name == "pctsp":
            reward = ops.pctsp_reward(td["locs"].contiguous(), td["penalty"].contiguous(), rows)
        elif self.env_name == "op":
            reward = ops.op_reward(td["prize"].contiguous(), rows)
        else:
            reward = ops.tour_length_reward(td["locs"].contiguous(), rows, with_depot=self.env_name == "cvrp")
        return -reward.view(S, B).t()

    def get_fitness(self, pop, td):
        size = pop.shape[-1]                                             # chromosome length, as the reference
        worst = np.float32({"tsp": 1.5 * size, "op": 0.0}.get(self.env_name, 2.5 * size))
        return torch.tensor(worst, device=pop.device) - self.get_cost(pop, td)

    # -- the run ----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def run(self, init_pop: torch.Tensor, td, draws: EADraws = None, generator=None):
        """init_pop [B, S, N] int64 on the GPU (not modified) -> (pop [B, S, N], fitness [B, S])."""
        squeeze = init_pop.dim() == 2
        pop = (init_pop[None] if squeeze else init_pop).to(torch.int64).contiguous().clone()
        B, S, N = pop.shape
        G = int(self.num_generations)
        if self.env_name == "cvrp":
            draws = (EACvrpDraws.sample(G, B, S, self.selection_rate, pop.device, generator) if draws is None
                     else draws.to(pop.device))
            fit = ops.ea_cvrp_run_(td["locs"].contiguous(), td["demand"].contiguous(),
                                   td["vehicle_capacity"].reshape(B).to(torch.float32).contiguous(), pop, G,
                                   float(self.mutation_rate), float(self.crossover_rate), float(self.selection_rate),
                                   self.method == "am", draws.init_mut_rand, draws.init_mut_u, draws.cross_rand,
                                   draws.cross_u, draws.mut_rand, draws.mut_u)
            return (pop[0], fit[0]) if squeeze else (pop, fit)
        if self.env_name in ("pctsp", "op"):
            draws = (EAPrizeDraws.sample(G, B, S, self.selection_rate, pop.device, generator) if draws is None
                     else draws.to(pop.device))
            pctsp = self.env_name == "pctsp"
            fit = ops.ea_prize_run_(self.env_name, td["locs"].contiguous(),
                                    td["real_prize" if pctsp else "prize"].to(torch.float32).contiguous(),
                                    td["penalty" if pctsp else "max_length"].to(torch.float32).contiguous(), pop, G,
                                    float(self.mutation_rate), float(self.crossover_rate), float(self.selection_rate),
                                    self.method == "am", draws.init_mut_rand, draws.init_mut_u, draws.cross_rand,
                                    draws.cross_u, draws.mut_rand, draws.mut_u)
            return (pop[0], fit[0]) if squeeze else (pop, fit)
        if draws is None:
            draws = EADraws.sample(G, B, S, N, self.selection_rate, pop.device, generator)
        else:
            draws = draws.to(pop.device)
        rate32 = float(np.float32(self.crossover_rate))      # order_crossover_tsp's signature takes float32
        fit = ops.ea_tsp_run_(td["locs"].contiguous(), pop, G, float(self.mutation_rate), rate32,
                              float(self.selection_rate), draws.cross_rand, draws.cross_idx, draws.mut_rand,
                              draws.mut_idx)
        return (pop[0], fit[0]) if squeeze else (pop, fit)


@torch.no_grad()
def evolution_worker(actions, _td, ea: EA, env, return_population: bool = False, draws: EADraws = None, generator=None):
    """Improve sampled tours by evolution (evolution.py:28-123).  actions [S*B, T] in "(s b)" order (multistart)
    or [B, T] (single start: the population is generated from the tour).  Returns (new_actions, init_td[, population]):
    multistart -> [S*B, T-1] without the start column, as the reference; single start -> [B, T]."""
    batch_size = _td.batch_size[0]
    init_td = _td.clone()
    n_start = 1
    if actions.dim() >= 2 and actions.shape[0] % batch_size == 0 and actions.shape[0] // batch_size > 1:
        n_start = actions.shape[0] // batch_size
    if n_start > 1:
        pop = unbatchify(actions, n_start).contiguous()                     # [B, S, T]
    else:
        pop = generate_batch_population(actions, env_code=1 if ea.env_name == "tsp" else 2)
    new_pop, _ = ea.run(pop, _td, draws=draws, generator=generator)
    if n_start > 1:
        new_actions = new_pop.permute(1, 0, 2).reshape(-1, new_pop.shape[-1])[:, 1:].contiguous()
    else:
        new_actions = new_pop[:, 0].contiguous()
    if return_population:
        return new_actions, init_td, new_pop
    return new_actions, init_td
