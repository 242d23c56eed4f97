"""TSP / CVRP (and CVRPTW, SDVRP, PCTSP, OP) environments behind the reference's RL4COEnvBase interface, stepping on
MI355X kernels.

Interface mirrored (same names, argument meaning, TensorDict keys / shapes / dtypes and error messages):
  rl4co/envs/common/base.py:19-346        RL4COEnvBase (reset / step / get_reward / get_action_mask / dataset ...)
  rl4co/envs/common/utils.py:21-102       Generator, get_sampler (uniform branch)
  rl4co/envs/routing/tsp/{env,generator}.py, rl4co/envs/routing/cvrp/{env,generator}.py,
  rl4co/envs/routing/{cvrptw,sdvrp,pctsp,op}/{env,generator}.py
Instances are generated on the host with torch's global CPU generator exactly like the reference
(SURVEY Appendix A10), so the same seed gives bit-identical instances; every state transition, mask,
reward and validity check runs in libeamrl_hip.so and requires the TensorDict to live on the GPU.
"""
from __future__ import annotations

import logging
import os
from typing import Iterable, Optional

import numpy as np
import torch

from . import ops
from .tensordict_lite import TensorDict

log = logging.getLogger(__name__)

# Kool et al. (2019) capacities, as tabulated by the reference generator (cvrp/generator.py:17-32)
CAPACITIES = {10: 20.0, 15: 25.0, 20: 30.0, 30: 33.0, 40: 37.0, 50: 40.0, 60: 43.0, 75: 45.0, 100: 50.0,
              125: 55.0, 150: 60.0, 200: 70.0, 500: 100.0, 1000: 150.0}


class UniformSampler:
    """`torch.distributions.Uniform(low, high).sample(shape)` restated: rand * (high - low) + low on the
    CPU with the global generator (envs/common/utils.py:61-64 -> torch Uniform.rsample)."""

    def __init__(self, low: float, high: float):
        self.low, self.high = float(low), float(high)

    def sample(self, shape):
        shape = tuple(int(s) for s in shape)
        rand = torch.rand(shape, dtype=torch.float32)
        low = torch.tensor(self.low, dtype=torch.float32)
        high = torch.tensor(self.high, dtype=torch.float32)
        return low + rand * (high - low)


def get_sampler(val_name, distribution, low=0.0, high=1.0, **kwargs):
    if isinstance(distribution, (int, float)):
        return UniformSampler(distribution, distribution)
    if distribution in ("uniform", "Uniform") or getattr(distribution, "__name__", "") == "Uniform":
        return UniformSampler(low, high)
    if callable(distribution):
        return distribution(**kwargs)
    raise ValueError(f"Invalid distribution type of {distribution}: only uniform samplers are built into "
                     "eam_rl4co_amd (pass `loc_sampler=` for anything else)")


class Generator:
    def __init__(self, **kwargs):
        self.kwargs = kwargs

    def __call__(self, batch_size) -> TensorDict:
        batch_size = [batch_size] if isinstance(batch_size, int) else list(batch_size)
        return self._generate(batch_size)

    def _generate(self, batch_size):
        raise NotImplementedError


class TSPGenerator(Generator):
    """locs [B, num_loc, 2] ~ U[min_loc, max_loc)   (tsp/generator.py:16-62)"""

    def __init__(self, num_loc: int = 20, min_loc: float = 0.0, max_loc: float = 1.0, init_sol_type: str = "random",
                 loc_distribution="uniform", **kwargs):
        self.num_loc, self.min_loc, self.max_loc, self.init_sol_type = num_loc, min_loc, max_loc, init_sol_type
        self.loc_sampler = kwargs.get("loc_sampler") or get_sampler("loc", loc_distribution, min_loc, max_loc, **kwargs)

    def _generate(self, batch_size):
        locs = self.loc_sampler.sample((*batch_size, self.num_loc, 2))
        return TensorDict({"locs": locs}, batch_size=batch_size)


class CVRPGenerator(Generator):
    """locs, depot, demand in {1..9}/capacity, capacity   (cvrp/generator.py:35-147)"""

    def __init__(self, num_loc: int = 20, min_loc: float = 0.0, max_loc: float = 1.0, loc_distribution="uniform",
                 depot_distribution=None, min_demand: int = 1, max_demand: int = 10, demand_distribution="uniform",
                 vehicle_capacity: float = 1.0, capacity: float = None, **kwargs):
        self.num_loc, self.min_loc, self.max_loc = num_loc, min_loc, max_loc
        self.min_demand, self.max_demand, self.vehicle_capacity = min_demand, max_demand, vehicle_capacity
        self.loc_sampler = kwargs.get("loc_sampler") or get_sampler("loc", loc_distribution, min_loc, max_loc, **kwargs)
        self.depot_sampler = kwargs.get("depot_sampler") or (
            get_sampler("depot", depot_distribution, min_loc, max_loc, **kwargs) if depot_distribution is not None else None)
        self.demand_sampler = kwargs.get("demand_sampler") or get_sampler(
            "demand", demand_distribution, min_demand - 1, max_demand - 1, **kwargs)
        if capacity is None:
            capacity = CAPACITIES.get(num_loc)
        if capacity is None:
            closest = min(CAPACITIES, key=lambda x: abs(x - num_loc))
            capacity = CAPACITIES[closest]
            log.warning("capacity for %d locations is not tabulated; using %.1f (table entry for %d)", num_loc,
                        capacity, closest)
        self.capacity = capacity

    def _generate(self, batch_size):
        if self.depot_sampler is not None:
            depot = self.depot_sampler.sample((*batch_size, 2))
            locs = self.loc_sampler.sample((*batch_size, self.num_loc, 2))
        else:  # one draw of num_loc + 1 points, the first is the depot
            pts = self.loc_sampler.sample((*batch_size, self.num_loc + 1, 2))
            depot, locs = pts[..., 0, :], pts[..., 1:, :]
        demand = self.demand_sampler.sample((*batch_size, self.num_loc))
        demand = (demand.int() + 1).float()
        capacity = torch.full((*batch_size, 1), self.capacity)
        return TensorDict({"locs": locs, "depot": depot, "demand": demand / self.capacity, "capacity": capacity},
                          batch_size=batch_size)


class CVRPTWGenerator(CVRPGenerator):
    """CVRP instances on a [0, 150]^2 grid plus integer time windows inside [distance from depot, max_time - distance
    back] and zero service durations (cvrptw/generator.py:14-142).  Same draw order as the reference."""

    def __init__(self, num_loc: int = 20, min_loc: float = 0.0, max_loc: float = 150.0, loc_distribution="uniform",
                 depot_distribution="uniform", min_demand: int = 1, max_demand: int = 10, demand_distribution="uniform",
                 vehicle_capacity: float = 1.0, capacity: float = None, max_time: float = 480, scale: bool = False, **kwargs):
        super().__init__(num_loc=num_loc, min_loc=min_loc, max_loc=max_loc, loc_distribution=loc_distribution,
                         depot_distribution=depot_distribution, min_demand=min_demand, max_demand=max_demand,
                         demand_distribution=demand_distribution, vehicle_capacity=vehicle_capacity, capacity=capacity,
                         **kwargs)
        self.max_loc, self.min_time, self.max_time, self.scale = max_loc, 0.0, max_time, scale

    def _generate(self, batch_size):
        td = super()._generate(batch_size)
        durations = torch.zeros(*batch_size, self.num_loc + 1, dtype=torch.float32)
        dist = (td["depot"] - td["locs"].transpose(0, 1)).norm(p=2, dim=-1).transpose(0, 1)
        dist = torch.cat((torch.zeros(*batch_size, 1), dist), dim=1)
        upper_bound = self.max_time - dist - durations
        ts_1 = torch.rand(*batch_size, self.num_loc + 1)
        ts_2 = torch.rand(*batch_size, self.num_loc + 1)
        min_ts = (dist + (upper_bound - dist) * ts_1).int()
        max_ts = (dist + (upper_bound - dist) * ts_2).int()
        min_times, max_times = torch.min(min_ts, max_ts), torch.max(min_ts, max_ts)
        min_times[..., :, 0] = 0.0
        max_times[..., :, 0] = self.max_time
        mask = min_times == max_times           # empty windows are widened by one unit, downwards first
        if torch.any(mask):
            min_tmp = min_times.clone()
            min_tmp[mask] = torch.max(dist[mask].int(), min_tmp[mask] - 1)
            min_times = min_tmp
            mask = min_times == max_times
            if torch.any(mask):
                max_tmp = max_times.clone()
                max_tmp[mask] = torch.min(torch.floor(upper_bound[mask]).int(),
                                          torch.max(torch.ceil(min_tmp[mask] + durations[mask]).int(), max_tmp[mask] + 1))
                max_times = max_tmp
        if self.scale:
            durations, min_times, max_times = durations / self.max_time, min_times / self.max_time, max_times / self.max_time
            td["depot"] = td["depot"] / self.max_time
            td["locs"] = td["locs"] / self.max_time
        time_windows = torch.stack((min_times, max_times), dim=-1)
        assert torch.all(min_times < max_times), \
            "Please make sure the relation between max_loc and max_time allows for feasible solutions."
        durations[:, 0] = 0.0
        td.update({"durations": durations, "time_windows": time_windows})
        return td


# Kool et al. (2019) expected tour lengths used to scale the PCTSP penalties (pctsp/generator.py:14)
MAX_LENGTHS = {20: 2.0, 50: 3.0, 100: 4.0}


class PCTSPGenerator(Generator):
    """locs, depot, penalty ~ U[0, max_penalty), deterministic_prize ~ U[0, 4/num_loc), stochastic_prize
    (pctsp/generator.py:17-148).  The draws come in the reference's order from torch's global generator."""

    def __init__(self, num_loc: int = 20, min_loc: float = 0.0, max_loc: float = 1.0, loc_distribution="uniform",
                 depot_distribution=None, penalty_factor: float = 3.0, prize_required: float = 1.0, **kwargs):
        self.num_loc, self.min_loc, self.max_loc = num_loc, min_loc, max_loc
        self.penalty_fctor, self.prize_required = penalty_factor, prize_required      # (sic: the reference's attribute name)
        self.loc_sampler = kwargs.get("loc_sampler") or get_sampler("loc", loc_distribution, min_loc, max_loc, **kwargs)
        self.depot_sampler = kwargs.get("depot_sampler") or (
            get_sampler("depot", depot_distribution, min_loc, max_loc, **kwargs) if depot_distribution is not None else None)
        self.deterministic_prize_sampler = get_sampler("deterministric_prize", "uniform", 0.0, 4.0 / self.num_loc)
        self.stochastic_prize_sampler = get_sampler("stochastic_prize", "uniform", 0.0, 2.0)
        self.max_penalty = kwargs.get("max_penalty", None)
        if self.max_penalty is None:
            self.max_penalty = MAX_LENGTHS.get(num_loc, None)
        if self.max_penalty is None:
            closest = min(MAX_LENGTHS, key=lambda x: abs(x - num_loc))
            self.max_penalty = MAX_LENGTHS[closest]
            log.warning("The max penalty for %d locations is not defined. Using the closest max penalty: %s with %d "
                        "locations.", num_loc, self.max_penalty, closest)
        self.max_penalty *= penalty_factor / self.num_loc
        self.penalty_sampler = get_sampler("penalty", "uniform", 0.0, self.max_penalty)

    def _generate(self, batch_size):
        if self.depot_sampler is not None:
            depot = self.depot_sampler.sample((*batch_size, 2))
            locs = self.loc_sampler.sample((*batch_size, self.num_loc, 2))
        else:
            pts = self.loc_sampler.sample((*batch_size, self.num_loc + 1, 2))
            depot, locs = pts[..., 0, :], pts[..., 1:, :]
        penalty = self.penalty_sampler.sample((*batch_size, self.num_loc))
        deterministic_prize = self.deterministic_prize_sampler.sample((*batch_size, self.num_loc))
        stochastic_prize = self.stochastic_prize_sampler.sample((*batch_size, self.num_loc)) * deterministic_prize
        return TensorDict({"locs": locs, "depot": depot, "penalty": penalty, "deterministic_prize": deterministic_prize,
                           "stochastic_prize": stochastic_prize}, batch_size=batch_size)


class OPGenerator(Generator):
    """locs, depot, prize (by default 1 + floor(99 * distance to depot / max distance), in hundredths), max_length
    (op/generator.py:19-147).  Only prize_type in {"dist", "unif", "const"} with the uniform location sampler."""

    def __init__(self, num_loc: int = 20, min_loc: float = 0.0, max_loc: float = 1.0, loc_distribution="uniform",
                 depot_distribution=None, min_prize: float = 1.0, max_prize: float = 1.0, prize_distribution="uniform",
                 prize_type: str = "dist", max_length=None, **kwargs):
        self.num_loc, self.min_loc, self.max_loc = num_loc, min_loc, max_loc
        self.min_prize, self.max_prize, self.prize_type = min_prize, max_prize, prize_type
        self.loc_sampler = kwargs.get("loc_sampler") or get_sampler("loc", loc_distribution, min_loc, max_loc, **kwargs)
        self.depot_sampler = kwargs.get("depot_sampler") or (
            get_sampler("depot", depot_distribution, min_loc, max_loc, **kwargs) if depot_distribution is not None else None)
        # the reference also builds a prize sampler here that _generate never reads (and whose default Uniform(1, 1)
        # recent torch versions reject); it is not needed
        self.max_length = max_length if max_length is not None else MAX_LENGTHS.get(num_loc, None)
        if self.max_length is None:
            closest = min(MAX_LENGTHS, key=lambda x: abs(x - num_loc))
            self.max_length = MAX_LENGTHS[closest]
            log.warning("The max length for %d locations is not defined. Using the closest max length: %s with %d "
                        "locations.", num_loc, self.max_length, closest)

    def _generate(self, batch_size):
        if self.depot_sampler is not None:
            depot = self.depot_sampler.sample((*batch_size, 2))
            locs = self.loc_sampler.sample((*batch_size, self.num_loc, 2))
        else:
            pts = self.loc_sampler.sample((*batch_size, self.num_loc + 1, 2))
            depot, locs = pts[..., 0, :], pts[..., 1:, :]
        locs_with_depot = torch.cat((depot.unsqueeze(1), locs), dim=1)
        if self.prize_type == "const":
            prize = torch.ones(*batch_size, self.num_loc)
        elif self.prize_type == "unif":
            prize = (1 + torch.randint(0, 100, (*batch_size, self.num_loc)).float()) / 100
        elif self.prize_type == "dist":
            prize = (locs_with_depot[..., 0:1, :] - locs_with_depot[..., 1:, :]).norm(p=2, dim=-1)
            prize = (1 + (prize / prize.max(dim=-1, keepdim=True)[0] * 99).int()).float() / 100
        else:
            raise ValueError(f"Invalid prize_type: {self.prize_type}")
        max_length = self.max_length if isinstance(self.max_length, torch.Tensor) else torch.full((*batch_size,),
                                                                                                 self.max_length)
        return TensorDict({"locs": locs_with_depot[..., 1:, :], "depot": locs_with_depot[..., 0, :], "prize": prize,
                           "max_length": max_length}, batch_size=batch_size)


# --------------------------------------------------------------------------------------------------------
class TensorDictDataset(torch.utils.data.Dataset):
    """List-of-dicts dataset with the reference's collate contract (rl4co/data/dataset.py:43-75)."""

    def __init__(self, td):
        self.data_len = td.batch_size[0]
        self.data = [{k: v[i] for k, v in td.items()} for i in range(self.data_len)]

    def __len__(self):
        return self.data_len

    def __getitem__(self, idx):
        return self.data[idx]

    @staticmethod
    def collate_fn(batch):
        return TensorDict({k: torch.stack([b[k] for b in batch]) for k in batch[0].keys()},
                          batch_size=torch.Size([len(batch)]))


def load_npz_to_tensordict(filename):
    """.npz -> TensorDict (rl4co/data/utils.py:11-20).  numpy.load default allow_pickle=False."""
    with np.load(filename) as z:
        d = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
    bs = next(iter(d.values())).shape[0]
    return TensorDict(d, batch_size=bs)


class RL4COEnvBase:
    """Duck-typed stand-in for the reference's RL4COEnvBase (a torchrl EnvBase subclass): the same public
    methods and attributes, without the TorchRL spec machinery the rollout path never touches."""

    name = "base"
    batch_locked = False

    def __init__(self, *, data_dir: str = "data/", train_file: str = None, val_file: str = None, test_file: str = None,
                 val_dataloader_names: list = None, test_dataloader_names: list = None, check_solution: bool = True,
                 dataset_cls: callable = TensorDictDataset, seed: int = None, device: str = "cpu",
                 batch_size=None, run_type_checks: bool = False, allow_done_after_reset: bool = False,
                 _torchrl_mode: bool = False, **kwargs):
        self.device = torch.device(device) if device is not None else None
        self.batch_size = torch.Size(batch_size if batch_size is not None else [])
        self.allow_done_after_reset = allow_done_after_reset
        kwargs.pop("name", None)
        if kwargs:
            log.error("Unused keyword arguments: %s (pass data generation arguments via `generator_params=`)",
                      ", ".join(kwargs.keys()))
        if _torchrl_mode:
            raise NotImplementedError("_torchrl_mode is not supported by eam_rl4co_amd")
        self.data_dir = data_dir
        self.dataset_cls = dataset_cls

        def files(f):
            if f is None:
                return None
            if isinstance(f, Iterable) and not isinstance(f, str):
                return [os.path.join(data_dir, x) for x in f]
            return os.path.join(data_dir, f)

        def names(f, nm):
            if f is not None and isinstance(f, Iterable) and not isinstance(f, str):
                if nm is None:
                    nm = [str(i) for i in range(len(f))]
                assert len(nm) == len(f), "Number of dataloader names must match number of files"
            return nm

        self.train_file = files(train_file)
        self.val_file, self.test_file = files(val_file), files(test_file)
        self.val_dataloader_names = names(self.val_file, val_dataloader_names)
        self.test_dataloader_names = names(self.test_file, test_dataloader_names)
        self.check_solution = check_solution
        if seed is None:
            seed = torch.empty((), dtype=torch.int64).random_().item()
        self.set_seed(seed)

    # ---- episode API -------------------------------------------------------------------------------------------
    def step(self, td) -> dict:
        """{"next": td}; td is updated in place (base.py:122-131)."""
        return {"next": self._step(td)}

    def reset(self, td: Optional[TensorDict] = None, batch_size=None) -> TensorDict:
        if batch_size is None:
            batch_size = self.batch_size if td is None else td.batch_size
        if td is None or td.is_empty():
            td = self.generator(batch_size=batch_size)
        batch_size = [batch_size] if isinstance(batch_size, int) else list(batch_size)
        self.to(td.device)
        td_reset = self._reset(td, batch_size=batch_size)
        dev = td_reset.device
        for key in ("done", "terminated"):  # what TorchRL's EnvBase.reset adds
            if key not in td_reset.keys():
                td_reset.set(key, torch.zeros(*batch_size, 1, dtype=torch.bool, device=dev))
        if not self.allow_done_after_reset:
            td_reset.set("done", torch.zeros_like(td_reset["done"]))
        return td_reset

    def get_reward(self, td, actions, check_solution: Optional[bool] = None) -> torch.Tensor:
        check_solution = self.check_solution if check_solution is None else check_solution
        if check_solution:
            self.check_solution_validity(td, actions)
        return self._get_reward(td, actions)

    def get_action_mask(self, td):
        raise NotImplementedError

    def get_num_starts(self, td):
        n = td["action_mask"].shape[-1]
        return n - 1 if self.name in ("cvrp", "sdvrp", "pctsp", "spctsp", "op", "cvrptw") else n          # depot cannot be a start node (utils/ops.py:120-130)

    def select_start_nodes(self, td, num_starts):
        """POMO start nodes: flat row j = s*B + b starts at node s (+1 with a depot) (utils/ops.py:133-169)."""
        num_loc = getattr(self.generator, "num_loc", 0xFFFFFFFF)
        sel = torch.arange(num_starts, device=td.device).repeat_interleave(td.shape[0]) % num_loc
        if self.name == "op" and bool((td["action_mask"][..., 1:].float().sum(-1) < num_starts).any()):
            # some customers are out of reach from the start: resample among the reachable ones (utils/ops.py:158-169)
            sel = torch.multinomial(td["action_mask"][..., 1:].float(), num_starts, replacement=True) + 1
            return sel.t().reshape(-1)                     # "b n -> (n b)"
        return sel + 1 if self.name in ("cvrp", "sdvrp", "pctsp", "spctsp", "op", "cvrptw") else sel

    def check_solution_validity(self, td, actions) -> None:
        raise NotImplementedError

    def replace_selected_actions(self, cur_actions, new_actions, selection_mask):
        raise NotImplementedError

    def local_search(self, td, actions, **kwargs):
        raise NotImplementedError(f"Local search is out of scope of eam_rl4co_amd ({self.name})")

    # ---- data --------------------------------------------------------------------------------------------------
    def dataset(self, batch_size=[], phase="train", filename=None):
        f = getattr(self, f"{phase}_file") if filename is None else filename
        if f is None:
            if phase != "train":
                log.warning("%s_file not set. Generating dataset instead", phase)
            td = self.generator(batch_size)
        else:
            try:
                if isinstance(f, Iterable) and not isinstance(f, str):
                    nm = getattr(self, f"{phase}_dataloader_names")
                    return {n: self.dataset_cls(self.load_data(_f, batch_size)) for n, _f in zip(nm, f)}
                td = self.load_data(f, batch_size)
            except FileNotFoundError:
                log.error("Provided file name %s not found; generating data instead", f)
                td = self.generator(batch_size)
        return self.dataset_cls(td)

    @staticmethod
    def load_data(fpath, batch_size=[]):
        return load_npz_to_tensordict(fpath)

    def transform(self):
        return self

    def render(self, *args, **kwargs):
        raise NotImplementedError(f"Render is out of scope of eam_rl4co_amd ({self.name})")

    # ---- seeding / device / pickling ---------------------------------------------------------------------------
    def set_seed(self, seed: Optional[int] = None):
        self._set_seed(seed)
        return seed

    def _set_seed(self, seed: Optional[int]):
        self.rng = torch.manual_seed(seed)   # the reference seeds the GLOBAL generator (base.py:300-303)

    def to(self, device):
        if device is not None:
            self.device = torch.device(device)
        return self

    def __getstate__(self):
        state = self.__dict__.copy()
        state["rng"] = state["rng"].get_state()
        return state

    def __setstate__(self, state):
        rng_state = state.pop("rng")
        self.__dict__.update(state)
        self.rng = torch.manual_seed(0)
        self.rng.set_state(rng_state)


def _flat(t, dtype):
    """[B] or [B,1] state tensor -> contiguous [B] view sharing storage (kernels update it in place)."""
    if t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    v = t.reshape(-1)
    if v.data_ptr() != t.data_ptr():
        raise ValueError("state tensor must be contiguous")
    return v


class TSPEnv(RL4COEnvBase):
    """Travelling Salesman Problem (rl4co/envs/routing/tsp/env.py:24-196)."""

    name = "tsp"

    def __init__(self, generator: TSPGenerator = None, generator_params: dict = {}, **kwargs):
        super().__init__(**kwargs)
        self.generator = generator if generator is not None else TSPGenerator(**generator_params)

    def _reset(self, td=None, batch_size=None):
        dev = td.device
        locs = td["locs"]
        n = locs.shape[-2]
        cur = torch.zeros(batch_size, dtype=torch.int64, device=dev)
        return TensorDict({
            "locs": locs,
            "first_node": cur,
            "current_node": cur.clone(),
            "i": torch.zeros(*batch_size, 1, dtype=torch.int64, device=dev),
            "action_mask": torch.ones(*batch_size, n, dtype=torch.bool, device=dev),
            "reward": torch.zeros(*batch_size, 1, dtype=torch.float32, device=dev),
        }, batch_size=batch_size)

    def _step(self, td):
        mask = td["action_mask"]
        if not mask.is_contiguous():
            mask = mask.contiguous()
        action = td["action"]
        if td["first_node"].data_ptr() == td["current_node"].data_ptr():
            td.set("current_node", td["current_node"].clone())   # reset hands out aliased tensors in the reference
        done = _flat(td["done"], torch.bool)
        ops.tsp_step_(mask, _flat(td["first_node"], torch.int64), _flat(td["current_node"], torch.int64),
                      _flat(td["i"], torch.int64), action.reshape(-1).contiguous(), done)
        # shapes after a step follow the reference: done [B], reward = zeros_like(done)
        td.update({"action_mask": mask, "done": done, "reward": torch.zeros_like(done)})
        return td

    def get_action_mask(self, td):
        return td["action_mask"]

    def _get_reward(self, td, actions):
        return ops.tour_length_reward(td["locs"].contiguous(), actions.contiguous(), with_depot=False)

    def check_solution_validity(self, td, actions) -> None:
        bad = ops.check_solution("tsp", actions.contiguous())
        assert int(bad[0]) == 0, "Invalid tour"

    def replace_selected_actions(self, cur_actions, new_actions, selection_mask):
        cur_actions[selection_mask] = new_actions[selection_mask]
        return cur_actions


class CVRPEnv(RL4COEnvBase):
    """Capacitated Vehicle Routing Problem (rl4co/envs/routing/cvrp/env.py:24-264)."""

    name = "cvrp"

    def __init__(self, generator: CVRPGenerator = None, generator_params: dict = {}, **kwargs):
        super().__init__(**kwargs)
        self.generator = generator if generator is not None else CVRPGenerator(**generator_params)

    def _reset(self, td=None, batch_size=None):
        dev = td.device
        n = td["locs"].shape[-2]
        demand = td["demand"]
        vcap = torch.full((*batch_size, 1), self.generator.vehicle_capacity, dtype=torch.float32, device=dev)
        # Reset-state mask in closed form (visited = 0, used = 0, vehicle at the depot): a customer is feasible
        # iff its demand alone fits; the depot is infeasible while any customer is (cvrp/env.py:132-144).
        fits = ~((demand + 0.0) > (vcap + 1e-5))
        depot_ok = ~fits.any(-1, keepdim=True)
        return TensorDict({
            "locs": torch.cat((td["depot"][..., None, :], td["locs"]), -2),
            "demand": demand,
            "current_node": torch.zeros(*batch_size, 1, dtype=torch.int64, device=dev),
            "used_capacity": torch.zeros(*batch_size, 1, dtype=torch.float32, device=dev),
            "vehicle_capacity": vcap,
            "visited": torch.zeros(*batch_size, n + 1, dtype=torch.uint8, device=dev),
            "action_mask": torch.cat((depot_ok, fits), -1),
        }, batch_size=batch_size)

    def _step(self, td):
        mask = td["action_mask"]
        if not mask.is_contiguous():
            mask = mask.contiguous()
        done = _flat(td["done"], torch.bool)
        ops.cvrp_step_mask_(td["visited"], _flat(td["used_capacity"], torch.float32),
                            _flat(td["vehicle_capacity"], torch.float32), td["demand"].contiguous(),
                            _flat(td["current_node"], torch.int64), td["action"].reshape(-1).contiguous(), mask, done)
        td.update({"action_mask": mask, "done": done, "reward": torch.zeros_like(done)})
        return td

    def get_action_mask(self, td):
        mask = torch.empty(td["visited"].shape, dtype=torch.bool, device=td["visited"].device)
        ops.cvrp_mask_(td["visited"], _flat(td["used_capacity"], torch.float32),
                       _flat(td["vehicle_capacity"], torch.float32), td["demand"].contiguous(),
                       _flat(td["current_node"], torch.int64), mask)
        return mask

    def _get_reward(self, td, actions):
        return ops.tour_length_reward(td["locs"].contiguous(), actions.contiguous(), with_depot=True)

    def check_solution_validity(self, td, actions) -> None:
        bad = ops.check_solution("cvrp", actions.contiguous(), td["demand"].contiguous(), td["vehicle_capacity"]).tolist()
        assert bad[0] == 0, "Invalid tour"
        assert bad[1] == 0, "Used more than capacity"

    @staticmethod
    def load_data(fpath, batch_size=[]):
        """demand is stored unnormalised in the .npz wire format (cvrp/env.py:187-194)."""
        td = load_npz_to_tensordict(fpath)
        td.set("demand", td["demand"] / td["capacity"][:, None])
        return td

    def replace_selected_actions(self, cur_actions, new_actions, selection_mask):
        diff = cur_actions.size(-1) - new_actions.size(-1)
        if diff > 0:
            new_actions = torch.nn.functional.pad(new_actions, (0, diff, 0, 0), value=0)
        elif diff < 0:
            cur_actions = torch.nn.functional.pad(cur_actions, (0, -diff, 0, 0), value=0)
        cur_actions[selection_mask] = new_actions[selection_mask]
        return cur_actions


class SDVRPEnv(CVRPEnv):
    """Split Delivery VRP (rl4co/envs/routing/sdvrp/env.py:17-200): CVRP instances whose customers may be served in
    several visits; the state keeps the remaining demand (`demand_with_depot`) instead of a visited set."""

    name = "sdvrp"

    def _reset(self, td=None, batch_size=None):
        dev = td.device
        demand = td["demand"]
        rem = torch.cat((torch.zeros_like(demand[..., 0:1]), demand), -1).contiguous()
        out = TensorDict({
            "locs": torch.cat((td["depot"][..., None, :], td["locs"]), -2),
            "demand": demand,
            "demand_with_depot": rem,
            "current_node": torch.zeros(*batch_size, 1, dtype=torch.int64, device=dev),
            "used_capacity": torch.zeros(*batch_size, 1, dtype=torch.float32, device=dev),
            "vehicle_capacity": torch.full((*batch_size, 1), self.generator.vehicle_capacity, dtype=torch.float32,
                                           device=dev),
        }, batch_size=batch_size)
        # Reset-state mask in closed form (vehicle empty, at the depot): a customer is feasible iff it has demand and
        # the capacity is positive; the depot is infeasible while any customer is (sdvrp/env.py:137-146).
        free = ~((rem[..., 1:] == 0) | (out["used_capacity"] >= out["vehicle_capacity"]))
        out.set("action_mask", torch.cat((~free.any(-1, keepdim=True), free), -1))
        return out

    def _step(self, td):
        mask = td["action_mask"]
        if not mask.is_contiguous():
            mask = mask.contiguous()
        done = _flat(td["done"], torch.bool)
        rem = td["demand_with_depot"]
        ops.sdvrp_step_mask_(rem, _flat(td["used_capacity"], torch.float32), _flat(td["vehicle_capacity"], torch.float32),
                             _flat(td["current_node"], torch.int64), td["action"].reshape(-1).contiguous(), mask, done)
        td.update({"action_mask": mask, "done": done, "reward": torch.zeros_like(done)})
        return td

    def get_action_mask(self, td):
        rem = td["demand_with_depot"]
        mask = torch.empty(rem.shape, dtype=torch.bool, device=rem.device)
        ops.sdvrp_step_mask_(rem.contiguous(), _flat(td["used_capacity"], torch.float32),
                             _flat(td["vehicle_capacity"], torch.float32), _flat(td["current_node"], torch.int64), None,
                             mask)
        return mask

    def check_solution_validity(self, td, actions) -> None:
        bad = ops.check_solution("sdvrp", actions.contiguous(), td["demand"].contiguous(), td["vehicle_capacity"]).tolist()
        assert bad[1] == 0, "Cannot visit depot twice if any nonzero demand"
        assert bad[0] == 0, "All demand must be satisfied"


class CVRPTWEnv(CVRPEnv):
    """CVRP with time windows (rl4co/envs/routing/cvrptw/env.py:26-330): the vehicle's clock advances by travel, waiting
    and service; a customer is feasible only if it can be reached before its window closes; the depot resets the clock."""

    name = "cvrptw"

    def __init__(self, generator: CVRPTWGenerator = None, generator_params: dict = {}, **kwargs):
        RL4COEnvBase.__init__(self, **kwargs)
        self.generator = generator if generator is not None else CVRPTWGenerator(**generator_params)

    @staticmethod
    def _tw_f32(td):
        """The kernels take the windows as float32 (the reference's int32 windows convert exactly)."""
        return td["time_windows"].to(torch.float32).contiguous(), td["durations"].to(torch.float32).contiguous()

    def _reset(self, td=None, batch_size=None):
        dev = td.device
        n = td["locs"].shape[-2]
        locs = torch.cat((td["depot"][..., None, :], td["locs"]), -2)
        demand = td["demand"]
        vcap = torch.full((*batch_size, 1), self.generator.vehicle_capacity, dtype=torch.float32, device=dev)
        out = TensorDict({
            "locs": locs,
            "demand": demand,
            "current_node": torch.zeros(*batch_size, 1, dtype=torch.int64, device=dev),
            "current_time": torch.zeros(*batch_size, 1, dtype=torch.float32, device=dev),
            "used_capacity": torch.zeros(*batch_size, 1, dtype=torch.float32, device=dev),
            "vehicle_capacity": vcap,
            "visited": torch.zeros(*batch_size, n + 1, dtype=torch.uint8, device=dev),
            "durations": td["durations"],
            "time_windows": td["time_windows"],
        }, batch_size=batch_size)
        # reset-state mask in closed form: the CVRP rule (cvrp/env.py:132-144) AND "reachable from the depot in time"
        fits = ~((demand + 0.0) > (vcap + 1e-5))
        cvrp_mask = torch.cat((~fits.any(-1, keepdim=True), fits), -1)
        dist = (locs[..., 0:1, :] - locs).norm(p=2, dim=-1)
        out.set("action_mask", cvrp_mask & (0.0 + dist <= td["time_windows"][..., 1]))
        return out

    def _step(self, td):
        mask = td["action_mask"]
        if not mask.is_contiguous():
            mask = mask.contiguous()
        done = _flat(td["done"], torch.bool)
        tw, dur = self._tw_f32(td)
        ops.cvrptw_step_mask_(td["visited"], _flat(td["used_capacity"], torch.float32),
                              _flat(td["vehicle_capacity"], torch.float32), td["demand"].contiguous(),
                              _flat(td["current_node"], torch.int64), _flat(td["current_time"], torch.float32),
                              td["locs"].contiguous(), tw, dur, td["action"].reshape(-1).contiguous(), mask, done)
        td.update({"action_mask": mask, "done": done, "reward": torch.zeros_like(done)})
        return td

    def get_action_mask(self, td):
        mask = torch.empty(td["visited"].shape, dtype=torch.bool, device=td["visited"].device)
        tw, dur = self._tw_f32(td)
        ops.cvrptw_step_mask_(td["visited"], _flat(td["used_capacity"], torch.float32),
                              _flat(td["vehicle_capacity"], torch.float32), td["demand"].contiguous(),
                              _flat(td["current_node"], torch.int64), _flat(td["current_time"], torch.float32),
                              td["locs"].contiguous(), tw, dur, None, mask)
        return mask

    def check_solution_validity(self, td, actions) -> None:
        super().check_solution_validity(td, actions)          # CVRP: tours and capacity
        tw, dur = self._tw_f32(td)
        late = ops.cvrptw_check_time(actions.contiguous(), td["locs"].contiguous(), tw, dur).tolist()
        assert late[0] == 0, "vehicle cannot start service before deadline"

    @staticmethod
    def load_data(fpath, batch_size=[]):
        return load_npz_to_tensordict(fpath)


class PCTSPEnv(RL4COEnvBase):
    """Prize Collecting TSP (rl4co/envs/routing/pctsp/env.py:21-260): visit customers until the collected prize
    reaches 1 (or everyone is visited), then return to the depot; reward = saved penalties - (length + all penalties)."""

    name = "pctsp"
    _stochastic = False

    def __init__(self, generator: PCTSPGenerator = None, generator_params: dict = {}, **kwargs):
        super().__init__(**kwargs)
        self.generator = generator if generator is not None else PCTSPGenerator(**generator_params)

    @property
    def stochastic(self):
        return self._stochastic

    def _reset(self, td=None, batch_size=None):
        dev = td.device
        real_prize = td["stochastic_prize"] if self.stochastic else td["deterministic_prize"]
        penalty = td["penalty"]
        n = penalty.shape[-1]
        zero = torch.zeros_like(penalty[..., :1])
        visited = torch.zeros(*batch_size, n + 1, dtype=torch.bool, device=dev)
        out = TensorDict({
            "locs": torch.cat((td["depot"][..., None, :], td["locs"]), -2),
            "current_node": torch.zeros(*batch_size, dtype=torch.int64, device=dev),
            "expected_prize": td["deterministic_prize"],
            "real_prize": torch.cat((zero, real_prize), -1),
            "penalty": torch.cat((zero, penalty), -1),
            "cur_total_prize": torch.zeros(*batch_size, dtype=torch.float32, device=dev),
            "cur_total_penalty": penalty.sum(-1),          # all penalties while nothing is visited
            "visited": visited,
            "prize_required": torch.full((*batch_size,), self.generator.prize_required, dtype=torch.float32, device=dev),
            "i": torch.zeros(*batch_size, dtype=torch.int64, device=dev),
        }, batch_size=batch_size)
        # reset-state mask in closed form (nothing visited, no prize yet): customers free, depot closed (env.py:156-163)
        mask = torch.ones(*batch_size, n + 1, dtype=torch.bool, device=dev)
        mask[..., 0] = not (n > 0)
        out.set("action_mask", mask)
        return out

    def _step(self, td):
        mask = td["action_mask"]
        if not mask.is_contiguous():
            mask = mask.contiguous()
        done = _flat(td["done"], torch.bool)
        ops.pctsp_step_mask_(td["visited"], _flat(td["cur_total_prize"], torch.float32),
                             _flat(td["cur_total_penalty"], torch.float32), td["real_prize"].contiguous(),
                             td["penalty"].contiguous(), _flat(td["current_node"], torch.int64),
                             _flat(td["i"], torch.int64), td["action"].reshape(-1).contiguous(), mask, done)
        td.update({"action_mask": mask, "done": done, "reward": torch.zeros_like(done)})
        return td

    def get_action_mask(self, td):
        mask = torch.empty(td["visited"].shape, dtype=torch.bool, device=td["visited"].device)
        ops.pctsp_step_mask_(td["visited"], _flat(td["cur_total_prize"], torch.float32), None, None, None, None, None, None,
                             mask)
        return mask

    def _get_reward(self, td, actions):
        if actions.size(-1) == 1:       # all tours return to the depot at once (env.py:168-171)
            assert bool((actions == 0).all()), "If all length 1 tours, they should be zero"
            return torch.zeros(actions.size(0), dtype=torch.float32, device=actions.device)
        return ops.pctsp_reward(td["locs"].contiguous(), td["penalty"].contiguous(), actions.contiguous())

    def check_solution_validity(self, td, actions) -> None:
        bad = ops.check_solution("pctsp", actions.contiguous(), td["real_prize"].contiguous()).tolist()
        assert bad[0] == 0, "Duplicates"
        assert bad[1] == 0, "Total prize does not satisfy min total prize"


class SPCTSPEnv(PCTSPEnv):
    """Stochastic PCTSP (rl4co/envs/routing/spctsp/env.py:8-33): as PCTSP, but the prize collected at a node is the
    stochastic one while the policy only sees the expected prize."""

    name = "spctsp"
    _stochastic = True


class OPEnv(RL4COEnvBase):
    """Orienteering Problem (rl4co/envs/routing/op/env.py:24-267): collect as much prize as possible on a tour from and
    to the depot no longer than max_length; a customer stays feasible while it can be reached AND the depot afterwards."""

    name = "op"

    def __init__(self, generator: OPGenerator = None, generator_params: dict = {}, prize_type: str = "dist", **kwargs):
        super().__init__(**kwargs)
        self.generator = generator if generator is not None else OPGenerator(**generator_params)
        self.prize_type = prize_type
        assert self.prize_type in ["dist", "unif", "const"], f"Invalid prize_type: {self.prize_type}"

    def _reset(self, td=None, batch_size=None):
        dev = td.device
        locs = torch.cat((td["depot"][:, None, :], td["locs"]), -2)
        # per-node arrival limit: max length minus the way back to the depot, minus an epsilon (env.py:122-126);
        # instance preparation, evaluated with the reference's own torch expression
        max_length = td["max_length"][..., None] - (td["depot"][..., None, :] - locs).norm(p=2, dim=-1) - 1e-6
        out = TensorDict({
            "locs": locs,
            "prize": torch.nn.functional.pad(td["prize"], (1, 0), mode="constant", value=0),
            "tour_length": torch.zeros(*batch_size, dtype=torch.float32, device=dev),
            "max_length": max_length,
            "current_node": torch.zeros(*batch_size, 1, dtype=torch.int64, device=dev),
            "visited": torch.zeros(*batch_size, locs.shape[-2], dtype=torch.bool, device=dev),
            "current_total_prize": torch.zeros(*batch_size, dtype=torch.float32, device=dev),
            "i": torch.zeros(*batch_size, dtype=torch.int64, device=dev),
        }, batch_size=batch_size)
        # reset-state mask in closed form (at the depot, nothing visited, length 0)  (env.py:149-165)
        exceeds = (locs - locs[..., 0:1, :]).norm(p=2, dim=-1) > max_length
        mask = ~exceeds
        mask[..., 0] = True
        out.set("action_mask", mask)
        return out

    def _step(self, td):
        mask = td["action_mask"]
        if not mask.is_contiguous():
            mask = mask.contiguous()
        done = _flat(td["done"], torch.bool)
        ops.op_step_mask_(td["visited"], _flat(td["tour_length"], torch.float32),
                          _flat(td["current_total_prize"], torch.float32), td["prize"].contiguous(), td["locs"].contiguous(),
                          td["max_length"].contiguous(), _flat(td["current_node"], torch.int64), _flat(td["i"], torch.int64),
                          td["action"].reshape(-1).contiguous(), mask, done)
        td.update({"action_mask": mask, "done": done, "reward": torch.zeros_like(done)})
        return td

    def get_action_mask(self, td):
        mask = torch.empty(td["visited"].shape, dtype=torch.bool, device=td["visited"].device)
        ops.op_step_mask_(td["visited"], _flat(td["tour_length"], torch.float32), None, None, td["locs"].contiguous(),
                          td["max_length"].contiguous(), _flat(td["current_node"], torch.int64), None, None, mask)
        return mask

    def _get_reward(self, td, actions):
        if actions.size(-1) == 1:       # all tours return to the depot at once (env.py:169-172)
            assert bool((actions == 0).all()), "If all length 1 tours, they should be zero"
            return torch.zeros(actions.size(0), dtype=torch.float32, device=actions.device)
        return ops.op_reward(td["prize"].contiguous(), actions.contiguous())

    def check_solution_validity(self, td, actions, add_distance_to_depot: bool = True) -> None:
        if not add_distance_to_depot:
            raise NotImplementedError("add_distance_to_depot=False is not built for MI355X")
        bad = ops.op_check_solution(actions.contiguous(), td["locs"].contiguous(), td["max_length"].contiguous()).tolist()
        assert bad[0] == 0, "Duplicates"
        assert bad[1] == 0, "Max length exceeded"


ENV_REGISTRY = {"tsp": TSPEnv, "cvrp": CVRPEnv, "sdvrp": SDVRPEnv, "pctsp": PCTSPEnv, "spctsp": SPCTSPEnv, "op": OPEnv, "cvrptw": CVRPTWEnv}


def get_env(env_name: str, *args, **kwargs) -> RL4COEnvBase:
    cls = ENV_REGISTRY.get(env_name)
    if cls is None:
        raise ValueError(f"Unknown environment {env_name}. Available environments: {list(ENV_REGISTRY)} "
                         "(only the TSP / CVRP / CVRPTW / SDVRP / PCTSP / OP rollout path is built for MI355X)")
    return cls(*args, **kwargs)
