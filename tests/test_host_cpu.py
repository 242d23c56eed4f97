"""CPU (no GPU needed): host logic, the C-ABI surface, generators against the goldens, loud failure off-GPU."""
import copy
import os
import pickle
import re

import numpy as np
import pytest
import torch

from _util import CONTRACT, golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_loads_and_exports_every_declared_symbol():
    """libeamrl_hip.so (built by __graft_entry__.build) exports exactly what include/eamrl.h declares."""
    from eam_rl4co_amd import _lib

    with open(os.path.join(ROOT, "include", "eamrl.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(eamrl_[a-z_0-9]+)\s*\(", text))
    assert len(declared) >= 16
    lib = _lib.load()   # binds every prototype; raises if one is missing
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert lib.eamrl_version() == 100
    assert lib.eamrl_debug_set(99, 0) == -1 and b"unknown key" in lib.eamrl_last_error()
    # argument validation happens before any launch: null pointers are rejected on a machine without a GPU too
    assert lib.eamrl_tsp_step(None, None, None, None, None, None, 4, 10, None) == -1
    assert b"eamrl_tsp_step" in lib.eamrl_last_error()


def test_state_dict_contract_matches_reference():
    import eam_rl4co_amd as ea

    cfgs = {"am_tsp": dict(env_name="tsp"), "am_cvrp": dict(env_name="cvrp"),
            "pomo_tsp": dict(env_name="tsp", num_encoder_layers=6, normalization="instance", use_graph_context=False),
            "pomo_cvrp": dict(env_name="cvrp", num_encoder_layers=6, normalization="instance", use_graph_context=False),
            "am_sdvrp": dict(env_name="sdvrp"), "am_pctsp": dict(env_name="pctsp"), "am_op": dict(env_name="op"),
            "am_cvrptw": dict(env_name="cvrptw")}
    for name, kw in cfgs.items():
        sd = ea.AttentionModelPolicy(**kw).state_dict()
        mine = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]
        assert mine == CONTRACT[name], name
    assert sum(p.numel() for p in ea.AttentionModelPolicy(env_name="tsp").parameters()) == 710144
    assert sum(p.numel() for p in ea.AttentionModelPolicy(env_name="cvrp").parameters()) == 694144


@pytest.mark.parametrize("name", ["env_tsp20_random", "env_cvrp20_random", "env_cvrp100_random", "env_sdvrp20_random",
                                  "env_pctsp20_random", "env_spctsp20_random", "env_op20_random", "env_op50_random",
                                  "env_cvrptw20_random", "env_cvrptw50_random"])
def test_generators_reproduce_reference_instances(name):
    """Same torch seed -> bit-identical instances as the reference generators (torch CPU RNG stream)."""
    import eam_rl4co_amd as ea

    fx = golden(name)
    if str(fx["torch_version"]) != torch.__version__:
        pytest.skip("goldens were generated with another torch version (RNG stream may differ)")
    env = ea.get_env(str(fx["env_name"]), generator_params=dict(num_loc=int(fx["num_loc"])), seed=int(fx["data_seed"]))
    torch.manual_seed(int(fx["data_seed"]))
    td = env.generator(batch_size=[fx["gen_locs"].shape[0]])
    for k in ("locs", "depot", "demand", "capacity", "penalty", "deterministic_prize", "stochastic_prize", "prize",
              "max_length", "durations", "time_windows"):
        if "gen_" + k in fx:
            assert np.array_equal(td[k].numpy(), fx["gen_" + k]), k
    td = env.reset(td)      # the reset-state mask is computed in closed form on the host
    assert np.array_equal(td["action_mask"].numpy(), fx["reset_action_mask"])


@pytest.mark.parametrize("name", ["tsp100_greedy", "cvrp100_greedy"])
def test_reset_reproduces_reference_post_reset_td(name):
    import eam_rl4co_amd as ea

    fx = golden(name)
    if str(fx["torch_version"]) != torch.__version__:
        pytest.skip("goldens were generated with another torch version")
    env_name = str(fx["env_name"])
    B, M = fx["locs"].shape[:2]
    env = ea.get_env(env_name, generator_params=dict(num_loc=M - (env_name == "cvrp")), seed=int(fx["data_seed"]))
    torch.manual_seed(int(fx["data_seed"]))
    td = env.reset(batch_size=[B])
    assert np.array_equal(td["locs"].numpy(), fx["locs"])
    # keys / shapes / dtypes of SURVEY Appendix A1
    if env_name == "tsp":
        exp = {"locs": ((B, M, 2), torch.float32), "first_node": ((B,), torch.int64), "current_node": ((B,), torch.int64),
               "i": ((B, 1), torch.int64), "action_mask": ((B, M), torch.bool), "reward": ((B, 1), torch.float32),
               "done": ((B, 1), torch.bool), "terminated": ((B, 1), torch.bool)}
    else:
        assert np.array_equal(td["demand"].numpy(), fx["demand"])
        exp = {"locs": ((B, M, 2), torch.float32), "demand": ((B, M - 1), torch.float32),
               "current_node": ((B, 1), torch.int64), "used_capacity": ((B, 1), torch.float32),
               "vehicle_capacity": ((B, 1), torch.float32), "visited": ((B, M), torch.uint8),
               "action_mask": ((B, M), torch.bool), "done": ((B, 1), torch.bool), "terminated": ((B, 1), torch.bool)}
        assert not td["action_mask"][:, 0].any() and td["action_mask"][:, 1:].all()
    got = {k: (tuple(v.shape), v.dtype) for k, v in td.items()}
    assert got == exp


def test_no_cpu_fallback_fails_loudly():
    import eam_rl4co_amd as ea

    env = ea.get_env("tsp", generator_params=dict(num_loc=10))
    td = env.reset(batch_size=[2])
    pol = ea.AttentionModelPolicy(env_name="tsp").eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pol(td, env, phase="test", decode_type="greedy")
    td.set("action", torch.zeros(2, dtype=torch.int64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        env.step(td)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        env.get_reward(td, torch.zeros(2, 10, dtype=torch.int64), check_solution=False)


def test_unsupported_features_raise():
    import eam_rl4co_amd as ea

    with pytest.raises(NotImplementedError):
        ea.AttentionModelPolicy(env_name="mtsp")
    with pytest.raises(NotImplementedError):
        ea.AttentionModelPolicy(env_name="tsp", normalization="layer")
    with pytest.raises(ValueError):
        ea.get_env("mtsp")


def test_multistart_helpers_follow_reference_layout():
    """get_num_starts / select_start_nodes: flat row j = s*B + b starts at node s (TSP) / s+1 (CVRP)."""
    import eam_rl4co_amd as ea

    for name, off in (("tsp", 0), ("cvrp", 1), ("sdvrp", 1), ("pctsp", 1), ("op", 1), ("cvrptw", 1)):
        env = ea.get_env(name, generator_params=dict(num_loc=7))
        td = env.reset(batch_size=[3])
        assert env.get_num_starts(td) == 7
        sel = env.select_start_nodes(td, 5)
        assert sel.tolist() == [s + off for s in range(5) for _ in range(3)]


def test_env_pickles_and_deepcopies():
    import eam_rl4co_amd as ea

    env = ea.get_env("cvrp", generator_params=dict(num_loc=20), seed=5)
    env2 = pickle.loads(pickle.dumps(env))
    env3 = copy.deepcopy(env)
    assert env2.generator.capacity == env3.generator.capacity == 30.0
    pol = ea.AttentionModelPolicy(env_name="cvrp")
    pol2 = copy.deepcopy(pol)
    assert pol2.state_dict().keys() == pol.state_dict().keys()


def test_tensordict_lite_surface():
    from eam_rl4co_amd.tensordict_lite import _LiteTensorDict as TD

    td = TD({"a": torch.arange(6).reshape(3, 2), "b": torch.zeros(3)}, batch_size=[3])
    assert td.batch_size == torch.Size([3]) and td.shape == torch.Size([3]) and td.dim() == 1 and len(td) == 3
    assert set(td.keys()) == {"a", "b"} and "a" in td and not td.is_empty()
    sub = td[1:]
    assert sub.batch_size == torch.Size([2]) and sub["a"].tolist() == [[2, 3], [4, 5]]
    c = td.clone()
    c["a"][0, 0] = 99
    assert td["a"][0, 0] == 0
    td.update({"c": torch.ones(3, 1)})
    assert td.get("zzz", None) is None and td.exclude("b").keys() == {"a", "c"}
    # batchify / unbatchify as rl4co/utils/ops.py:13-56 does them
    rep = td.expand(4, 3).contiguous().view(12)
    assert rep.batch_size == torch.Size([12]) and rep["a"][3].tolist() == td["a"][0].tolist()
    un = rep.view(4, 3).permute(1, 0)
    assert un.batch_size == torch.Size([3, 4]) and un["a"].shape == (3, 4, 2)
    with pytest.raises(RuntimeError):
        TD({"a": torch.zeros(2)}, batch_size=[3])


def test_cvrp_npz_wire_format(tmp_path):
    """.npz dataset keys of rl4co/data/generate_data.py:38-83; demand is normalised by capacity on load."""
    import eam_rl4co_amd as ea

    f = tmp_path / "vrp.npz"
    np.savez(f, depot=np.random.rand(4, 2).astype(np.float32), locs=np.random.rand(4, 10, 2).astype(np.float32),
             demand=np.random.randint(1, 10, (4, 10)).astype(np.float32), capacity=np.full(4, 20.0, np.float32))
    env = ea.get_env("cvrp", generator_params=dict(num_loc=10), data_dir=str(tmp_path), val_file="vrp.npz")
    ds = env.dataset(phase="val")
    assert len(ds) == 4
    batch = ds.collate_fn([ds[0], ds[1]])
    assert batch["demand"].shape == (2, 10) and float(batch["demand"].max()) <= 9 / 20
    td = env.reset(batch)
    assert td["locs"].shape == (2, 11, 2)
