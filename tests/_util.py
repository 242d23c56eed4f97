"""Shared helpers for the test-suite: golden fixtures + closed-form weights."""
import json
import os

import numpy as np

import goldweights  # tests/golden/goldweights.py (pure numpy)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

with open(os.path.join(GOLDEN, "state_dict_contract.json")) as _f:
    CONTRACT = json.load(_f)
for _name in ("state_dict_contract_sdvrp.json", "state_dict_contract_pctsp.json", "state_dict_contract_op.json",
              "state_dict_contract_cvrptw.json"):      # the sibling envs
    with open(os.path.join(GOLDEN, _name)) as _f:
        CONTRACT.update(json.load(_f))


CONTRACT["am_spctsp"] = CONTRACT["am_pctsp"]      # same policy (embeddings see the expected prize either way)


def golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def golden_weights(cfg):
    """{key: float32 ndarray} for a contract config: am_tsp, am_cvrp, am_sdvrp, pomo_tsp, pomo_cvrp."""
    sd = {}
    for k, shape, dt in CONTRACT[cfg]:
        if dt != "float32":
            continue
        v = goldweights.tensor_for(k, shape)
        if v is not None:
            sd[k] = v
    return sd


def cfg_for(fx):
    pomo = "policy_kw_num_encoder_layers" in fx
    return ("pomo_" if pomo else "am_") + str(fx["env_name"])


def instance_of(fx):
    """The per-instance tensors besides locs, as the oracle / make_td take them: the demand array (CVRP, SDVRP), the
    dict of prize tensors (PCTSP) or None (TSP)."""
    if str(fx["env_name"]) in ("pctsp", "spctsp"):
        return {k: fx[k] for k in ("expected_prize", "real_prize", "penalty", "prize_required")}
    if str(fx["env_name"]) == "op":
        return {k: fx[k] for k in ("prize", "max_length")}
    if str(fx["env_name"]) == "cvrptw":
        return {k: fx[k] for k in ("demand", "time_windows", "durations")}
    return fx.get("demand")


def instance_from_td(env_name, td):
    """The same, from a post-reset (CPU) TensorDict of the package's own envs."""
    keys = {"cvrp": "demand", "sdvrp": "demand", "pctsp": ("expected_prize", "real_prize", "penalty", "prize_required"),
            "spctsp": ("expected_prize", "real_prize", "penalty", "prize_required"),
            "op": ("prize", "max_length"), "cvrptw": ("demand", "time_windows", "durations")}.get(env_name)
    if keys is None:
        return None
    if isinstance(keys, str):
        return td[keys].numpy()
    return {k: td[k].numpy() for k in keys}
