#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/*.npz by RUNNING THE REFERENCE.

Run only in the build container (the reference never travels to the GPU box):

    python tests/golden/make_golden.py [extra | beam | filtering | sdvrp | pctsp | op | cvrptw | train | eval | inject | augment]

(no argument: the first batch, TSP / CVRP / POMO; `extra`: larger graphs and decoding options; `beam`: beam search;
`filtering`: top-k / top-p; `sdvrp`, `pctsp` (incl. SPCTSP), `op`, `cvrptw`: the sibling envs and their state_dict
contracts.)  The reference modules (TSPEnv, CVRPEnv, AttentionModelPolicy, decoding strategies,
PointerAttention ...) are imported unmodified from /root/reference through the
container stand-ins of `_refshim.py` (tensordict/torchrl are absent; they carry no
arithmetic of this path).  Weights are the closed-form `goldweights` streams, instances
come from the reference generators under a pinned torch seed, and for sampling cases the
Exp(1) noise consumed by `torch.multinomial` is recorded so a
deterministic implementation can replay it.

Each fixture holds inputs (locs/demand, noise), per-step decoder outputs for the steps
listed in `steps_kept`, and the rollout outputs (actions, reward, log-likelihood).
Torch version used is stored in every file.
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import _refshim  # noqa: E402

_refshim.install()

import torch  # noqa: E402

import goldweights  # noqa: E402
import rl4co.utils.decoding as ref_decoding  # noqa: E402
from rl4co.envs.routing.cvrp.env import CVRPEnv  # noqa: E402
from rl4co.envs.routing.cvrptw.env import CVRPTWEnv  # noqa: E402
from rl4co.envs.routing.op.env import OPEnv  # noqa: E402
from rl4co.envs.routing.pctsp.env import PCTSPEnv  # noqa: E402
from rl4co.envs.routing.sdvrp.env import SDVRPEnv  # noqa: E402
from rl4co.envs.routing.spctsp.env import SPCTSPEnv  # noqa: E402
from rl4co.envs.routing.tsp.env import TSPEnv  # noqa: E402
from rl4co.models.zoo.am.policy import AttentionModelPolicy  # noqa: E402

torch.set_num_threads(1)  # results are thread-count invariant; 1 keeps the run reproducible


def make_policy(env_name, **kw):
    pol = AttentionModelPolicy(env_name=env_name, **kw).eval()
    sd = pol.state_dict()
    new = goldweights.fill_state_dict(sd)
    for k, v in new.items():
        sd[k].copy_(torch.from_numpy(v))
    return pol


class Recorder:
    """Records decoder logits / masks, processed logprobs and multinomial noise."""

    def __init__(self, policy):
        self.policy = policy
        self.logits, self.masks, self.logprobs, self.noise, self.starts = [], [], [], [], []

    def __enter__(self):
        dec = self.policy.decoder
        self._dec_fwd = dec.forward

        def fwd(td, cached, num_starts=0):
            logits, mask = self._dec_fwd(td, cached, num_starts)
            self.logits.append(logits.detach().clone())
            self.masks.append(mask.detach().clone())
            return logits, mask

        dec.forward = fwd

        self._pl = ref_decoding.process_logits

        def pl(*a, **k):
            out = self._pl(*a, **k)
            self.logprobs.append(out.detach().clone())
            return out

        ref_decoding.process_logits = pl

        self._mn = torch.multinomial

        def mn(probs, num_samples, *a, **k):
            if num_samples != 1:        # sample_n_random_actions (SamplingEval's random start nodes): record, do not replay
                res = self._mn(probs, num_samples, *a, **k)
                self.starts.append(res.clone())
                return res
            state = torch.get_rng_state()
            res = self._mn(probs, num_samples, *a, **k)
            after = torch.get_rng_state()
            torch.set_rng_state(state)
            q = torch.empty_like(probs).exponential_(1)
            replay = torch.argmax(probs / q, dim=-1, keepdim=True)
            assert torch.equal(replay, res), "multinomial != argmax(p / Exp(1)) replay"
            torch.set_rng_state(after)
            self.noise.append(q.clone())
            return res

        torch.multinomial = mn
        return self

    def __exit__(self, *exc):
        self.policy.decoder.forward = self._dec_fwd
        ref_decoding.process_logits = self._pl
        torch.multinomial = self._mn


def gen_params(env_name, num_loc):
    """OPGenerator's defaults build Uniform(1.0, 1.0), which this torch rejects; prize_distribution="dist" skips that
    sampler (it is unused: the prize comes from prize_type="dist", the distance to the depot)."""
    return dict(num_loc=num_loc, prize_distribution="dist") if env_name == "op" else dict(num_loc=num_loc)


def np_(t):
    return t.detach().cpu().numpy()


def run_case(name, env_name, num_loc, batch, decode_type, policy_kw=None, num_starts=None,
             keep_steps=None, keep_embeds=False, data_seed=1234, sample_seed=4321, actions=None,
             td_init=None, decode_kw=None):
    Env = {"tsp": TSPEnv, "cvrp": CVRPEnv, "sdvrp": SDVRPEnv, "pctsp": PCTSPEnv, "op": OPEnv, "cvrptw": CVRPTWEnv,
           "spctsp": SPCTSPEnv}[env_name]
    env = Env(generator_params=gen_params(env_name, num_loc), seed=data_seed)
    if td_init is None:
        torch.manual_seed(data_seed)
        td_init = env.reset(batch_size=[batch])
    policy = make_policy(env_name, **(policy_kw or {}))
    kw = dict(decode_type=decode_type, **(decode_kw or {}))
    if num_starts is not None:
        kw["num_starts"] = num_starts
    torch.manual_seed(sample_seed)
    with torch.inference_mode(), Recorder(policy) as rec:
        out = policy(td_init.clone(), env, phase="test", return_hidden=keep_embeds,
                     return_init_embeds=keep_embeds, return_sum_log_likelihood=False,
                     actions=actions, **kw)
    T = len(rec.logits)
    steps = list(range(T)) if keep_steps is None else [s for s in keep_steps if s < T]
    fx = {
        "torch_version": np.array(torch.__version__),
        "env_name": np.array(env_name),
        "decode_type": np.array(decode_type if actions is None else "evaluate"),
        "num_starts": np.array(0 if num_starts is None else num_starts, dtype=np.int64),
        "data_seed": np.array(data_seed, dtype=np.int64),
        "locs": np_(td_init["locs"]),
        "actions": np_(out["actions"]),
        "reward": np_(out["reward"]),
        "logp_steps": np_(out["log_likelihood"]),          # [B(*S), T(+1)] per-step selected logp
        "log_likelihood": np_(out["log_likelihood"].sum(1)),
        "steps_kept": np.array(steps, dtype=np.int64),
        "step_logits": np.stack([np_(rec.logits[s]) for s in steps], 1),      # raw decoder logits
        "step_logprobs": np.stack([np_(rec.logprobs[s]) for s in steps], 1),  # after process_logits
        "step_mask": np.stack([np_(rec.masks[s]) for s in steps], 1),
        "n_decoder_steps": np.array(T, dtype=np.int64),
    }
    if env_name in ("cvrp", "sdvrp", "cvrptw"):
        fx["demand"] = np_(td_init["demand"])
        fx["vehicle_capacity"] = np_(td_init["vehicle_capacity"])
    if env_name == "cvrptw":
        for k in ("durations", "time_windows"):
            fx[k] = np_(td_init[k])
    if env_name in ("pctsp", "spctsp"):
        for k in ("expected_prize", "real_prize", "penalty", "prize_required"):
            fx[k] = np_(td_init[k])
    if env_name == "op":
        for k in ("prize", "max_length"):
            fx[k] = np_(td_init[k])
    if rec.noise:
        fx["noise"] = np.stack([np_(q) for q in rec.noise], 1)  # [rows, T, M] Exp(1) draws
    if policy_kw:
        for k, v in policy_kw.items():
            fx["policy_kw_" + k] = np.array(v)
    for k, v in (decode_kw or {}).items():
        fx["decode_kw_" + k] = np.array(v)
    if keep_embeds:
        cache = out["hidden"]  # PrecomputedCache after the decoder's pre_decoder_hook
        fx["init_embeds"] = np_(out["init_embeds"])
        fx["embeddings"] = np_(cache.node_embeddings)
        fx["glimpse_key"] = np_(cache.glimpse_key)
        fx["glimpse_val"] = np_(cache.glimpse_val)
        fx["logit_key"] = np_(cache.logit_key)
        gc = cache.graph_context
        fx["graph_context"] = np_(gc) if isinstance(gc, torch.Tensor) else np.zeros((0,), np.float32)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **fx)
    print(f"{name}: T={T} reward[:3]={fx['reward'][:3]} -> {os.path.getsize(path) / 1024:.0f} KiB")
    return fx, td_init


def run_env_case(name, env_name, num_loc, batch, data_seed=99, act_seed=7):
    """Env-only golden: random feasible policy, every state tensor after every step."""
    Env = {"tsp": TSPEnv, "cvrp": CVRPEnv, "sdvrp": SDVRPEnv, "pctsp": PCTSPEnv, "op": OPEnv, "cvrptw": CVRPTWEnv,
           "spctsp": SPCTSPEnv}[env_name]
    env = Env(generator_params=gen_params(env_name, num_loc), seed=data_seed)
    torch.manual_seed(data_seed)
    gen = env.generator(batch_size=[batch])
    fx = {"torch_version": np.array(torch.__version__), "env_name": np.array(env_name),
          "data_seed": np.array(data_seed, dtype=np.int64), "num_loc": np.array(num_loc, dtype=np.int64)}
    for k, v in gen.items():
        fx["gen_" + k] = np_(v)
    td = env.reset(gen.clone())
    fx["reset_action_mask"] = np_(td["action_mask"])
    torch.manual_seed(act_seed)
    per = {k: [] for k in ("action", "action_mask", "done", "current_node")}
    extra = {"tsp": ("first_node", "i"), "cvrp": ("used_capacity", "visited"),
             "sdvrp": ("used_capacity", "demand_with_depot"),
             "pctsp": ("cur_total_prize", "cur_total_penalty", "visited", "i"),
             "spctsp": ("cur_total_prize", "cur_total_penalty", "visited", "i"),
             "op": ("tour_length", "current_total_prize", "visited", "i"),
             "cvrptw": ("used_capacity", "visited", "current_time")}[env_name]
    for k in extra:
        per[k] = []
    while not td["done"].all():
        td = ref_decoding.random_policy(td)
        td = env.step(td)["next"]
        for k in per:
            per[k].append(np_(td[k]).copy())
    for k, v in per.items():
        fx["step_" + k] = np.stack(v, 1)
    actions = torch.from_numpy(fx["step_action"])
    fx["reward"] = np_(env.get_reward(td, actions))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **fx)
    print(f"{name}: T={actions.shape[1]} reward[:3]={fx['reward'][:3]} -> {os.path.getsize(path) / 1024:.0f} KiB")


def dump_state_dict_contract():
    """Key names / shapes / dtypes of the reference policies' state_dict (the checkpoint contract)."""
    import json
    out = {}
    cfgs = {
        "am_tsp": dict(env_name="tsp"),
        "am_cvrp": dict(env_name="cvrp"),
        "pomo_tsp": dict(env_name="tsp", num_encoder_layers=6, normalization="instance", use_graph_context=False),
        "pomo_cvrp": dict(env_name="cvrp", num_encoder_layers=6, normalization="instance", use_graph_context=False),
    }
    for name, kw in cfgs.items():
        sd = AttentionModelPolicy(**kw).state_dict()
        out[name] = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]
    with open(os.path.join(HERE, "state_dict_contract.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("state_dict_contract.json:", {k: len(v) for k, v in out.items()})


def main():
    first4 = [0, 1, 2, 3]
    dump_state_dict_contract()
    # ---- TSP (configs C1/C2 shapes, scaled-down batch) --------------------------------------
    run_case("tsp20_greedy", "tsp", 20, 4, "greedy", keep_embeds=True)
    fx, td0 = run_case("tsp20_sampling", "tsp", 20, 4, "sampling")
    run_case("tsp20_evaluate", "tsp", 20, 4, "sampling", actions=torch.from_numpy(fx["actions"]), td_init=td0)
    run_case("tsp20_multistart_greedy", "tsp", 20, 4, "multistart_greedy", num_starts=20, keep_steps=first4)
    run_case("tsp100_greedy", "tsp", 100, 16, "greedy", keep_steps=first4 + [50, 99])
    run_case("tsp100_sampling", "tsp", 100, 8, "sampling", keep_steps=first4)
    # ---- CVRP (C3 shapes) -------------------------------------------------------------------
    run_case("cvrp20_greedy", "cvrp", 20, 4, "greedy", keep_embeds=True)
    fx, td0 = run_case("cvrp20_sampling", "cvrp", 20, 4, "sampling")
    run_case("cvrp20_evaluate", "cvrp", 20, 4, "sampling", actions=torch.from_numpy(fx["actions"]), td_init=td0)
    run_case("cvrp20_multistart_greedy", "cvrp", 20, 4, "multistart_greedy", num_starts=20, keep_steps=first4)
    run_case("cvrp100_greedy", "cvrp", 100, 16, "greedy", keep_steps=first4 + [50])
    run_case("cvrp100_sampling", "cvrp", 100, 16, "sampling", keep_steps=first4)
    # ---- POMO policy defaults (C4 shapes): 6 layers, instance norm, no graph context ----------
    pomo = dict(num_encoder_layers=6, normalization="instance", use_graph_context=False)
    run_case("pomo_tsp20_multistart_sampling", "tsp", 20, 4, "multistart_sampling", policy_kw=pomo,
             num_starts=20, keep_steps=first4, keep_embeds=True)
    # ---- env-only (integer / bool state machine) ------------------------------------------------
    run_env_case("env_tsp20_random", "tsp", 20, 8)
    run_env_case("env_cvrp20_random", "cvrp", 20, 8)
    run_env_case("env_cvrp100_random", "cvrp", 100, 4)


def extra():
    """Second batch of fixtures (python make_golden.py extra): graphs above 128 nodes (the streaming decode kernel and
    the key-tiled attention), mid sizes, the POMO policy on CVRP, and non-default temperature / tanh clipping."""
    first4 = [0, 1, 2, 3]
    pomo = dict(num_encoder_layers=6, normalization="instance", use_graph_context=False)
    run_case("tsp50_greedy", "tsp", 50, 8, "greedy", keep_steps=first4, data_seed=50)
    run_case("cvrp50_sampling", "cvrp", 50, 8, "sampling", keep_steps=first4, data_seed=51)
    run_case("tsp200_greedy", "tsp", 200, 2, "greedy", keep_steps=first4 + [100, 199], data_seed=200)
    run_case("cvrp200_greedy", "cvrp", 200, 2, "greedy", keep_steps=first4 + [100], data_seed=201)
    run_case("pomo_cvrp20_multistart_greedy", "cvrp", 20, 4, "multistart_greedy", policy_kw=pomo, num_starts=20,
             keep_steps=first4, data_seed=21)
    run_case("cvrp20_sampling_temp", "cvrp", 20, 4, "sampling", keep_steps=first4, data_seed=22,
             decode_kw=dict(temperature=1.5, tanh_clipping=8.0))
    run_case("tsp20_greedy_noclip", "tsp", 20, 4, "greedy", keep_steps=first4, data_seed=23,
             decode_kw=dict(temperature=0.5, tanh_clipping=0.0))


def beam():
    """Third batch (python make_golden.py beam): decode_type="beam_search" of the reference."""
    first4 = [0, 1, 2, 3]
    run_case("tsp20_beam", "tsp", 20, 4, "beam_search", keep_steps=first4, data_seed=31, decode_kw=dict(select_best=True))
    run_case("tsp20_beam5_all", "tsp", 20, 3, "beam_search", keep_steps=first4, data_seed=32,
             decode_kw=dict(beam_width=5, select_best=False))
    run_case("cvrp20_beam", "cvrp", 20, 4, "beam_search", keep_steps=first4, data_seed=33, decode_kw=dict(select_best=True))
    run_case("tsp50_beam12_all", "tsp", 50, 2, "beam_search", keep_steps=first4, data_seed=34,
             decode_kw=dict(beam_width=12, select_best=False))


def cvrptw():
    """Eighth batch (python make_golden.py cvrptw): CVRP with time windows."""
    import json
    first4 = [0, 1, 2, 3]
    sd = AttentionModelPolicy(env_name="cvrptw").state_dict()
    with open(os.path.join(HERE, "state_dict_contract_cvrptw.json"), "w") as f:
        json.dump({"am_cvrptw": [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]}, f, indent=0)
    run_case("cvrptw20_greedy", "cvrptw", 20, 4, "greedy", keep_embeds=True, data_seed=91)
    run_case("cvrptw20_sampling", "cvrptw", 20, 4, "sampling", keep_steps=first4, data_seed=92)
    run_case("cvrptw50_greedy", "cvrptw", 50, 4, "greedy", keep_steps=first4, data_seed=93)
    run_case("cvrptw100_sampling", "cvrptw", 100, 4, "sampling", keep_steps=first4, data_seed=95)
    run_case("cvrptw20_multistart_greedy", "cvrptw", 20, 3, "multistart_greedy", num_starts=20, keep_steps=first4, data_seed=94)
    run_env_case("env_cvrptw20_random", "cvrptw", 20, 8)
    run_env_case("env_cvrptw50_random", "cvrptw", 50, 4, data_seed=97)


def op():
    """Seventh batch (python make_golden.py op): orienteering problem."""
    import json
    first4 = [0, 1, 2, 3]
    sd = AttentionModelPolicy(env_name="op").state_dict()
    with open(os.path.join(HERE, "state_dict_contract_op.json"), "w") as f:
        json.dump({"am_op": [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]}, f, indent=0)
    run_case("op20_greedy", "op", 20, 4, "greedy", keep_embeds=True, data_seed=81)
    run_case("op20_sampling", "op", 20, 4, "sampling", keep_steps=first4, data_seed=82)
    run_case("op50_greedy", "op", 50, 4, "greedy", keep_steps=first4, data_seed=83)
    run_case("op100_sampling", "op", 100, 4, "sampling", keep_steps=first4, data_seed=85)
    run_case("op20_multistart_greedy", "op", 20, 3, "multistart_greedy", num_starts=20, keep_steps=first4, data_seed=84)
    run_env_case("env_op20_random", "op", 20, 8)
    run_env_case("env_op50_random", "op", 50, 4, data_seed=98)


def pctsp():
    """Sixth batch (python make_golden.py pctsp): prize-collecting TSP."""
    import json
    first4 = [0, 1, 2, 3]
    sd = AttentionModelPolicy(env_name="pctsp").state_dict()
    with open(os.path.join(HERE, "state_dict_contract_pctsp.json"), "w") as f:
        json.dump({"am_pctsp": [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]}, f, indent=0)
    run_case("pctsp20_greedy", "pctsp", 20, 4, "greedy", keep_embeds=True, data_seed=71)
    run_case("pctsp20_sampling", "pctsp", 20, 4, "sampling", keep_steps=first4, data_seed=72)
    run_case("pctsp50_greedy", "pctsp", 50, 4, "greedy", keep_steps=first4, data_seed=73)
    run_case("pctsp100_sampling", "pctsp", 100, 4, "sampling", keep_steps=first4, data_seed=75)
    run_case("pctsp20_multistart_greedy", "pctsp", 20, 3, "multistart_greedy", num_starts=20, keep_steps=first4, data_seed=74)
    run_env_case("env_pctsp20_random", "pctsp", 20, 8)
    run_case("spctsp20_sampling", "spctsp", 20, 4, "sampling", keep_steps=first4, data_seed=77)
    run_case("spctsp50_greedy", "spctsp", 50, 4, "greedy", keep_steps=first4, data_seed=78)
    run_env_case("env_spctsp20_random", "spctsp", 20, 8, data_seed=96)
    run_case("pctsp20_beam", "pctsp", 20, 3, "beam_search", keep_steps=first4, data_seed=76, decode_kw=dict(beam_width=6, select_best=True))


def sdvrp():
    """Fifth batch (python make_golden.py sdvrp): the split-delivery sibling env."""
    import json
    first4 = [0, 1, 2, 3]
    sd = AttentionModelPolicy(env_name="sdvrp").state_dict()
    with open(os.path.join(HERE, "state_dict_contract_sdvrp.json"), "w") as f:
        json.dump({"am_sdvrp": [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]}, f, indent=0)
    run_case("sdvrp20_greedy", "sdvrp", 20, 4, "greedy", keep_embeds=True, data_seed=61)
    run_case("sdvrp20_sampling", "sdvrp", 20, 4, "sampling", keep_steps=first4, data_seed=62)
    run_case("sdvrp50_greedy", "sdvrp", 50, 4, "greedy", keep_steps=first4, data_seed=63)
    run_case("sdvrp20_multistart_greedy", "sdvrp", 20, 3, "multistart_greedy", num_starts=20, keep_steps=first4, data_seed=64)
    run_env_case("env_sdvrp20_random", "sdvrp", 20, 8)
    run_case("sdvrp20_beam", "sdvrp", 20, 3, "beam_search", keep_steps=first4, data_seed=65, decode_kw=dict(beam_width=6, select_best=True))


def filtering():
    """Fourth batch (python make_golden.py filtering): top-k / top-p (nucleus) filtering of process_logits."""
    first4 = [0, 1, 2, 3]
    run_case("tsp20_sampling_topk5", "tsp", 20, 4, "sampling", keep_steps=first4, data_seed=41, decode_kw=dict(top_k=5))
    run_case("tsp20_sampling_topp", "tsp", 20, 4, "sampling", keep_steps=first4, data_seed=42,
             decode_kw=dict(top_p=0.8, temperature=2.0))
    run_case("cvrp20_sampling_topk_topp", "cvrp", 20, 4, "sampling", keep_steps=first4, data_seed=43,
             decode_kw=dict(top_k=6, top_p=0.9, temperature=1.5))
    run_case("tsp100_greedy_topk", "tsp", 100, 2, "greedy", keep_steps=first4, data_seed=44, decode_kw=dict(top_k=3))


def run_train_case(name, env_name, num_loc, batch, policy_kw=None, num_starts=None, baseline="shared", data_seed=1234,
                   sample_seed=4321):
    """One training forward + backward of the reference policy in train() mode (BatchNorm: batch statistics), as
    REINFORCE.shared_step / POMO.shared_step run it: out = policy(td, env, phase="train"[, num_starts]);
    loss = -((reward - baseline) * log_likelihood).mean()  [rl/reinforce/reinforce.py:59-106, zoo/pomo/model.py:89-112,
    rl/reinforce/baselines.py:45-61: NoBaseline -> 0, SharedBaseline -> reward.mean(1)].  The trainer classes themselves
    need Lightning (absent), so those four lines are applied here to the reference policy's own outputs.  Stored: the
    recorded sampling noise, actions, reward, per-step log-probs, loss, the gradient of every parameter (norm; full
    tensors for the small ones) and the BatchNorm running statistics after the forward."""
    from rl4co.utils.ops import unbatchify as ref_unbatchify

    Env = {"tsp": TSPEnv, "cvrp": CVRPEnv, "sdvrp": SDVRPEnv}[env_name]
    env = Env(generator_params=gen_params(env_name, num_loc), seed=data_seed)
    torch.manual_seed(data_seed)
    td_init = env.reset(batch_size=[batch])
    policy = make_policy(env_name, **(policy_kw or {})).train()
    kw = {} if num_starts is None else dict(num_starts=num_starts)
    decode_type = "sampling" if num_starts is None else "multistart_sampling"
    torch.manual_seed(sample_seed)
    with Recorder(policy) as rec:
        out = policy(td_init.clone(), env, phase="train", decode_type=decode_type, return_sum_log_likelihood=False,
                     return_hidden=True, **kw)
    logp = out["log_likelihood"]
    ll, reward = logp.sum(1), out["reward"]
    if baseline == "shared":
        r = ref_unbatchify(reward, num_starts)
        l2 = ref_unbatchify(ll, num_starts)
        loss = -((r - r.mean(dim=1, keepdims=True)) * l2).mean()
    elif baseline == "no":
        loss = -(reward * ll).mean()
    else:
        raise ValueError(baseline)
    loss.backward()
    fx = {
        "torch_version": np.array(torch.__version__), "env_name": np.array(env_name),
        "decode_type": np.array(decode_type), "baseline": np.array(baseline),
        "num_starts": np.array(0 if num_starts is None else num_starts, dtype=np.int64),
        "locs": np_(td_init["locs"]), "actions": np_(out["actions"]), "reward": np_(reward), "logp_steps": np_(logp),
        "loss": np_(loss), "noise": np.stack([np_(q) for q in rec.noise], 1),
        "embeddings": np_(out["hidden"].node_embeddings),
    }
    if env_name in ("cvrp", "sdvrp"):
        fx["demand"] = np_(td_init["demand"])
    for k, v in (policy_kw or {}).items():
        fx["policy_kw_" + k] = np.array(v)
    names, norms = [], []
    for k, prm in policy.named_parameters():
        g = prm.grad if prm.grad is not None else torch.zeros_like(prm)
        names.append(k)
        norms.append(float(g.double().norm()))
        if g.numel() <= 512:
            fx["grad__" + k] = np_(g)
    fx["grad_names"] = np.array(names)
    fx["grad_norms"] = np.array(norms, dtype=np.float64)
    for k, b in policy.named_buffers():
        if k.endswith("running_mean") or k.endswith("running_var"):
            fx["buf__" + k] = np_(b)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **fx)
    print(f"{name}: loss={float(loss):.6f} |grad|={np.sqrt((fx['grad_norms'] ** 2).sum()):.6f} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


def run_eval_case(name, env_name, num_loc, batch, method, policy_kw=None, data_seed=1234, rng_seed=77, **eval_kw):
    """The reference's evaluators (rl4co/tasks/eval.py:88-297) on one batch: eval_fn(policy, [batch]) exactly as
    evaluate_policy drives them.  Stored: the generator's batch, the best actions and rewards, and what the run drew
    at random (rotation angles of the 'symmetric' augmentation; start nodes and Exp(1) noise of SamplingEval)."""
    import math

    import rl4co.data.transforms as ref_tf
    import rl4co.tasks.eval as ref_eval

    Env = {"tsp": TSPEnv, "cvrp": CVRPEnv}[env_name]
    env = Env(generator_params=gen_params(env_name, num_loc), seed=data_seed)
    torch.manual_seed(data_seed)
    data = env.generator(batch_size=[batch])
    policy = make_policy(env_name, **(policy_kw or {}))
    phis = []
    orig_sym = ref_tf.symmetric_augmentation

    def rec_sym(xy, num_augment=8, first_augment=False):
        state = torch.get_rng_state()
        phis.append(torch.rand(xy.shape[0]) * 4 * math.pi)        # the draw symmetric_augmentation is about to make
        torch.set_rng_state(state)
        return orig_sym(xy, num_augment, first_augment)

    ref_tf.symmetric_augmentation = rec_sym
    try:
        cls = {"greedy": ref_eval.GreedyEval, "augment": ref_eval.AugmentationEval, "sampling": ref_eval.SamplingEval,
               "multistart_greedy": ref_eval.GreedyMultiStartEval,
               "multistart_greedy_augment": ref_eval.GreedyMultiStartAugmentEval}[method]
        eval_fn = cls(env, progress=False, **eval_kw)
        torch.manual_seed(rng_seed)
        with Recorder(policy) as rec:
            res = eval_fn(policy, [data.clone()])
    finally:
        ref_tf.symmetric_augmentation = orig_sym
    fx = {"torch_version": np.array(torch.__version__), "env_name": np.array(env_name), "method": np.array(method),
          "actions": np_(res["actions"]), "rewards": np_(res["rewards"])}
    for k, v in data.items():
        fx["gen_" + k] = np_(v)
    for k, v in eval_kw.items():
        fx["eval_kw_" + k] = np.array(v)
    for k, v in (policy_kw or {}).items():
        fx["policy_kw_" + k] = np.array(v)
    if phis:
        fx["phi"] = np_(phis[0])
    if rec.starts:
        fx["start_nodes"] = np_(rec.starts[0])                      # [B, samples] as torch.multinomial returned them
    if rec.noise:
        fx["noise"] = np.stack([np_(q) for q in rec.noise], 1)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **fx)
    print(f"{name}: rewards[:3]={fx['rewards'][:3]} -> {os.path.getsize(path) / 1024:.0f} KiB")


def eval_batch():
    """Tenth batch (python make_golden.py eval): the evaluation harness."""
    pomo = dict(num_encoder_layers=6, normalization="instance", use_graph_context=False)
    for env_name, seed in (("tsp", 501), ("cvrp", 502)):
        run_eval_case(f"eval_{env_name}20_greedy", env_name, 20, 6, "greedy", data_seed=seed)
        run_eval_case(f"eval_{env_name}20_augment_dihedral8", env_name, 20, 6, "augment", data_seed=seed, num_augment=8,
                      force_dihedral_8=True)
        run_eval_case(f"eval_{env_name}20_augment_symmetric", env_name, 20, 6, "augment", data_seed=seed, num_augment=8)
        run_eval_case(f"eval_{env_name}20_multistart", env_name, 20, 6, "multistart_greedy", data_seed=seed, num_starts=20)
        run_eval_case(f"eval_{env_name}20_multistart_augment_dihedral8", env_name, 20, 4, "multistart_greedy_augment",
                      data_seed=seed, num_starts=20, num_augment=8, force_dihedral_8=True)
        run_eval_case(f"eval_{env_name}20_sampling", env_name, 20, 5, "sampling", data_seed=seed, samples=6)
    run_eval_case("eval_pomo_tsp20_multistart_augment_symmetric", "tsp", 20, 4, "multistart_greedy_augment", policy_kw=pomo,
                  data_seed=503, num_starts=20, num_augment=4)


def augment_fixture():
    """Twelfth batch (python make_golden.py augment): the reference's StateAugmentation itself (data/transforms.py:106-153) with
    first_aug_identity=False -- the option whose save / restore indexes `[list(td.size()), 0]` -- and with the
    default, on one TSP batch; the drawn angles are recorded."""
    import math

    import rl4co.data.transforms as ref_tf
    from rl4co.data.transforms import StateAugmentation

    env = TSPEnv(generator_params=gen_params("tsp", 20), seed=601)
    torch.manual_seed(601)
    td = env.reset(batch_size=[5])
    fx = {"torch_version": np.array(torch.__version__), "locs": np_(td["locs"])}
    for tag, kw in (("noident", dict(num_augment=4, first_aug_identity=False)), ("default", dict(num_augment=4)),
                    ("dihedral8", dict(num_augment=8, augment_fn="dihedral8"))):
        phis = []
        orig = ref_tf.symmetric_augmentation

        def rec(xy, num_augment=8, first_augment=False):
            state = torch.get_rng_state()
            phis.append(torch.rand(xy.shape[0]) * 4 * math.pi)
            torch.set_rng_state(state)
            return orig(xy, num_augment, first_augment)

        ref_tf.symmetric_augmentation = rec
        try:
            torch.manual_seed(77)
            out = StateAugmentation(**kw)(td.clone())
        finally:
            ref_tf.symmetric_augmentation = orig
        fx[f"{tag}_locs"] = np_(out["locs"])
        if phis:
            fx[f"{tag}_phi"] = np_(phis[0])
    path = os.path.join(HERE, "state_augmentation.npz")
    np.savez_compressed(path, **fx)
    print(f"state_augmentation -> {os.path.getsize(path) / 1024:.0f} KiB")


def inject():
    """Eleventh batch (python make_golden.py inject): the reference's PointerAttention module (nn/attention.py:224-328) on
    recorded inputs -- what a `pointer=` replacement must reproduce."""
    from rl4co.models.nn.attention import PointerAttention

    fx = {"torch_version": np.array(torch.__version__)}
    E, H = 128, 8
    pa = PointerAttention(E, H, mask_inner=True, out_bias=False, check_nan=True)
    w = goldweights.tensor_for("decoder.pointer.project_out.weight", (E, E))
    with torch.no_grad():
        pa.project_out.weight.copy_(torch.from_numpy(w))
    g = torch.Generator().manual_seed(2024)
    for tag, (B, L, M) in {"single": (3, 1, 20), "multi": (2, 5, 33), "wide": (2, 1, 150)}.items():
        q = torch.randn(B, L, E, generator=g)
        kvl = torch.randn(B, M, 3 * E, generator=g)
        k, v, lk = kvl.chunk(3, dim=-1)                      # strided views, as the decoder's cache chunks are
        mask = torch.rand(B, L, M, generator=g) > 0.3
        mask[..., 0] = True
        if L == 1:
            mask = mask[:, 0]
        with torch.inference_mode():
            logits = pa(q, k, v, lk, mask)
        fx.update({f"{tag}_q": np_(q), f"{tag}_kvl": np_(kvl), f"{tag}_mask": np_(mask), f"{tag}_logits": np_(logits)})
    path = os.path.join(HERE, "pointer_attention.npz")
    np.savez_compressed(path, **fx)
    print(f"pointer_attention -> {os.path.getsize(path) / 1024:.0f} KiB")


def train():
    """Ninth batch (python make_golden.py train): training forward / backward of the reference policy."""
    pomo = dict(num_encoder_layers=6, normalization="instance", use_graph_context=False)
    run_train_case("train_pomo_tsp20", "tsp", 20, 4, policy_kw=pomo, num_starts=20, baseline="shared")
    run_train_case("train_am_tsp20_bn", "tsp", 20, 8, baseline="no", data_seed=11)
    run_train_case("train_am_cvrp20_bn", "cvrp", 20, 8, baseline="no", data_seed=12)
    run_train_case("train_am_tsp20_bn_multistart", "tsp", 20, 4, num_starts=10, baseline="shared", data_seed=13)
    run_train_case("train_am_sdvrp20", "sdvrp", 20, 6, baseline="no", data_seed=14)      # round 3: the split-delivery gradient path


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "train":
        train()
    elif len(sys.argv) > 1 and sys.argv[1] == "eval":
        eval_batch()
    elif len(sys.argv) > 1 and sys.argv[1] == "inject":
        inject()
    elif len(sys.argv) > 1 and sys.argv[1] == "augment":
        augment_fixture()
    elif len(sys.argv) > 1 and sys.argv[1] == "cvrptw":
        cvrptw()
    elif len(sys.argv) > 1 and sys.argv[1] == "op":
        op()
    elif len(sys.argv) > 1 and sys.argv[1] == "pctsp":
        pctsp()
    elif len(sys.argv) > 1 and sys.argv[1] == "sdvrp":
        sdvrp()
    elif len(sys.argv) > 1 and sys.argv[1] == "filtering":
        filtering()
    elif len(sys.argv) > 1 and sys.argv[1] == "beam":
        beam()
    elif len(sys.argv) > 1 and sys.argv[1] == "extra":
        extra()
    else:
        main()
