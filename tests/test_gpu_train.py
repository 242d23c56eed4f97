"""GPU: the training-side boundary.

`AttentionModelPolicy.forward(phase="train")` under autograd returns a `log_likelihood` with a grad_fn, so the
reference's trainers run on it unchanged.  The tests restate `REINFORCE.calculate_loss`
(rl4co/models/rl/reinforce/reinforce.py:79-106) and `POMO.shared_step` (rl4co/models/zoo/pomo/model.py:89-112) line
for line around the policy call, backpropagate, and compare with gradients recorded from the reference itself
(`tests/golden/make_golden.py train`: the reference policy in train() mode, sampling with recorded noise).
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import cfg_for, golden, golden_weights, instance_of
from test_gpu_parity import DEV, assert_bits_equal, make_policy, make_td, t

pytestmark = pytest.mark.gpu

TRAIN_CASES = ["train_pomo_tsp20", "train_am_tsp20_bn", "train_am_cvrp20_bn", "train_am_tsp20_bn_multistart", "train_am_sdvrp20"]


class NoBaseline:
    """rl4co/models/rl/reinforce/baselines.py:45-49"""

    def eval(self, td, reward, env=None):
        return 0, 0


class SharedBaseline:
    """rl4co/models/rl/reinforce/baselines.py:52-61"""

    def eval(self, td, reward, env=None, on_dim=1):
        return reward.mean(dim=on_dim, keepdims=True), 0


def calculate_loss(baseline, td, env, policy_out, reward=None, log_likelihood=None):
    """REINFORCE.calculate_loss, reinforce.py:79-106 (no `extra`, advantage scaler = identity)."""
    reward = reward if reward is not None else policy_out["reward"]
    log_likelihood = log_likelihood if log_likelihood is not None else policy_out["log_likelihood"]
    bl_val, bl_loss = baseline.eval(td, reward, env)
    advantage = reward - bl_val
    reinforce_loss = -(advantage * log_likelihood).mean()
    loss = reinforce_loss + bl_loss
    policy_out.update({"loss": loss, "reinforce_loss": reinforce_loss, "bl_loss": bl_loss, "bl_val": bl_val})
    return policy_out


def _train_policy(fx):
    pol = make_policy(cfg_for(fx)).train()
    return pol


def _shared_step(fx, pol, env, td):
    """REINFORCE.shared_step (reinforce.py:59-71) / POMO.shared_step (pomo/model.py:89-112), training phase."""
    import eam_rl4co_amd as ea

    ns = int(fx["num_starts"])
    noise = t(fx["noise"])
    if ns > 1:      # POMO.shared_step
        out = pol(td, env, phase="train", num_starts=ns, decode_type="multistart_sampling", noise=noise)
        reward = ea.unbatchify(out["reward"], (0, ns))
        log_likelihood = ea.unbatchify(out["log_likelihood"], (0, ns))
        calculate_loss(SharedBaseline(), td, env, out, reward, log_likelihood)
    else:           # REINFORCE.shared_step with the "no" baseline
        out = pol(td, env, phase="train", select_best=False, decode_type="sampling", noise=noise)
        calculate_loss(NoBaseline(), td, env, out)
    return out


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_reference_trainer_step_runs_on_the_policy_and_matches_reference_gradients(oracle, name):
    fx = golden(name)
    env_name = str(fx["env_name"])
    pol = _train_policy(fx)
    env, td = make_td(env_name, fx["locs"], instance_of(fx))
    out = _shared_step(fx, pol, env, td)
    # forward: the native rollout in train() mode (batch statistics) reproduces the reference's sampled tours ...
    assert_bits_equal(out["actions"], fx["actions"], "actions")
    np.testing.assert_allclose(out["reward"].cpu().numpy(), fx["reward"], rtol=1e-6)
    ll = out["log_likelihood"]
    assert ll.requires_grad and ll.grad_fn is not None
    np.testing.assert_allclose(ll.detach().cpu().numpy(), fx["logp_steps"].sum(1), rtol=2e-5, atol=2e-5)
    # ... and is bit-equal to the oracle in the same mode (the value of the differentiable log-likelihood IS the native one)
    sd = golden_weights(cfg_for(fx))
    o = oracle.policy_rollout(sd, env_name, fx["locs"], instance_of(fx), decode_type=str(fx["decode_type"]),
                              num_starts=int(fx["num_starts"]), noise=fx["noise"],
                              use_graph_context=bool(fx.get("policy_kw_use_graph_context", True)), training=True)
    assert_bits_equal(ll.detach(), o["log_likelihood"], "log_likelihood (native value)")
    for k in fx:     # running statistics after the forward: bit-equal to the oracle, 1e-6 from the reference
        if k.startswith("buf__"):
            buf = dict(pol.named_buffers())[k[5:]]
            assert_bits_equal(buf, sd[k[5:]], k)
            np.testing.assert_allclose(buf.cpu().numpy(), fx[k], rtol=0, atol=1e-6)
    np.testing.assert_allclose(float(out["loss"].detach()), float(fx["loss"]), rtol=2e-5)
    # backward: gradient of every parameter against the reference's
    out["loss"].backward()
    params = dict(pol.named_parameters())
    total_ref = float(np.sqrt((fx["grad_norms"] ** 2).sum()))
    for k, ref_norm in zip(fx["grad_names"], fx["grad_norms"]):
        g = params[str(k)].grad
        got = 0.0 if g is None else float(g.double().norm())
        assert abs(got - ref_norm) <= 1e-4 * max(ref_norm, 1e-3 * total_ref) + 1e-7, (str(k), got, ref_norm)
        if "grad__" + str(k) in fx and g is not None:
            ref = fx["grad__" + str(k)]
            np.testing.assert_allclose(g.cpu().numpy(), ref, rtol=0, atol=1e-4 * max(float(np.abs(ref).max()), 1e-3 * total_ref),
                                       err_msg=str(k))


@pytest.mark.parametrize("B,N,E", [(1, 5, 128), (3, 20, 128), (64, 100, 128), (5, 101, 64), (2, 7, 200)])
def test_instance_norm_training_kernels_match_torch(B, N, E):
    """eamrl_instance_norm_forward / _backward (the normalisation of the differentiable POMO encoder) against
    torch.nn.functional.instance_norm and its autograd: values, input gradient, affine gradients."""
    import torch.nn.functional as F
    from eam_rl4co_amd.train import _InstanceNormFn

    torch.manual_seed(B * 100 + N)
    x = (torch.randn(B, N, E, device=DEV) * 2 + 0.5).requires_grad_()
    g = (torch.rand(E, device=DEV) + 0.5).requires_grad_()
    b = torch.randn(E, device=DEV).requires_grad_()
    w = torch.randn(B, N, E, device=DEV)
    y = _InstanceNormFn.apply(x, g, b, 1e-5)
    (y * w).sum().backward()
    got = (y.detach(), x.grad.clone(), g.grad.clone(), b.grad.clone())
    for t_ in (x, g, b):
        t_.grad = None
    yr = F.instance_norm(x.permute(0, 2, 1), weight=g, bias=b, eps=1e-5).permute(0, 2, 1)
    (yr * w).sum().backward()
    ref = (yr.detach(), x.grad, g.grad, b.grad)
    for a_, r_, nm in zip(got, ref, ("y", "dx", "dgamma", "dbeta")):
        scale = max(float(r_.abs().max()), 1e-6)
        assert float((a_ - r_).abs().max()) <= 2e-5 * scale + 1e-6, nm


@pytest.mark.parametrize("env_name,N,B,ns", [("cvrp", 20, 7, 0), ("cvrp", 100, 3, 6), ("cvrp", 127, 2, 0), ("cvrptw", 20, 5, 0),
                                             ("cvrptw", 50, 2, 4), ("pctsp", 20, 6, 0), ("pctsp", 100, 2, 5), ("op", 20, 6, 0),
                                             ("op", 100, 3, 4),
                                             # graphs above 112 nodes: the chunked mask layout, both ways
                                             ("cvrp", 150, 3, 0), ("cvrp", 230, 2, 3), ("pctsp", 130, 2, 0), ("op", 300, 2, 0)])
def test_replay_states_kernel_equals_step_by_step(env_name, N, B, ns):
    """eamrl_replay_states (the env transitions replayed inside one kernel) gives bit for bit what T rounds of
    {pack mask bits, copy current node, state scalar, env step kernel} give: the inputs of the re-evaluation kernels."""
    import os

    import eam_rl4co_amd as ea
    from eam_rl4co_amd.train import replay_states

    env = ea.get_env(env_name, generator_params=dict(num_loc=N), seed=N + B)
    torch.manual_seed(N)
    td = env.reset(batch_size=[B]).to(DEV)
    pol = make_policy("am_" + env_name)
    kw = dict(decode_type="multistart_sampling", num_starts=ns) if ns else dict(decode_type="sampling")
    with torch.no_grad():
        acts = pol(td.clone(), env, phase="test", **kw)["actions"].contiguous()
    res = []
    for loop in ("0", "1"):
        os.environ["EAMRL_REPLAY_LOOP"] = loop
        try:
            res.append(replay_states(pol, td, acts, max(ns, 1), bool(ns)))
        finally:
            os.environ.pop("EAMRL_REPLAY_LOOP", None)
    a, b = res
    assert torch.equal(a["maskbits"], b["maskbits"]) and torch.equal(a["idxA"], b["idxA"])
    assert_bits_equal(a["sc"], b["sc"].cpu().numpy(), "state scalars")
    assert a["tstart"] == b["tstart"] and a["idxB"] is None and a["sc"].shape[0] == (2 if env_name == "cvrptw" else 1)
    assert int((a["maskbits"] != 0).sum()) > 0


@pytest.mark.parametrize("N,B,ns", [(20, 7, 0), (50, 3, 5), (100, 2, 4), (111, 2, 0), (150, 2, 0), (300, 2, 3)])
def test_sdvrp_replay_kernel_equals_torch_state_loop(N, B, ns):
    """eamrl_replay_states_sdvrp (SDVRPEnv._step + get_action_mask replayed inside one kernel, sdvrp/env.py:58-92,137-146) gives
    the masks, current nodes, free capacities and remaining demands of the PyTorch step loop the fallback path runs (bit for bit:
    the same float operations in the same order)."""
    import eam_rl4co_amd as ea
    from eam_rl4co_amd.train import _sdvrp_states, replay_states

    env = ea.get_env("sdvrp", generator_params=dict(num_loc=N), seed=N + B)
    torch.manual_seed(N)
    td = env.reset(batch_size=[B]).to(DEV)
    pol = make_policy("am_sdvrp")
    kw = dict(decode_type="multistart_sampling", num_starts=ns) if ns else dict(decode_type="sampling")
    with torch.no_grad():
        acts = pol(td.clone(), env, phase="test", **kw)["actions"].contiguous()
    S = max(ns, 1)
    got = replay_states(pol, td, acts, S, bool(ns))
    M = N + 1
    rep = lambda x: x.repeat(S, *([1] * (x.dim() - 1)))
    cur, free, mask, rem = _sdvrp_states(acts, rep(td["demand"]), rep(td["vehicle_capacity"].reshape(-1)), M)
    bits = got["maskbits"].cpu().numpy().astype(np.uint32)                       # [R, T, 4], or [R, T, nkc, 4] above 112 nodes
    flat = ((bits[..., None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(*bits.shape[:-1], 128).astype(bool)
    grem = got["rem"]
    if M > 112:      # chunked layout: node n at chunk n // 112, slot n % 112
        nkc = -(-M // 112)
        assert flat.shape[2] == nkc and grem.shape[2:] == (nkc, 128)
        assert not flat[..., 112:].any() and float(grem[..., 112:].abs().max()) == 0.0
        flat = flat[..., :112].reshape(*flat.shape[:2], nkc * 112)
        grem = grem[..., :112].reshape(*grem.shape[:2], nkc * 112)
    assert np.array_equal(flat[..., :M], mask.cpu().numpy()) and not flat[..., M:].any()
    assert torch.equal(got["idxA"].long(), cur)
    assert_bits_equal(got["sc"][0], free.cpu().numpy(), "free capacity")
    assert_bits_equal(grem[..., :M], rem.cpu().numpy(), "remaining demands")
    assert float(grem[..., M:].abs().max()) == 0.0 and got["tstart"] == (1 if ns else 0)


@pytest.mark.parametrize("env_name,cfg,N", [("tsp", "pomo_tsp", 20), ("cvrp", "am_cvrp", 20)])
def test_eam_shared_step_restated_runs_on_the_policy(env_name, cfg, N):
    """EAM.shared_step's training branch (zoo/earl/model.py:146-247, POMO baseline) restated line by line around the policy
    and evolution_worker -- including the `return_entropy=True` of its first policy call: differentiable log-likelihoods
    for the sampled and the improved tours, the combined loss equal to train.eam_loss with the same evolution draws."""
    import eam_rl4co_amd as ea
    from eam_rl4co_amd import train

    B, S = 5, 8
    env = ea.get_env(env_name, generator_params=dict(num_loc=N), seed=11)
    torch.manual_seed(11)
    td = env.reset(batch_size=[B]).to(DEV)
    init_td = td.clone()
    pol = make_policy(cfg).train()
    runner = ea.EA(env, dict(num_generations=2, mutation_rate=0.3, crossover_rate=0.8, selection_rate=0.6))
    noise = torch.empty(B * S, 3 * td["locs"].shape[1] + 1, td["locs"].shape[1], device=DEV).exponential_(1)

    def align(improved, original):                       # EAM._align_improved_actions, model.py:116-127
        if improved.shape[-1] + 1 == original.shape[-1]:
            return torch.cat([original[..., :1], improved], dim=-1)
        return improved

    # run_original_policy (model.py:150-156)
    original_out = pol(td, env, phase="train", num_starts=S, return_entropy=True, decode_type="multistart_sampling", noise=noise)
    assert original_out["entropy"].shape == (B * S,) and (original_out["entropy"] > 0).all()
    # run_improved_policy (model.py:158-203)
    gen = torch.Generator(device=DEV).manual_seed(3)
    improved_actions, _ = ea.evolution_worker(original_out["actions"], init_td, runner, env, generator=gen)
    improved_actions = align(improved_actions, original_out["actions"])
    improved_out = pol(init_td, env, phase="train", num_starts=S, actions=improved_actions)
    # losses (model.py:205-244)
    original_reward = ea.unbatchify(original_out["reward"], (0, S))
    original_ll = ea.unbatchify(original_out["log_likelihood"], (0, S))
    calculate_loss(SharedBaseline(), td, env, original_out, original_reward, original_ll)
    improved_reward = ea.unbatchify(improved_out["reward"], (0, S))
    improved_ll = ea.unbatchify(improved_out["log_likelihood"], (0, S))
    combined_out = {k: torch.cat([original_out[k], improved_out[k]], dim=0) for k in original_out
                    if k in improved_out and isinstance(original_out[k], torch.Tensor) and original_out[k].dim() > 0}
    combined_reward = torch.cat([original_reward, improved_reward], dim=0)
    combined_ll = torch.cat([original_ll, improved_ll], dim=0)
    calculate_loss(SharedBaseline(), None, env, combined_out, combined_reward, combined_ll)
    loss = combined_out["loss"]
    assert loss.requires_grad
    pol.zero_grad()
    loss.backward()
    grads = {k: p.grad.clone() for k, p in pol.named_parameters() if p.grad is not None}
    assert grads and all(torch.isfinite(g).all() for g in grads.values())
    # the packaged step with the same noise and evolution draws
    pol.zero_grad()
    pol2 = pol
    gen = torch.Generator(device=DEV).manual_seed(3)
    orig_forward = pol2.forward

    def with_noise(td_, env_, **kw):                     # train.eam_loss samples its own noise: pin it to the recorded one
        if kw.get("actions") is None:
            kw["noise"] = noise
        return orig_forward(td_, env_, **kw)

    pol2.forward = with_noise
    calls, enc = [], train.encode_autograd
    train.encode_autograd = lambda *a, **k: (calls.append(1), enc(*a, **k))[1]
    try:
        res = train.eam_loss(pol2, env, td.clone(), runner, num_starts=S, generator=gen)
    finally:
        pol2.forward = orig_forward
        train.encode_autograd = enc
    assert torch.equal(res["improved_actions"], improved_actions)
    np.testing.assert_allclose(float(res["loss"].detach()), float(loss.detach()), rtol=1e-6)
    # ... in which the sampled and the improved tours share ONE differentiable encoder pass (train.shared_decoder_tensors):
    # same gradients as the two separate graphs above
    assert len(calls) == 1 and pol2._shared_dt is None
    res["loss"].backward()
    gnorm = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())))
    for k, p_ in pol2.named_parameters():
        if k in grads:
            rel = float((p_.grad.double() - grads[k].double()).norm()) / max(float(grads[k].double().norm()), 1e-2 * gnorm)
            assert rel <= 1e-4, (k, rel)


@pytest.mark.parametrize("cfg,env_name,ns", [("am_tsp", "tsp", 0), ("am_cvrp", "cvrp", 0), ("am_cvrp", "cvrp", 5),
                                              ("pomo_tsp", "tsp", 8)])
def test_differentiated_policy_is_the_sampled_policy(cfg, env_name, ns):
    """ADVICE r1 (train.py:319): in train() mode the re-evaluated log-likelihood equals the native rollout's (same
    normalisation statistics in both passes), for batch-norm (am_*) and instance-norm policies."""
    import eam_rl4co_amd as ea
    from eam_rl4co_amd.train import evaluate_log_likelihood

    env = ea.get_env(env_name, generator_params=dict(num_loc=20), seed=7)
    torch.manual_seed(7)
    td = env.reset(batch_size=[16]).to(DEV)
    pol = make_policy(cfg).train()
    kw = dict(num_starts=ns) if ns else {}
    with torch.no_grad():
        out = pol(td, env, phase="train", return_sum_log_likelihood=False, **kw)
    assert not out["log_likelihood"].requires_grad
    re = evaluate_log_likelihood(pol, td, env, out["actions"], num_starts=ns)
    np.testing.assert_allclose(re.detach().cpu().numpy(), out["log_likelihood"].cpu().numpy(), rtol=0, atol=1e-4)
    # and eval() mode (running statistics) is a different network for a BatchNorm policy
    if cfg.startswith("am"):
        re_eval = evaluate_log_likelihood(pol.eval(), td, env, out["actions"], num_starts=ns)
        assert (re_eval - re).abs().max() > 1e-3


def test_forward_is_not_differentiated_outside_training():
    import eam_rl4co_amd as ea

    env = ea.get_env("tsp", generator_params=dict(num_loc=10), seed=1)
    td = env.reset(batch_size=[4]).to(DEV)
    pol = make_policy("am_tsp").train()
    assert not pol(td, env, phase="val", decode_type="greedy")["log_likelihood"].requires_grad
    with torch.no_grad():
        assert not pol(td, env, phase="train")["log_likelihood"].requires_grad
    with torch.inference_mode():
        assert not pol(td, env, phase="train")["log_likelihood"].requires_grad
    out = pol(td, env, phase="train")
    assert out["log_likelihood"].requires_grad and out["log_likelihood"].shape == (4,)
    out2 = pol(td, env, phase="train", actions=out["actions"])       # teacher forcing (EAM, earl/model.py:189-195)
    assert out2["log_likelihood"].requires_grad
    assert torch.equal(out2["reward"], out["reward"])
    best = pol(td, env, phase="train", num_starts=4, select_best=True)      # (round 3: differentiable too; values and gradients
    assert best["log_likelihood"].requires_grad and best["log_likelihood"].shape == (4,)     # in test_gradients_through_filtering_...)


def test_checkpointed_reevaluation_gives_the_same_gradients():
    """The chunked, recompute-in-backward form (bounded memory at POMO scale) == the plain autograd graph."""
    import eam_rl4co_amd as ea
    from eam_rl4co_amd.train import evaluate_log_likelihood

    env = ea.get_env("cvrp", generator_params=dict(num_loc=15), seed=2)
    torch.manual_seed(2)
    td = env.reset(batch_size=[6]).to(DEV)
    pol = make_policy("am_cvrp").train()
    with torch.no_grad():
        out = pol(td, env, phase="train", num_starts=9, return_sum_log_likelihood=False)
    w = torch.randn(out["actions"].shape[0], device=DEV)
    grads = []
    for ck in (False, True):
        pol.zero_grad()
        lp = evaluate_log_likelihood(pol, td, env, out["actions"], num_starts=9, chunk_rows=12, checkpoint=ck)
        (lp.sum(1) * w).sum().backward()
        grads.append({k: p.grad.clone() for k, p in pol.named_parameters() if p.grad is not None})
    assert grads[0].keys() == grads[1].keys() and len(grads[0]) > 20
    top = max(float(g.abs().max()) for g in grads[0].values())
    for k in grads[0]:      # (rounding noise is relative to the largest gradients: e.g. key biases have a zero true gradient)
        scale = max(float(grads[0][k].abs().max()), 1e-2 * top)
        assert float((grads[0][k] - grads[1][k]).abs().max()) <= 1e-4 * scale, k


def test_batchnorm_train_kernel_bit_exact(oracle):
    from eam_rl4co_amd import ops

    rng = np.random.default_rng(3)
    for rows, E in ((1, 128), (127, 128), (128, 128), (129, 64), (5000, 128), (2100, 96)):
        x = (rng.standard_normal((rows, E)) * 2 + 0.5).astype(np.float32)
        g, b = rng.standard_normal(E).astype(np.float32), rng.standard_normal(E).astype(np.float32)
        rm, rv = rng.standard_normal(E).astype(np.float32), (rng.random(E) + 0.5).astype(np.float32)
        orm, orv = rm.copy(), rv.copy()
        y, m, v = oracle.batchnorm_train(x, g, b, orm, orv, momentum=0.1, eps=1e-5)
        xt, rmt, rvt = t(x), t(rm), t(rv)
        _, mt, vt = ops.batchnorm_train_(xt, t(g), t(b), rmt, rvt, 0.1, 1e-5)
        for got, want, what in ((xt, y, "y"), (mt, m, "mean"), (vt, v, "var"), (rmt, orm, "running_mean"), (rvt, orv, "running_var")):
            assert_bits_equal(got, want, f"{what} rows={rows} E={E}")


@pytest.mark.parametrize("cfg,env_name,N,B,ns", [("pomo_tsp", "tsp", 50, 5, 50), ("pomo_tsp", "tsp", 100, 3, 100),
                                                 ("am_cvrp", "cvrp", 50, 4, 20), ("am_cvrp", "cvrp", 100, 2, 100)])
def test_backward_with_the_rollouts_heads_equals_recomputation(cfg, env_name, N, B, ns, monkeypatch):
    """policy.forward(phase="train") on a multistart batch: the start-sharing rollout kernel keeps every step's glimpse output
    (eamrl_state.heads_out) and the logits-backward kernel reads it instead of recomputing the glimpse -- same gradients as with
    EAMRL_REEVAL_RECOMPUTE_HEADS=1 (done CVRP rows and the steps after an instance's last have zero heads and zero gradient)."""
    import eam_rl4co_amd as ea
    from eam_rl4co_amd import ops

    env = ea.get_env(env_name, generator_params=dict(num_loc=N), seed=N + 1)
    torch.manual_seed(N + B)
    td = env.reset(batch_size=[B]).to(DEV)
    pol = make_policy(cfg).train()
    noise = torch.empty(B * ns, (2 * N + 2 if env_name == "cvrp" else N), td["locs"].shape[1], device=DEV).exponential_(1)
    seen, plan_init = [], ops.ReevalPlan.__init__

    def spy(self, *a, **k):
        seen.append(k.get("rollout_heads") is not None)
        return plan_init(self, *a, **k)

    monkeypatch.setattr(ops.ReevalPlan, "__init__", spy)
    res = []
    for recompute in ("0", "1"):
        monkeypatch.setenv("EAMRL_REEVAL_RECOMPUTE_HEADS", recompute)
        pol.zero_grad()
        out = pol(td.clone(), env, phase="train", decode_type="multistart_sampling", num_starts=ns, noise=noise,
                  return_entropy=True)          # (the entropy pass reads the rollout's heads too)
        w = torch.randn(out["log_likelihood"].shape, generator=torch.Generator().manual_seed(1)).to(DEV)
        (out["log_likelihood"] * w).sum().backward()
        res.append((out["actions"], {k: p.grad.clone() for k, p in pol.named_parameters() if p.grad is not None}, out["entropy"]))
    assert seen == [True, True, False, False]       # entropy plan, gradient plan
    assert torch.equal(res[0][0], res[1][0])
    np.testing.assert_allclose(res[0][2].cpu().numpy(), res[1][2].cpu().numpy(), rtol=2e-5, atol=1e-5)
    gnorm = float(torch.sqrt(sum((g.double() ** 2).sum() for g in res[1][1].values())))
    for k, ref in res[1][1].items():
        rel = float((res[0][1][k].double() - ref.double()).norm()) / max(float(ref.double().norm()), 1e-2 * gnorm)
        assert rel <= 2e-5, (k, rel)


def test_training_steps_do_not_accumulate_device_memory():
    """The autograd node of the native re-evaluation holds its plan (log-probs, mask bits, the rollout's heads -- gigabytes at
    the POMO sizes): nothing may tie it into a reference cycle, or step after step stays allocated until Python's cyclic
    collector happens to run (seen: one heads buffer per step, 78 GiB peak in the CVRP training bench)."""
    import gc

    import eam_rl4co_amd as ea
    from eam_rl4co_amd.train import PolicyGradientStep

    env = ea.get_env("cvrp", generator_params=dict(num_loc=20), seed=2)
    torch.manual_seed(2)
    pol = make_policy("am_cvrp").train()
    tds = [env.reset(batch_size=[16]).to(DEV) for _ in range(2)]
    stepper = PolicyGradientStep(pol, env, num_starts=20)
    gc.collect()
    gc.disable()
    try:
        after = []
        for i in range(6):
            out = stepper(tds[i % 2])
            torch.cuda.synchronize()
            after.append(torch.cuda.memory_allocated())
    finally:
        gc.enable()
    heads_bytes = 16 * 20 * 43 * 128 * 4
    assert max(after[2:]) - after[1] < heads_bytes // 2, after


@pytest.mark.parametrize("B,N", [(3, 20), (2, 21), (5, 50), (2, 64), (3, 100), (1, 101), (2, 112), (1, 1), (1, 17)])
def test_encoder_attention_backward_kernel_matches_torch(B, N):
    """eamrl_mha_encoder_backward against autograd through scaled_dot_product_attention on the same packed qkv."""
    from eam_rl4co_amd.train import _self_attention

    g = torch.Generator(device="cpu").manual_seed(B * 131 + N)
    qkv0 = (torch.randn(B, N, 384, generator=g) * 1.5).to(DEV)
    w = torch.randn(B, N, 128, generator=g).to(DEV)
    res = []
    for native in (True, False):
        qkv = qkv0.clone().requires_grad_()
        if native:
            y = _self_attention(qkv, B, N, 128, 8)
        else:
            q = qkv.double().view(B, N, 3, 8, 16).permute(2, 0, 3, 1, 4)
            y = F.scaled_dot_product_attention(q[0], q[1], q[2]).permute(0, 2, 1, 3).reshape(B, N, 128)
        (y * w.to(y.dtype)).sum().backward()
        res.append((y.detach().double(), qkv.grad.double()))
    assert type(res[0][0]) is torch.Tensor
    np.testing.assert_allclose(res[0][0].cpu().numpy(), res[1][0].cpu().numpy(), rtol=0, atol=2e-6 * float(res[1][0].abs().max()))
    scale = float(res[1][1].abs().max())
    assert float((res[0][1] - res[1][1]).abs().max()) <= 1e-5 * scale


@pytest.mark.parametrize("env_name,N,B,layers,gctx", [
    ("tsp", 20, 7, 6, False), ("tsp", 100, 5, 6, False), ("tsp", 50, 64, 6, False), ("tsp", 21, 3, 3, True),
    ("cvrp", 20, 6, 6, False), ("cvrp", 100, 3, 6, False), ("pctsp", 20, 4, 3, True), ("op", 20, 4, 3, True),
    ("cvrptw", 20, 4, 3, True), ("sdvrp", 20, 4, 3, True),
])
def test_training_graph_encoder_equals_native_encoder(env_name, N, B, layers, gctx):
    """The differentiable encoder of the training graph (train.encode_autograd: eamrl_linear, eamrl_mha_encoder and
    eamrl_instance_norm_forward behind autograd Functions, the init embedding's value from the native kernel) reproduces
    the native encoder's embeddings bit for bit for instance-norm policies -- the fused kernel's, which in turn equal the
    oracle's.  That is what lets one encoder pass serve rollout and gradient (AttentionModelPolicy._one_encoder_pass)."""
    import eam_rl4co_amd as ea
    from eam_rl4co_amd.train import encode_autograd, graph_encoder_equals_native

    env = ea.get_env(env_name, generator_params=dict(num_loc=N), seed=N)
    torch.manual_seed(N + B)
    td = env.reset(batch_size=[B]).to(DEV)
    pol = ea.AttentionModelPolicy(env_name=env_name, num_encoder_layers=layers, normalization="instance",
                                  use_graph_context=gctx).to(DEV).train()
    assert graph_encoder_equals_native(pol, td)
    with torch.no_grad():
        native, _ = pol.encoder(td)
    graph = encode_autograd(pol, td)
    assert graph.requires_grad
    assert_bits_equal(graph.detach(), native.cpu().numpy(), "embeddings")
    # ... and a training forward built on it gives the rollout of the two-pass version, bit for bit
    res = []
    for separate in ("0", "1"):
        os.environ["EAMRL_SEPARATE_ENCODER_PASSES"] = separate
        try:
            torch.manual_seed(5)
            out = pol(td.clone(), env, phase="train", decode_type="sampling")
        finally:
            os.environ.pop("EAMRL_SEPARATE_ENCODER_PASSES", None)
        assert out["log_likelihood"].requires_grad
        res.append(out)
    assert torch.equal(res[0]["actions"], res[1]["actions"]) and torch.equal(res[0]["reward"], res[1]["reward"])
    assert torch.equal(res[0]["log_likelihood"].detach(), res[1]["log_likelihood"].detach())
    if env_name in ("tsp", "cvrp"):     # batch-norm policies keep the two passes (torch's batch_norm in the graph)
        assert not graph_encoder_equals_native(make_policy("am_" + env_name), td)


@pytest.mark.parametrize("rows,out_dim,in_dim,strided", [
    (1, 128, 128, False), (15, 128, 128, False), (64, 384, 128, False), (1000, 512, 128, False), (777, 128, 512, False),
    (6400, 384, 128, True), (102400, 128, 128, False),
])
def test_linear_weight_gradient_kernel_matches_torch(rows, out_dim, in_dim, strided):
    """eamrl_linear_wgrad (fp32 MFMA over the rows, chunked, fixed-order reduction) against dy^T x in float64."""
    from eam_rl4co_amd import ops

    g = torch.Generator(device="cpu").manual_seed(rows + out_dim)
    if strided:      # operands as column slices of wider tensors (the K | V | L cache projections)
        dy = torch.randn(rows, out_dim + 128, generator=g).to(DEV)[:, 128:]
        x = torch.randn(rows, in_dim + 256, generator=g).to(DEV)[:, 128:128 + in_dim]
    else:
        dy, x = torch.randn(rows, out_dim, generator=g).to(DEV), torch.randn(rows, in_dim, generator=g).to(DEV)
    dW, db = ops.linear_wgrad(dy, x)
    ref_w, ref_b = dy.double().t() @ x.double(), dy.double().sum(0)
    scale = float(ref_w.abs().max())
    assert float((dW.double() - ref_w).abs().max()) <= 2e-6 * max(scale, 1.0) * max(1.0, rows ** 0.5 / 8)
    assert float((db.double() - ref_b).abs().max()) <= 2e-6 * max(float(ref_b.abs().max()), 1.0) * max(1.0, rows ** 0.5 / 8)
    dW2, none = ops.linear_wgrad(dy, x, need_bias=False)
    assert none is None and torch.equal(dW2, dW)             # fixed-order reduction: bitwise reproducible
    assert not ops.linear_wgrad_supported(96, 128) and ops.linear_wgrad_supported(384, 128)


@pytest.mark.parametrize("relu,with_res", [(False, False), (True, False), (False, True)])
def test_linear_autograd_function_matches_torch(relu, with_res):
    from eam_rl4co_amd.train import _linear

    torch.manual_seed(11)
    x0 = torch.randn(7, 33, 128, device=DEV)
    W0, b0 = torch.randn(512, 128, device=DEV) * 0.1, torch.randn(512, device=DEV)
    w = torch.randn(7, 33, 512, device=DEV)
    res = []
    r0 = torch.randn(7, 33, 512, device=DEV)
    for native in (True, False):
        x, W, b = x0.clone().requires_grad_(), W0.clone().requires_grad_(), b0.clone().requires_grad_()
        r = r0.clone().requires_grad_() if with_res else None
        if native:
            y = _linear(x, W, b, relu=relu, residual=r)
        else:
            y = F.relu(F.linear(x, W, b)) if relu else F.linear(x, W, b)
            y = y + r if with_res else y
        (y * w).sum().backward()
        res.append((y.detach(), x.grad, W.grad, b.grad) + ((r.grad,) if with_res else ()))
    for got, want in zip(res[0], res[1]):
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("cfg,env_name,N,B,ns,ms", [
    ("am_tsp", "tsp", 20, 6, 0, None), ("am_tsp", "tsp", 20, 3, 7, None), ("am_tsp", "tsp", 20, 3, 5, False),
    ("pomo_tsp", "tsp", 50, 4, 50, None), ("pomo_tsp", "tsp", 100, 3, 10, None), ("am_tsp", "tsp", 112, 2, 0, None),
    ("am_cvrp", "cvrp", 20, 5, 0, None), ("am_cvrp", "cvrp", 50, 3, 6, None), ("am_cvrp", "cvrp", 100, 2, 3, None),
    ("am_pctsp", "pctsp", 20, 4, 0, None), ("am_op", "op", 20, 4, 4, None), ("am_cvrptw", "cvrptw", 20, 4, 0, None),
    # graphs above 112 nodes: key chunks of 112 (2, 2, 3 and 5 chunks; the last ones ragged), statistics combined across them
    ("am_tsp", "tsp", 150, 3, 0, None), ("am_tsp", "tsp", 200, 2, 4, None), ("am_cvrp", "cvrp", 120, 3, 0, None),
    ("am_cvrp", "cvrp", 230, 2, 3, None), ("am_tsp", "tsp", 500, 1, 2, None), ("am_op", "op", 130, 2, 0, None),
    ("am_pctsp", "pctsp", 150, 2, 3, None), ("am_cvrptw", "cvrptw", 120, 2, 0, None),
    ("am_sdvrp", "sdvrp", 130, 2, 0, None), ("am_sdvrp", "sdvrp", 240, 2, 3, None),
    # SDVRP: the dynamic embedding's rank-one terms (remaining demands per step) in all three kernels
    ("am_sdvrp", "sdvrp", 20, 5, 0, None), ("am_sdvrp", "sdvrp", 50, 3, 6, None), ("am_sdvrp", "sdvrp", 100, 2, 3, None),
    # the gather kernel's cooperative bins (the depot of CVRP names > 512 queries of an instance) and its own chunking
    # (more than 24,576 queries of an instance: 250 samples x 100 steps)
    ("am_cvrp", "cvrp", 50, 2, 50, None), ("am_tsp", "tsp", 100, 1, 250, False),
])
def test_native_reevaluation_matches_autograd(cfg, env_name, N, B, ns, ms):
    """eamrl_reeval_forward / _backward (fp32 MFMA kernels) against the PyTorch-autograd re-evaluation of the same actions:
    log-probs within 1e-5 of the native rollout's, every parameter gradient within 1e-4 (relative to the largest ones)."""
    import eam_rl4co_amd as ea
    from eam_rl4co_amd.train import evaluate_log_likelihood, native_reeval_supported

    env = ea.get_env(env_name, generator_params=dict(num_loc=N), seed=N + B)
    torch.manual_seed(N * 13 + B)
    td = env.reset(batch_size=[B]).to(DEV)
    pol = make_policy(cfg)
    kw = dict(num_starts=ns) if ns else {}
    if ms is False:
        kw = dict(num_samples=ns)
    with torch.no_grad():
        out = pol(td, env, phase="train", return_sum_log_likelihood=False, **kw)
    acts = out["actions"]
    assert native_reeval_supported(pol, td["locs"].shape[1])
    w = torch.randn(acts.shape, device=DEV)
    res = []
    for native in (True, False, "from_rollout"):
        pol.zero_grad()
        # "from_rollout": the forward kernel is skipped, the backward recovers the normalisers from the rollout's log-probs
        lp = evaluate_log_likelihood(pol, td, env, acts, num_starts=ns, multistart=ms, native=bool(native),
                                     rollout_logp=out["log_likelihood"] if native == "from_rollout" else None)
        (lp * w).sum().backward()
        res.append((lp.detach(), {k: p.grad.clone() for k, p in pol.named_parameters() if p.grad is not None}))
    tol = 2e-4 if env_name == "cvrptw" else 1e-5          # cvrptw: unscaled inputs (see the oracle tests)
    if td["locs"].shape[1] <= 112:
        assert torch.equal(res[2][0], out["log_likelihood"])
    else:       # key chunks: the forward kernels always run (their statistics feed the backward), the rollout's values are not handed back
        np.testing.assert_allclose(res[2][0].cpu().numpy(), out["log_likelihood"].cpu().numpy(), rtol=0, atol=tol)
    np.testing.assert_allclose(res[0][0].cpu().numpy(), out["log_likelihood"].cpu().numpy(), rtol=0, atol=tol)
    np.testing.assert_allclose(res[0][0].cpu().numpy(), res[1][0].cpu().numpy(), rtol=0, atol=tol)
    assert res[0][1].keys() == res[1][1].keys()
    # norm-wise relative error per parameter <= 1e-4 (parameters whose gradient is rounding noise -- e.g. key biases, true
    # gradient 0 -- are measured against the global gradient norm), and no element further than 5e-4 of the tensor's scale
    gnorm = float(torch.sqrt(sum((g.double() ** 2).sum() for g in res[1][1].values())))
    top = max(float(g.abs().max()) for g in res[1][1].values())
    # cvrptw: unscaled inputs, an ill-conditioned network (see the oracle tests); the error grows with the horizon -- 0.019 / 0.040 /
    # 0.038 / 0.050 at 50 / 100 / 120 / 150 customers, single-chunk and key-chunked kernels alike (forward values agree to 1e-5)
    loose = (300.0 if N < 100 else 600.0) if env_name == "cvrptw" else 1.0
    for which in (0, 2):
        for k in res[1][1]:
            ref, got = res[1][1][k].double(), res[which][1][k].double()
            rel = float((got - ref).norm()) / max(float(ref.norm()), 1e-2 * gnorm)
            assert rel <= 1e-4 * loose, (which, k, rel)
            scale = max(float(ref.abs().max()), 1e-2 * top)
            assert float((got - ref).abs().max()) <= 5e-4 * loose * scale, (which, k)


@pytest.mark.parametrize("rows,E", [(37, 128), (129, 128), (5000, 128), (102400, 128), (300, 64)])
def test_batchnorm_train_autograd_function_matches_torch(rows, E):
    """train._BatchNormTrainFn (eamrl_batchnorm_train forward, eamrl_batchnorm_backward): value and all three gradients of
    F.batch_norm(training=True) -- what the reference's Normalization("batch") runs under policy.train() (nn/ops.py:45-47)."""
    import torch.nn.functional as F
    from eam_rl4co_amd.train import _BatchNormTrainFn

    g = torch.Generator(device=DEV).manual_seed(rows + E)
    x = (torch.randn(rows, E, device=DEV, generator=g) * 1.7 + 0.3).requires_grad_()
    w = (torch.rand(E, device=DEV, generator=g) + 0.5).requires_grad_()
    b = torch.randn(E, device=DEV, generator=g).requires_grad_()
    dy = torch.randn(rows, E, device=DEV, generator=g)
    y = _BatchNormTrainFn.apply(x, w, b, 1e-5)
    y.backward(dy)
    got = [y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone()]
    for t in (x, w, b):
        t.grad = None
    yr = F.batch_norm(x.double(), None, None, w.double(), b.double(), True, 0.0, 1e-5)
    yr.backward(dy.double())
    want = [yr.detach(), x.grad, w.grad, b.grad]
    for name, a, r in zip(("y", "dx", "dgamma", "dbeta"), got, want):
        scale = float(r.abs().max()) + 1e-12
        assert float((a.double() - r.double()).abs().max()) <= 2e-5 * scale + 1e-6, name


@pytest.mark.parametrize("rows,K,strided", [(1, 2, False), (100, 2, False), (513, 3, False), (102400, 2, False), (2000, 6, True),
                                            (300, 4, False)])
def test_init_embedding_linear_autograd_function_matches_torch(rows, K, strided):
    """train._SmallLinearFn: the init embeddings' Linear(K -> 128) -- value = the rollout's tiny-K kernel, weight / bias gradient
    = eamrl_small_linear_wgrad -- against torch's F.linear under autograd."""
    import torch.nn.functional as F
    from eam_rl4co_amd.train import _SmallLinearFn

    g = torch.Generator(device=DEV).manual_seed(rows * 7 + K)
    x = torch.rand(rows, K + (2 if strided else 0), device=DEV, generator=g)[:, :K]
    w = (torch.randn(128, K, device=DEV, generator=g) * 0.5).requires_grad_()
    b = torch.randn(128, device=DEV, generator=g).requires_grad_()
    dy = torch.randn(rows, 128, device=DEV, generator=g)
    y = _SmallLinearFn.apply(x, w, b)
    y.backward(dy)
    got = [y.detach(), w.grad.clone(), b.grad.clone()]
    w.grad = b.grad = None
    yr = F.linear(x.double(), w.double(), b.double())
    yr.backward(dy.double())
    for name, a, r in zip(("y", "dW", "db"), got, (yr.detach(), w.grad, b.grad)):
        scale = float(r.abs().max()) + 1e-12
        assert float((a.double() - r.double()).abs().max()) <= 2e-5 * scale, name


def test_default_am_policy_trains_without_torch_norm_or_linear_kernels(monkeypatch):
    """AttentionModelPolicy's default (batch normalisation) under policy.train(): the gradient graph's BatchNorm and init-embedding
    Linears run on this library's kernels, and give the gradients of the torch ops they replace (EAMRL_TORCH_BATCHNORM=1 /
    EAMRL_TORCH_INIT_EMBED=1) within 1e-3 norm-wise (two fp32 implementations of a backward through three batch norms over
    120 rows; the gradient of a bias in front of a batch norm is a sum that cancels to rounding level).  The reference's own
    gradients are matched to 1e-4 in test_reference_trainer_step_runs_on_the_policy_and_matches_reference_gradients."""
    import eam_rl4co_amd as ea
    from eam_rl4co_amd import train

    for env_name in ("tsp", "cvrp"):
        env = ea.get_env(env_name, generator_params=dict(num_loc=20), seed=5)
        torch.manual_seed(11)
        td = env.reset(batch_size=[6]).to(DEV)
        pol = make_policy("am_" + env_name).train()
        g = torch.Generator().manual_seed(3)
        M = td["locs"].shape[1]
        noise = torch.empty(6, 2 * M + 1, M).exponential_(1, generator=g).to(DEV)
        grads = []
        for torch_ops in ("0", "1"):
            monkeypatch.setenv("EAMRL_TORCH_BATCHNORM", torch_ops)
            monkeypatch.setenv("EAMRL_TORCH_INIT_EMBED", torch_ops)
            for p in pol.parameters():
                p.grad = None
            out = train.reinforce_loss(pol, env, td.clone(), baseline="no", noise=noise)
            out["loss"].backward()
            grads.append({k: (p.grad.clone() if p.grad is not None else None) for k, p in pol.named_parameters()})
        top = max(float(g.norm()) for g in grads[1].values() if g is not None)
        for k in grads[0]:
            a, r = grads[0][k], grads[1][k]
            assert (a is None) == (r is None), k
            if a is not None:       # (biases in front of a batch norm have a gradient that is zero up to rounding: absolute floor)
                assert float((a - r).norm()) <= 1e-3 * float(r.norm()) + 1e-5 * top, k


@pytest.mark.parametrize("env_name,kw", [("tsp", dict(decode_type="sampling", top_k=5)),
                                         ("tsp", dict(decode_type="sampling", top_p=0.8, temperature=2.0)),
                                         ("cvrp", dict(decode_type="sampling", top_k=6, top_p=0.9, temperature=1.5)),
                                         ("tsp", dict(decode_type="multistart_sampling", num_starts=8, select_best=True)),
                                         ("cvrp", dict(decode_type="multistart_greedy", num_starts=6, select_best=True))])
def test_gradients_through_filtering_and_select_best(env_name, kw):
    """policy(..., phase="train") with top-k / top-p filtering (process_logits, utils/decoding.py:111-137,170-176) or
    select_best (decoding.py:419-427) under autograd -- NotImplementedError until round 3: the returned log-likelihood carries
    a grad_fn, the re-evaluation's VALUES equal the rollout's own per-step log-probs (same kept entries, same selected rows)
    and the gradient equals that of the re-evaluated log-likelihood itself."""
    import eam_rl4co_amd as ea
    from eam_rl4co_amd import train

    env = ea.get_env(env_name, generator_params=dict(num_loc=20), seed=8)
    torch.manual_seed(21)
    td = env.reset(batch_size=[5]).to(DEV)
    pol = make_policy("am_" + env_name).eval()       # eval-mode norms: the value comparison is then exact up to rounding
    torch.manual_seed(5)
    out = pol(td.clone(), env, phase="train", return_sum_log_likelihood=False, **kw)
    lp = out["log_likelihood"]
    assert lp.requires_grad and lp.shape[0] == (5 if kw.get("select_best") or "num_starts" not in kw else 5 * kw["num_starts"])
    acts = out["actions"]
    re = train.evaluate_log_likelihood(pol, td, env, acts, num_starts=0, multistart="multistart" in kw["decode_type"],
                                       temperature=kw.get("temperature"), top_k=kw.get("top_k", 0), top_p=kw.get("top_p", 0.0),
                                       native=False)
    assert torch.isfinite(re).all()
    np.testing.assert_allclose(re.detach().cpu().numpy(), lp.detach().cpu().numpy(), rtol=0, atol=2e-5)
    w = torch.linspace(0.5, 1.5, lp.numel(), device=DEV).view_as(lp)
    g1 = torch.autograd.grad((lp * w).sum(), [q for q in pol.parameters() if q.requires_grad], allow_unused=True)
    g2 = torch.autograd.grad((re * w).sum(), [q for q in pol.parameters() if q.requires_grad], allow_unused=True)
    top = max(float(b.norm()) for b in g2 if b is not None)
    assert top > 0
    for a, b in zip(g1, g2):
        assert (a is None) == (b is None)
        if a is not None:
            assert float((a - b).norm()) <= 1e-4 * float(b.norm()) + 1e-6 * top


@pytest.mark.parametrize("env_name,S", [("tsp", 0), ("cvrp", 0), ("tsp", 6)])
def test_symeam_shared_step_restated_runs_on_the_policy(env_name, S):
    """SymEAM.shared_step's training branch (zoo/earl/model.py:535-660; the fork's second trainer) restated line by line around
    SymNCOPolicy, StateAugmentation and evolution_worker, with SymNCO's three losses written out (zoo/symnco/losses.py) -- and
    train.symeam_loss gives the same loss and gradients for the same angles, noise and evolution draws.  (No fixture from the
    reference: its SymNCOPolicy needs torchrl's MLP, absent here -- parity of this trainer step is unpinned beyond the pieces
    it is built from, which are pinned.)"""
    import eam_rl4co_amd as ea
    from eam_rl4co_amd import train
    from eam_rl4co_amd.utils import StateAugmentation

    B, A, N = 4, 4, 20
    env = ea.get_env(env_name, generator_params=dict(num_loc=N), seed=12)
    torch.manual_seed(12)
    td0 = env.reset(batch_size=[B]).to(DEV)
    pol = ea.SymNCOPolicy(env_name=env_name).to(DEV).train()
    sd = pol.state_dict()
    for k, v in golden_weights("am_" + env_name).items():
        sd[k].copy_(torch.from_numpy(v))
    runner = ea.EA(env, dict(num_generations=2, mutation_rate=0.3, crossover_rate=0.8, selection_rate=0.6))
    phi = torch.rand(A * B, generator=torch.Generator().manual_seed(4)) * 4 * math.pi
    aug = StateAugmentation(num_augment=A, phi=phi)
    M = td0["locs"].shape[1]
    noise = torch.empty(A * B * max(S, 1), 3 * M + 1, M, device=DEV).exponential_(1)
    orig_forward = pol.forward

    def with_noise(td_, env_=None, **kw):            # both versions sample with the recorded noise
        if kw.get("actions") is None:
            kw["noise"] = noise
        return orig_forward(td_, env_, **kw)

    pol.forward = with_noise
    try:
        # ---- the reference's lines -----------------------------------------------------------------------------------
        td = aug(td0.clone())
        init_td = td.clone()
        kw = dict(num_starts=S) if S > 1 else {}
        original_out = pol(td, env, phase="train", return_entropy=True, **kw)
        assert original_out["proj_embeddings"].requires_grad and original_out["proj_embeddings"].shape == (A * B, M, 128)
        gen = torch.Generator(device=DEV).manual_seed(9)
        improved_actions, _ = ea.evolution_worker(original_out["actions"], init_td, runner, env, generator=gen)
        if improved_actions.shape[-1] + 1 == original_out["actions"].shape[-1]:
            improved_actions = torch.cat([original_out["actions"][:, :1], improved_actions], -1)
        improved_out = pol(init_td, env, phase="train", actions=improved_actions, **kw)
        reward = torch.cat([ea.unbatchify(original_out["reward"], (A, S)), ea.unbatchify(improved_out["reward"], (A, S))], 0)
        ll = torch.cat([ea.unbatchify(original_out["log_likelihood"], (A, S)),
                        ea.unbatchify(improved_out["log_likelihood"], (A, S))], 0)
        proj = torch.cat([original_out["proj_embeddings"], improved_out["proj_embeddings"]], 0)

        def reinforce(r, l, dim):
            return (-(r - r.mean(dim=dim, keepdim=True)) * l).mean() if r.shape[dim] >= 2 else 0

        loss_ps = reinforce(reward, ll, 1) if S > 1 else 0
        loss_ss = reinforce(reward, ll, -1) if A > 1 else 0
        pe = proj.reshape(proj.shape[0] // A, A, *proj.shape[1:])
        loss_inv = sum(F.cosine_similarity(pe[:, 0], pe[:, i], dim=-1) for i in range(1, A)).mean()
        loss = loss_ps + 1.0 * loss_ss + 0.2 * loss_inv
        pol.zero_grad()
        loss.backward()
        grads = {k: p.grad.clone() for k, p in pol.named_parameters() if p.grad is not None}
        assert all(torch.isfinite(g).all() for g in grads.values())
        assert float(grads["projection_head.2.weight"].norm()) > 0 and float(grads["encoder.init_embedding.init_embed.weight"].norm()) > 0
        # ---- the packaged step -----------------------------------------------------------------------------------------
        pol.zero_grad()
        gen = torch.Generator(device=DEV).manual_seed(9)
        res = train.symeam_loss(pol, env, td0.clone(), runner, num_augment=A, num_starts=S, augment=aug, generator=gen)
    finally:
        pol.forward = orig_forward
    assert torch.equal(res["improved_actions"], improved_actions)
    np.testing.assert_allclose(float(res["loss"].detach()), float(loss.detach()), rtol=1e-5)
    res["loss"].backward()
    gnorm = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())))
    for k, p_ in pol.named_parameters():
        if k in grads:
            rel = float((p_.grad.double() - grads[k].double()).norm()) / max(float(grads[k].double().norm()), 1e-2 * gnorm)
            assert rel <= 1e-4, (k, rel)
