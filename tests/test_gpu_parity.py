"""GPU: the HIP path (through the C ABI) against the CPU oracle and the reference goldens.

Bar: BIT-EXACT against the oracle for every output (integers, masks, and -- because both sides use the
same defined evaluation order -- every float: logits, log-probs, rewards), and identical tours to the
goldens captured from the reference (floats there within the tolerances of test_oracle_golden.py).
"""
import numpy as np
import pytest
import torch

from _util import cfg_for, golden, golden_weights, instance_from_td, instance_of

pytestmark = pytest.mark.gpu

DEV = "cuda"


def t(x, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(DEV)


def bits(a):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_bits_equal(a, b, what=""):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    if a.dtype == np.float32:
        # identical bit patterns, except that +0 == -0 is accepted (sign of zero carries no information here)
        same = (a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))
    else:
        same = a == b
    if not same.all():
        idx = np.argwhere(~same)[:5]
        raise AssertionError(f"{what}: {int((~same).sum())} of {same.size} elements differ, first at {idx.tolist()}: "
                             f"{a[tuple(idx[0])]!r} vs {b[tuple(idx[0])]!r}")


def make_policy(cfg, **kw):
    import eam_rl4co_amd as ea

    env_name = cfg.split("_")[1]          # (am_spctsp: the policy of the stochastic variant = PCTSP's)
    if cfg.startswith("pomo"):
        kw = dict(num_encoder_layers=6, normalization="instance", use_graph_context=False, **kw)
    pol = ea.AttentionModelPolicy(env_name=env_name, **kw).eval()
    sd = pol.state_dict()
    for k, v in golden_weights(cfg).items():
        sd[k].copy_(torch.from_numpy(v))
    return pol.to(DEV)


def make_td(env_name, locs, demand=None):
    """Post-reset TensorDict on the GPU from golden inputs (CVRP locs already include the depot)."""
    import eam_rl4co_amd as ea

    env = ea.get_env(env_name, generator_params=dict(num_loc=locs.shape[1] - (env_name != "tsp")))
    if env_name == "tsp":
        td = ea.TensorDict({"locs": torch.from_numpy(locs)}, batch_size=[locs.shape[0]])
    elif env_name == "cvrptw":      # demand: {"demand", "time_windows" (int32, as the reference keeps them), "durations"}
        td = ea.TensorDict({"locs": torch.from_numpy(locs[:, 1:]), "depot": torch.from_numpy(locs[:, 0]),
                            "demand": torch.from_numpy(demand["demand"]),
                            "time_windows": torch.from_numpy(demand["time_windows"]),
                            "durations": torch.from_numpy(demand["durations"])}, batch_size=[locs.shape[0]])
    elif env_name == "op":          # demand: {"prize" [B,M], "max_length" [B,M]} of the post-reset state
        B = locs.shape[0]
        # generator-style td; the per-node limits and prizes of the fixture replace what reset derives from them
        td = ea.TensorDict({"locs": torch.from_numpy(locs[:, 1:]), "depot": torch.from_numpy(locs[:, 0]),
                            "prize": torch.from_numpy(demand["prize"][:, 1:]),
                            "max_length": torch.full((B,), float(env.generator.max_length))}, batch_size=[B])
        td = env.reset(td)
        assert np.array_equal(td["max_length"].numpy(), demand["max_length"])        # reset == the reference's reset
        return env, td.to(DEV)
    elif env_name in ("pctsp", "spctsp"):       # demand: the dict of prize tensors (tests/_util.py instance_of)
        td = ea.TensorDict({"locs": torch.from_numpy(locs[:, 1:]), "depot": torch.from_numpy(locs[:, 0]),
                            "deterministic_prize": torch.from_numpy(demand["expected_prize"]),
                            "stochastic_prize": torch.from_numpy(np.ascontiguousarray(demand["real_prize"][:, 1:])),
                            "penalty": torch.from_numpy(demand["penalty"][:, 1:])}, batch_size=[locs.shape[0]])
    else:
        td = ea.TensorDict({"locs": torch.from_numpy(locs[:, 1:]), "depot": torch.from_numpy(locs[:, 0]),
                            "demand": torch.from_numpy(demand)}, batch_size=[locs.shape[0]])
    return env, env.reset(td).to(DEV)


# ------------------------------------------------------------------------------------------------------------
def test_library_loads_and_shares_torch_stream():
    from eam_rl4co_amd import _lib, ops

    lib = _lib.load()
    assert lib.eamrl_version() == 100
    x = torch.arange(6, dtype=torch.float32, device=DEV).reshape(3, 2)
    w = torch.tensor([[1.0, 10.0]], device=DEV)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        y = ops.linear(x, w)
    s.synchronize()
    assert y.cpu().reshape(-1).tolist() == [10.0, 32.0, 54.0]


@pytest.mark.parametrize("rows,k,n", [(1, 256, 128), (77, 128, 384), (300, 512, 128), (513, 128, 640), (2500, 128, 128), (50, 6, 7),
                                       (64, 129, 33), (9, 3, 128), (4100, 128, 384), (2049, 128, 512), (6000, 128, 128)])
@pytest.mark.parametrize("variant", ["mfma64", "mfma64_generic_epilogue", "mfma128", "valu"])
def test_linear_is_a_k_ordered_fma_chain(oracle, rows, k, n, variant):
    """MFMA (v_mfma_f32_32x32x2_f32, 64- and 128-row tiles) and VALU kernels all equal the oracle's fmaf chain."""
    valu = int(variant == "valu")
    from eam_rl4co_amd import _lib, ops

    rng = np.random.default_rng(rows * 7 + k)
    x = rng.standard_normal((rows, k)).astype(np.float32)
    W = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    res = rng.standard_normal((rows, n)).astype(np.float32)
    lib = _lib.load()
    lib.eamrl_debug_set(0, valu)
    lib.eamrl_debug_set(4, int(variant == "mfma128"))
    lib.eamrl_debug_set(10, int(variant == "mfma64_generic_epilogue"))
    try:
        y = ops.linear(t(x), t(W), t(b))
        assert_bits_equal(y, oracle.linear(x, W, b), "linear")
        y = ops.linear(t(x), t(W), None, relu=True, residual=t(res))
        assert_bits_equal(y, res + oracle.linear(x, W, None, relu=True), "linear relu+res")
        assert_bits_equal(ops.linear(t(x), t(W), t(b), relu=True), oracle.linear(x, W, b, relu=True), "linear relu")
        assert_bits_equal(ops.linear(t(x), t(W), t(b), residual=t(res)), res + oracle.linear(x, W, b), "linear res")
        if k == n:
            assert_bits_equal(ops.matmul_right(t(x), t(W)), oracle.matmul_right(x, W), "matmul_right")
        g, bt = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
        mu, var = rng.standard_normal(n).astype(np.float32), (rng.random(n) + 0.5).astype(np.float32)
        y = ops.linear(t(x), t(W), t(b), residual=t(res), bn=(t(g), t(bt), t(mu), t(var), 1e-5))
        ref = oracle.batchnorm_eval((res + oracle.linear(x, W, b))[None], g, bt, mu, var)[0]
        assert_bits_equal(y, ref, "linear + fused batch-norm")
    finally:
        lib.eamrl_debug_set(0, 0)
        lib.eamrl_debug_set(4, 0)
        lib.eamrl_debug_set(10, 0)


def test_matmul_right(oracle):
    from eam_rl4co_amd import ops

    rng = np.random.default_rng(3)
    x = rng.standard_normal((333, 128)).astype(np.float32)
    Wt = rng.standard_normal((128, 128)).astype(np.float32)
    assert_bits_equal(ops.matmul_right(t(x), t(Wt)), oracle.matmul_right(x, Wt), "matmul_right")


@pytest.mark.parametrize("B,N", [(3, 20), (2, 100), (2, 101), (1, 127), (2, 128), (1, 129), (1, 256), (1, 257), (1, 301),
                                 (1, 501)])
def test_encoder_attention_norms_mean(oracle, B, N):
    from eam_rl4co_amd import ops

    rng = np.random.default_rng(N)
    E, H = 128, 8
    from eam_rl4co_amd import _lib
    qkv = rng.standard_normal((B, N, 3 * E)).astype(np.float32)
    ref = oracle.mha_encoder(qkv, H)
    assert_bits_equal(ops.mha_encoder(t(qkv), H), ref, "mha_encoder default dispatch")
    for key, variant in ((7, 1), (3, 1)):     # matrix-core kernel (N <= 128), then the plain VALU kernel
        _lib.load().eamrl_debug_set(key, variant)
        try:
            assert_bits_equal(ops.mha_encoder(t(qkv), H), ref, f"mha_encoder debug key {key}")
        finally:
            _lib.load().eamrl_debug_set(key, 0)
    x = rng.standard_normal((B, N, E)).astype(np.float32)
    g, bt = rng.standard_normal(E).astype(np.float32), rng.standard_normal(E).astype(np.float32)
    mu, var = rng.standard_normal(E).astype(np.float32), (rng.random(E) + 0.5).astype(np.float32)
    y = ops.normalize_(t(x).clone(), ops.NORM_BATCH_EVAL, t(g), t(bt), t(mu), t(var), 1e-5)
    assert_bits_equal(y, oracle.batchnorm_eval(x, g, bt, mu, var), "batchnorm eval")
    y = ops.normalize_(t(x).clone(), ops.NORM_INSTANCE, t(g), t(bt), eps=1e-5)
    assert_bits_equal(y, oracle.instancenorm(x, g, bt), "instance norm")
    assert_bits_equal(ops.mean_nodes(t(x)), oracle.mean_nodes(x), "mean_nodes")


ENC_CASES = ["tsp20_greedy", "cvrp20_greedy", "pomo_tsp20_multistart_sampling", "tsp100_greedy", "cvrp100_greedy"]


@pytest.mark.parametrize("name", ENC_CASES)
def test_encoder_and_cache_bit_exact(oracle, name):
    fx = golden(name)
    cfg = cfg_for(fx)
    env_name = str(fx["env_name"])
    pol = make_policy(cfg)
    env, td = make_td(env_name, fx["locs"], instance_of(fx))
    with torch.no_grad():
        emb, init_h = pol.encoder(td)
        cache = pol.decoder._precompute_cache(emb)
    sd = golden_weights(cfg)
    o_init, o_emb = oracle.encode(sd, env_name, fx["locs"], instance_of(fx))
    assert_bits_equal(init_h, o_init, "init embedding")
    assert_bits_equal(emb, o_emb, "encoder output")
    oc = oracle.precompute(sd, env_name, o_emb, use_graph_context=pol.decoder.use_graph_context)
    for nm in ("K", "V", "L", "Pa", "Lp") + (("Pb",) if env_name == "tsp" else ()):
        assert_bits_equal(cache.view(nm), oc[nm], f"cache {nm}")
    assert_bits_equal(cache.cvec, oc["cvec"], "cvec")
    if oc["gctx"] is not None:
        assert_bits_equal(cache.gctx, oc["gctx"], "graph context")
    # and within tolerance of what the reference computed
    if "embeddings" in fx:
        np.testing.assert_allclose(emb.cpu().numpy(), fx["embeddings"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(cache.glimpse_key.cpu().numpy(), fx["glimpse_key"], rtol=0, atol=1e-5)


STEP_CASES = ["tsp20_greedy", "tsp100_greedy", "cvrp20_greedy", "cvrp100_greedy", "cvrp100_sampling",
              "tsp20_multistart_greedy", "cvrp20_multistart_greedy", "pomo_tsp20_multistart_sampling",
              "sdvrp20_greedy", "sdvrp20_sampling", "sdvrp50_greedy", "sdvrp20_multistart_greedy",
              "pctsp20_greedy", "pctsp20_sampling", "pctsp100_sampling", "pctsp20_multistart_greedy",
              "op20_greedy", "op20_sampling", "op100_sampling", "op20_multistart_greedy",
              "cvrptw20_greedy", "cvrptw20_sampling", "cvrptw100_sampling", "cvrptw20_multistart_greedy"]


@pytest.mark.parametrize("name", STEP_CASES)
def test_decode_step_api_bit_exact_every_step(oracle, name):
    """Step API (decode kernel + stand-alone env kernels) replaying the reference's actions: logits,
    log-probs, selected action, masks and all state equal the oracle at EVERY step."""
    from eam_rl4co_amd import ops
    from eam_rl4co_amd.policy import _env_step_, state_from_td

    fx = golden(name)
    cfg = cfg_for(fx)
    env_name = str(fx["env_name"])
    ns = int(fx["num_starts"])
    pol = make_policy(cfg)
    env, td = make_td(env_name, fx["locs"], instance_of(fx))
    sd = golden_weights(cfg)
    with torch.no_grad():
        emb, _ = pol.encoder(td)
        cache = pol.decoder._precompute_cache(emb)
    _, o_emb = oracle.encode(sd, env_name, fx["locs"], instance_of(fx))
    oc = oracle.precompute(sd, env_name, o_emb, use_graph_context=pol.decoder.use_graph_context)
    ost = oracle.State(env_name, fx["locs"], instance_of(fx), num_starts=ns)
    st = state_from_td(env_name, td, ns)
    actions = fx["actions"]
    col = 0
    if ns > 1:
        a0 = np.ascontiguousarray(actions[:, 0])
        ost.step(a0)
        _env_step_(st, t(a0))
        col = 1
    has_noise = "noise" in fx
    for step in range(int(fx["n_decoder_steps"])):
        nz = np.ascontiguousarray(fx["noise"][:, step]) if has_noise else None
        mode = "sampling" if has_noise else "greedy"
        oa, olp, ologits, ologp = oracle.decode_step(ost, oc, mode, noise=nz, want_all=True)
        a, lp, alls, logits, status = ops.decode_step(st, cache, mode, noise=None if nz is None else t(nz),
                                                      want_logprobs=True, want_logits=True)
        assert int(status.item()) == 0
        assert_bits_equal(a, oa, f"action step {step}")
        assert np.array_equal(oa, actions[:, col + step]), "oracle left the reference's tour"
        feas = ost.mask.astype(bool)
        assert_bits_equal(torch.where(t(feas), logits, torch.zeros_like(logits)), np.where(feas, ologits, 0),
                          f"logits step {step}")
        assert_bits_equal(alls, ologp, f"logprobs step {step}")
        assert_bits_equal(lp, olp, f"logp step {step}")
        # env transition through the stand-alone kernels
        ost.step(oa)
        _env_step_(st, a)
        if env_name == "tsp":
            assert_bits_equal(st.first, ost.first, "first")
            assert_bits_equal(st.istep, ost.istep, "i")
        elif env_name in ("cvrp", "cvrptw"):
            assert_bits_equal(st.visited, ost.visited, "visited")
            assert_bits_equal(st.used, ost.used, "used")
            if env_name == "cvrptw":
                assert_bits_equal(st.time, ost.time, "current_time")
        elif env_name in ("pctsp", "op"):
            assert_bits_equal(st.visited.to(torch.uint8), ost.visited, "visited")
            assert_bits_equal(st.used, ost.used, "collected prize / tour length")
            assert_bits_equal(st.istep, ost.istep, "i")
        else:
            assert_bits_equal(st.rem, ost.rem, "remaining demand")
            assert_bits_equal(st.used, ost.used, "used")
        assert_bits_equal(st.mask.to(torch.uint8), ost.mask, f"mask step {step}")
        assert_bits_equal(st.cur, ost.cur, "cur")
        assert_bits_equal(st.done.to(torch.uint8), ost.done, "done")
    assert bool(st.done.all())


POLICY_CASES = ["tsp20_greedy", "tsp20_sampling", "tsp20_evaluate", "tsp20_multistart_greedy", "tsp100_greedy",
                "tsp100_sampling", "cvrp20_greedy", "cvrp20_sampling", "cvrp20_evaluate", "cvrp20_multistart_greedy",
                "cvrp100_greedy", "cvrp100_sampling", "pomo_tsp20_multistart_sampling",
                "tsp50_greedy", "cvrp50_sampling", "tsp200_greedy", "cvrp200_greedy", "pomo_cvrp20_multistart_greedy",
                "cvrp20_sampling_temp", "tsp20_greedy_noclip",
                "tsp20_sampling_topk5", "tsp20_sampling_topp", "cvrp20_sampling_topk_topp", "tsp100_greedy_topk",
                "sdvrp20_greedy", "sdvrp20_sampling", "sdvrp50_greedy", "sdvrp20_multistart_greedy",
                "pctsp20_greedy", "pctsp20_sampling", "pctsp50_greedy", "pctsp100_sampling", "pctsp20_multistart_greedy",
                "spctsp20_sampling", "spctsp50_greedy",
                "op20_greedy", "op20_sampling", "op50_greedy", "op100_sampling", "op20_multistart_greedy",
                "cvrptw20_greedy", "cvrptw20_sampling", "cvrptw50_greedy", "cvrptw100_sampling",
                "cvrptw20_multistart_greedy"]


@pytest.mark.parametrize("stream_kernel", [0, 1])
@pytest.mark.parametrize("name", POLICY_CASES)
def test_policy_forward_reproduces_reference_tours(oracle, name, stream_kernel):
    """AttentionModelPolicy.forward (whole rollout in one launch) on the golden inputs: tours identical to
    the reference, floats bit-equal to the oracle and within tolerance of the reference."""
    from eam_rl4co_amd import _lib

    fx = golden(name)
    cfg = cfg_for(fx)
    env_name = str(fx["env_name"])
    ns = int(fx["num_starts"])
    pol = make_policy(cfg)
    env, td = make_td(env_name, fx["locs"], instance_of(fx))
    decode_type = str(fx["decode_type"])
    kw = {}
    if name.endswith("evaluate"):
        kw["actions"] = t(fx["actions"])
        decode_type = "sampling"
    if ns > 1:
        kw["num_starts"] = ns
        if "multistart" not in decode_type:
            decode_type = "multistart_" + decode_type
    if "noise" in fx:
        kw["noise"] = t(fx["noise"])
    for k in ("temperature", "tanh_clipping", "top_p"):
        if "decode_kw_" + k in fx:
            kw[k] = float(fx["decode_kw_" + k])
    if "decode_kw_top_k" in fx:
        kw["top_k"] = int(fx["decode_kw_top_k"])
    okw = {}
    if env_name == "op" and ns > 1:     # OP may resample its start nodes at random: replay the recorded ones
        starts = t(np.ascontiguousarray(fx["actions"][:, 0]))
        kw["select_start_nodes_fn"] = lambda td_, env_, n_: starts
        okw["start_nodes"] = fx["actions"][:, 0]
    lib = _lib.load()
    lib.eamrl_debug_set(1, stream_kernel)
    try:
        out = pol(td, env, phase="test", decode_type=decode_type, return_sum_log_likelihood=False, **kw)
    finally:
        lib.eamrl_debug_set(1, 0)
    assert_bits_equal(out["actions"], fx["actions"], "tours vs reference")
    np.testing.assert_allclose(out["reward"].cpu().numpy(), fx["reward"], rtol=1e-6, atol=0)
    # (CVRPTW's unscaled inputs make the network ill-conditioned: looser bound vs the reference, see test_oracle_golden)
    np.testing.assert_allclose(out["log_likelihood"].cpu().numpy(), fx["logp_steps"], rtol=0,
                               atol=2e-4 if env_name == "cvrptw" else 1e-5)
    o = oracle.policy_rollout(golden_weights(cfg), env_name, fx["locs"], instance_of(fx),
                              decode_type=decode_type if "actions" not in kw else "evaluate", num_starts=ns,
                              noise=fx.get("noise"), given=fx["actions"] if "actions" in kw else None,
                              use_graph_context=pol.decoder.use_graph_context,
                              clip=kw.get("tanh_clipping", 10.0), temp=kw.get("temperature", 1.0),
                              top_k=kw.get("top_k", 0), top_p=kw.get("top_p", 0.0), **okw)
    assert_bits_equal(out["log_likelihood"], o["logp_steps"], "per-step logp vs oracle")
    assert_bits_equal(out["reward"], o["reward"], "reward vs oracle")


@pytest.mark.parametrize("name", ["env_tsp20_random", "env_cvrp20_random", "env_cvrp100_random", "env_sdvrp20_random",
                                  "env_pctsp20_random", "env_spctsp20_random", "env_op20_random",
                                  "env_op50_random", "env_cvrptw20_random", "env_cvrptw50_random"])
def test_env_api_matches_reference_state_machine(name):
    """env.reset / env.step / env.get_reward through the RL4COEnvBase API against the reference's recorded states."""
    import eam_rl4co_amd as ea

    fx = golden(name)
    env_name = str(fx["env_name"])
    env = ea.get_env(env_name, generator_params=dict(num_loc=int(fx["num_loc"])))
    gen = {k[4:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("gen_")}
    td = env.reset(ea.TensorDict(gen, batch_size=[fx["gen_locs"].shape[0]])).to(DEV)
    assert_bits_equal(td["action_mask"], fx["reset_action_mask"], "reset mask")
    T = fx["step_action"].shape[1]
    for step in range(T):
        td.set("action", t(fx["step_action"][:, step]))
        td = env.step(td)["next"]
        assert_bits_equal(td["action_mask"], fx["step_action_mask"][:, step], f"mask {step}")
        assert_bits_equal(td["done"], fx["step_done"][:, step], f"done {step}")
        assert td["done"].shape == (fx["gen_locs"].shape[0],) and td["reward"].dtype == torch.bool
        assert_bits_equal(td["current_node"], fx["step_current_node"][:, step], f"cur {step}")
        if env_name == "tsp":
            assert_bits_equal(td["first_node"], fx["step_first_node"][:, step], "first")
            assert_bits_equal(td["i"], fx["step_i"][:, step], "i")
        elif env_name == "op":
            for k in ("visited", "tour_length", "current_total_prize", "i"):
                assert_bits_equal(td[k], fx["step_" + k][:, step], k)
            assert_bits_equal(env.get_action_mask(td), fx["step_action_mask"][:, step], "get_action_mask")
        elif env_name in ("pctsp", "spctsp"):
            for k in ("visited", "cur_total_prize", "cur_total_penalty", "i"):
                assert_bits_equal(td[k], fx["step_" + k][:, step], k)
            assert_bits_equal(env.get_action_mask(td), fx["step_action_mask"][:, step], "get_action_mask")
        else:
            if env_name in ("cvrp", "cvrptw"):
                assert_bits_equal(td["visited"], fx["step_visited"][:, step], "visited")
                if env_name == "cvrptw":
                    assert_bits_equal(td["current_time"], fx["step_current_time"][:, step], "current_time")
            else:
                assert_bits_equal(td["demand_with_depot"], fx["step_demand_with_depot"][:, step], "remaining demand")
            assert_bits_equal(td["used_capacity"], fx["step_used_capacity"][:, step], "used")
            assert_bits_equal(env.get_action_mask(td), fx["step_action_mask"][:, step], "get_action_mask")
    reward = env.get_reward(td, t(fx["step_action"]))
    np.testing.assert_allclose(reward.cpu().numpy(), fx["reward"], rtol=1e-6, atol=0)
    bad = fx["step_action"].copy()
    if env_name == "cvrptw":    # CVRP asserts + the time-window replay (cvrptw/env.py:203-227)
        N = int(fx["num_loc"])
        locs = np.concatenate([fx["gen_depot"][:, None], fx["gen_locs"]], 1).astype(np.float64)
        tw = fx["gen_time_windows"].astype(np.float64)
        rows = []
        for b in range(bad.shape[0]):       # i then j, where j's window is over before one can get there from i
            d = np.linalg.norm(locs[b][:, None] - locs[b][None], axis=-1)
            late = (tw[b, :, 0][:, None] + d > tw[b, :, 1][None] + 1.0)
            late[0, :] = late[:, 0] = False
            i, j = np.argwhere(late)[0]
            rest = [k for k in range(1, N + 1) if k not in (i, j)]
            rows.append([i, j, 0] + [x for k in rest for x in (k, 0)])
        with pytest.raises(AssertionError, match="vehicle cannot start service before deadline"):
            env.get_reward(td, t(np.array(rows, dtype=np.int64)))
        bad[0, np.nonzero(bad[0])[0][0]] = 0
        with pytest.raises(AssertionError, match="Invalid tour"):
            env.get_reward(td, t(bad))
        return
    if env_name == "op":        # the reference's own asserts (op/env.py:179-212)
        rows = np.nonzero(bad[:, 1] != 0)[0]
        bad[rows[0], 0] = bad[rows[0], 1]
        with pytest.raises(AssertionError, match="Duplicates"):
            env.get_reward(td, t(bad))
        far = np.tile(np.arange(1, int(fx["num_loc"]) + 1, dtype=np.int64), (bad.shape[0], 1))
        with pytest.raises(AssertionError, match="Max length exceeded"):
            env.get_reward(td, t(far))
        return
    if env_name in ("pctsp", "spctsp"):     # the reference's own asserts (pctsp/env.py:189-205)
        bad[0, 1] = bad[0, 0]
        with pytest.raises(AssertionError, match="Duplicates"):
            env.get_reward(td, t(bad))
        early = np.zeros_like(bad); early[:, 0] = fx["step_action"][:, 0]
        with pytest.raises(AssertionError, match="Total prize does not satisfy min total prize"):
            env.get_reward(td, t(early))
        return
    if env_name == "sdvrp":     # the reference's own asserts (sdvrp/env.py:148-171)
        with pytest.raises(AssertionError, match="All demand must be satisfied"):
            env.get_reward(td, t(np.ascontiguousarray(bad[:, :5])))      # the tours stop early
        twice = np.concatenate([np.zeros((bad.shape[0], 2), np.int64), fx["step_action"]], 1)
        with pytest.raises(AssertionError, match="Cannot visit depot twice"):
            env.get_reward(td, t(twice))
        return
    bad[0, -1] = bad[0, 0] if env_name == "tsp" else bad[0, np.nonzero(bad[0])[0][0]]
    with pytest.raises(AssertionError, match="Invalid tour"):
        env.get_reward(td, t(bad))


def test_random_policy_rollout_helper_shapes():
    """The reference's own env test (tests/test_envs.py:62-65): rollout(env, reset, random_policy) -> reward [B]."""
    import eam_rl4co_amd as ea

    for name in ("tsp", "cvrp"):
        env = ea.get_env(name, generator_params=dict(num_loc=20))
        td = env.reset(batch_size=[2]).to(DEV)
        reward, td, actions = ea.rollout(env, td, ea.random_policy)
        assert reward.shape == (2,)


@pytest.mark.parametrize("env_name,N,B,mode", [("tsp", 100, 1024, "greedy"), ("cvrp", 100, 1024, "sampling"),
                                                ("tsp", 20, 128, "greedy"), ("cvrp", 500, 16, "greedy"),
                                                ("sdvrp", 100, 256, "sampling"), ("sdvrp", 200, 32, "greedy"),
                                                ("pctsp", 100, 256, "sampling"), ("pctsp", 200, 32, "greedy"),
                                                ("op", 100, 256, "sampling"), ("op", 200, 32, "greedy"),
                                                ("cvrptw", 100, 256, "sampling"), ("cvrptw", 200, 32, "greedy")])
def test_full_size_rollout_equals_oracle(oracle, env_name, N, B, mode):
    """BASELINE.json configs at full size (C2, C3, C1, C5 with a reduced batch so the CPU oracle finishes in
    seconds): tours bit-identical to the oracle, plus size-independent properties."""
    import eam_rl4co_amd as ea

    cfg = "am_" + env_name
    pol = make_policy(cfg)
    env = ea.get_env(env_name, generator_params=dict(num_loc=N), seed=1234)
    td_cpu = env.reset(batch_size=[B])
    td = td_cpu.to(DEV)
    locs = td_cpu["locs"].numpy()
    demand = td_cpu["demand"].numpy() if env_name in ("cvrp", "sdvrp") else None
    if env_name == "cvrptw":
        demand = {k: td_cpu[k].numpy() for k in ("demand", "time_windows", "durations")}
    if env_name == "pctsp":
        demand = {k: td_cpu[k].numpy() for k in ("expected_prize", "real_prize", "penalty", "prize_required")}
    if env_name == "op":
        demand = {k: td_cpu[k].numpy() for k in ("prize", "max_length")}
    M = locs.shape[1]
    kw = {}
    noise = None
    if mode == "sampling":
        g = torch.Generator().manual_seed(7)
        noise = torch.empty(B, (3 if env_name == "sdvrp" else 2) * M + 1, M).exponential_(1, generator=g)
        kw["noise"] = noise.to(DEV)
    out = pol(td, env, phase="test", decode_type=mode, return_sum_log_likelihood=False, **kw)
    acts = out["actions"].cpu().numpy()
    # properties: valid tours (the env's own check ran inside get_reward), reward == recomputed closed length
    if env_name == "tsp":
        assert (np.sort(acts, 1) == np.arange(N)).all()
        pts = np.take_along_axis(locs, acts[..., None].repeat(2, -1), 1).astype(np.float64)
    elif env_name == "op":          # customers at most once; the closed tour fits; reward = collected prize
        for b, row in enumerate(acts):
            nz = row[row != 0]
            assert len(set(nz)) == len(nz)
        pts = np.take_along_axis(locs, acts[..., None].repeat(2, -1), 1).astype(np.float64)
        length = np.linalg.norm(np.roll(pts, -1, 1) - pts, axis=-1).sum(1)
        assert (length <= float(env.generator.max_length) + 1e-4).all()
        want = np.take_along_axis(demand["prize"].astype(np.float64), acts, 1).sum(1)
        np.testing.assert_allclose(out["reward"].cpu().numpy(), want, rtol=1e-5)
    elif env_name == "pctsp":       # customers at most once; enough prize or everyone visited; reward by the definition
        for b, row in enumerate(acts):
            nz = row[row != 0]
            assert len(set(nz)) == len(nz)
            assert demand["real_prize"][b, nz].sum() >= 1 - 1e-5 or len(nz) == N
        pts = np.take_along_axis(locs, acts[..., None].repeat(2, -1), 1).astype(np.float64)
        pts = np.concatenate([locs[:, :1].astype(np.float64), pts], 1)
        length = np.linalg.norm(np.roll(pts, -1, 1) - pts, axis=-1).sum(1)
        pen = demand["penalty"].astype(np.float64)
        want = np.take_along_axis(pen, acts, 1).sum(1) - (length + pen[:, 1:].sum(1))
        np.testing.assert_allclose(out["reward"].cpu().numpy(), want, rtol=1e-5)
    else:
        if env_name in ("cvrp", "cvrptw"):
            srt = np.sort(acts, 1)
            assert (srt[:, -N:] == np.arange(1, N + 1)).all() and (srt[:, :-N] == 0).all()
        else:       # split deliveries: every customer at least once, and everything delivered
            assert all(set(range(1, N + 1)) <= set(row) for row in acts)
            assert float(pol._last_td["demand_with_depot"].abs().max()) == 0.0
        pts = np.take_along_axis(locs, acts[..., None].repeat(2, -1), 1).astype(np.float64)
        pts = np.concatenate([locs[:, :1].astype(np.float64), pts], 1)
    if env_name not in ("pctsp", "op"):
        length = np.linalg.norm(np.roll(pts, -1, 1) - pts, axis=-1).sum(1)
        np.testing.assert_allclose(-out["reward"].cpu().numpy(), length, rtol=2e-6)
    o = oracle.policy_rollout(golden_weights(cfg), env_name, locs, demand, decode_type=mode,
                              noise=None if noise is None else noise.numpy())
    T = o["actions"].shape[1]
    assert acts.shape[1] == T
    assert_bits_equal(acts, o["actions"], "tours vs oracle")
    assert_bits_equal(out["log_likelihood"], o["logp_steps"], "logp vs oracle")
    assert_bits_equal(out["reward"], o["reward"], "reward vs oracle")


@pytest.mark.parametrize("env_name,N,B", [
    # tiny graphs (a 2-city tour, a single customer) and a batch of one
    ("tsp", 2, 1), ("tsp", 3, 5), ("tsp", 5, 1), ("cvrp", 1, 3), ("cvrp", 2, 1), ("cvrp", 3, 4), ("pctsp", 2, 2), ("op", 3, 2),
    # the size boundaries of the kernels: fused encoder / MFMA kernels up to 112 nodes, register-resident decode up to 128,
    # the streaming kernel beyond
    ("tsp", 111, 2), ("tsp", 112, 2), ("tsp", 113, 2), ("tsp", 128, 2), ("tsp", 129, 1),
    ("cvrp", 110, 2), ("cvrp", 111, 2), ("cvrp", 112, 1), ("cvrp", 127, 1), ("cvrp", 128, 1),
])
@pytest.mark.parametrize("mode", ["greedy", "sampling", "multistart_greedy"])
def test_edge_sizes_match_oracle(oracle, env_name, N, B, mode):
    """Smallest graphs, a batch of one, and the node counts at which the rollout changes kernels: bit-identical to the oracle."""
    import eam_rl4co_amd as ea

    if env_name in ("pctsp", "op") and mode == "multistart_greedy":
        pytest.skip("forced start nodes can be infeasible for these envs (the reference's validity check fails as well)")
    cfg = "am_" + env_name
    pol = make_policy(cfg)
    env = ea.get_env(env_name, generator_params=dict(num_loc=N), seed=N * 3 + B)
    torch.manual_seed(N)
    td_cpu = env.reset(batch_size=[B])
    locs = td_cpu["locs"].numpy()
    demand = instance_from_td(env_name, td_cpu)
    M = locs.shape[1]
    S = min(N, 7) if mode.startswith("multistart") else 0
    if S == 1:
        pytest.skip("a single start is not a multistart batch")
    kw, noise = (dict(num_starts=S) if S else {}), None
    if mode == "sampling":
        noise = torch.empty(B, 3 * M + 1, M).exponential_(1, generator=torch.Generator().manual_seed(N + B))
        kw["noise"] = noise.to(DEV)
    out = pol(td_cpu.to(DEV), env, phase="test", decode_type=mode, return_sum_log_likelihood=False, **kw)
    o = oracle.policy_rollout(golden_weights(cfg), env_name, locs, demand, decode_type=mode, num_starts=S,
                              noise=None if noise is None else noise.numpy())
    assert out["actions"].shape == o["actions"].shape
    assert_bits_equal(out["actions"], o["actions"], "tours")
    assert_bits_equal(out["log_likelihood"], o["logp_steps"], "logp")
    assert_bits_equal(out["reward"], o["reward"], "reward")


def test_entropy_and_stepwise_path_agree_with_single_launch():
    import eam_rl4co_amd as ea

    fx = golden("cvrp20_greedy")
    pol = make_policy("am_cvrp")
    env, td = make_td("cvrp", fx["locs"], fx["demand"])
    a = pol(td.clone(), env, phase="test", decode_type="greedy")
    b = pol(td.clone(), env, phase="test", decode_type="greedy", return_entropy=True)
    assert_bits_equal(a["actions"], b["actions"], "actions")
    assert_bits_equal(a["log_likelihood"], b["log_likelihood"], "ll")
    assert torch.isfinite(b["entropy"]).all() and (b["entropy"] >= 0).all()


@pytest.mark.parametrize("env_name,N,B,kw", [
    ("tsp", 20, 6, dict(decode_type="sampling")), ("tsp", 100, 4, dict(decode_type="multistart_sampling", num_starts=10)),
    ("tsp", 50, 3, dict(decode_type="greedy", num_samples=4, multisample=True)),
    ("cvrp", 20, 6, dict(decode_type="sampling")), ("cvrp", 100, 3, dict(decode_type="multistart_greedy", num_starts=6)),
    ("pctsp", 20, 5, dict(decode_type="sampling")), ("op", 20, 5, dict(decode_type="greedy")),
    ("cvrptw", 20, 4, dict(decode_type="sampling")), ("tsp", 112, 2, dict(decode_type="sampling", temperature=1.7)),
    ("sdvrp", 20, 5, dict(decode_type="sampling")), ("sdvrp", 50, 3, dict(decode_type="multistart_sampling", num_starts=5)),
    # graphs above 112 nodes: the key-chunked forward kernels
    ("tsp", 150, 3, dict(decode_type="sampling")), ("cvrp", 200, 2, dict(decode_type="multistart_sampling", num_starts=4)),
])
def test_return_entropy_single_launch_equals_stepwise(env_name, N, B, kw):
    """`return_entropy=True` (what the fork's EAM trainer passes, earl/model.py:153-155): the rollout stays one launch and
    the entropy comes from one teacher-forced pass over the finished tours; same tours, log-likelihood and -- within 1e-4
    relative -- the entropy of the host-driven step loop over full log-prob vectors (EAMRL_ENTROPY_STEPWISE=1)."""
    import os

    import eam_rl4co_amd as ea

    env = ea.get_env(env_name, generator_params=dict(num_loc=N), seed=N)
    torch.manual_seed(N + B)
    td = env.reset(batch_size=[B]).to(DEV)
    pol = make_policy("am_" + env_name)
    R = B * max(kw.get("num_starts", 1), kw.get("num_samples", 1))
    noise = None
    if "sampling" in kw["decode_type"]:
        noise = torch.empty(R, 3 * td["locs"].shape[1] + 1, td["locs"].shape[1], device=DEV).exponential_(1)
    outs = []
    for stepwise in ("0", "1"):
        os.environ["EAMRL_ENTROPY_STEPWISE"] = stepwise
        try:
            outs.append(pol(td.clone(), env, phase="test", return_entropy=True, **(dict(noise=noise) if noise is not None else {}),
                            **kw))
        finally:
            os.environ.pop("EAMRL_ENTROPY_STEPWISE", None)
    a, b = outs
    assert torch.equal(a["actions"], b["actions"]) and torch.equal(a["reward"], b["reward"])
    assert_bits_equal(a["log_likelihood"], b["log_likelihood"].cpu().numpy(), "log-likelihood")
    ea_, eb = a["entropy"].cpu().numpy(), b["entropy"].cpu().numpy()
    assert ea_.shape == (R,) and (eb > 0).all()
    np.testing.assert_allclose(ea_, eb, rtol=1e-4, atol=1e-4)


def test_status_flags_mirror_reference_asserts():
    """NaN weights -> 'Logits contain NaNs'; teacher forcing an already visited node -> 'infeasible action selected'."""
    import eam_rl4co_amd as ea

    fx = golden("tsp20_greedy")
    pol = make_policy("am_tsp")
    env, td = make_td("tsp", fx["locs"])
    bad = fx["actions"].copy()
    bad[:, 5] = bad[:, 4]
    with pytest.raises(AssertionError, match="infeasible action selected"):
        pol(td.clone(), env, phase="test", actions=t(bad), calc_reward=False)
    with torch.no_grad():
        pol.decoder.project_node_embeddings.weight[300, 0] = float("nan")
    with pytest.raises(AssertionError, match="Logits contain NaNs"):
        pol(td.clone(), env, phase="test", decode_type="greedy")


def test_full_size_pomo_multistart_equals_oracle(oracle):
    """BASELINE.json configs[3] shape per instance: POMO policy, TSP-100, 100 starts, multistart sampling (64 instances, so
    that the CPU oracle finishes in about half a minute): every tour, log-prob and reward bit-identical to the oracle."""
    import eam_rl4co_amd as ea

    N, B, S = 100, 64, 100          # (64 instances since round 3: ~30 s of oracle on the GPU box's cores)
    pol = make_policy("pomo_tsp")
    env = ea.get_env("tsp", generator_params=dict(num_loc=N), seed=99)
    td_cpu = env.reset(batch_size=[B])
    noise = torch.empty(B * S, N - 1, N).exponential_(1, generator=torch.Generator().manual_seed(9))
    out = pol(td_cpu.to(DEV), env, phase="train", decode_type="multistart_sampling", num_starts=S, noise=noise.to(DEV),
              return_sum_log_likelihood=False)
    o = oracle.policy_rollout(golden_weights("pomo_tsp"), "tsp", td_cpu["locs"].numpy(), None,
                              decode_type="multistart_sampling", num_starts=S, noise=noise.numpy(), use_graph_context=False)
    assert_bits_equal(out["actions"], o["actions"], "tours vs oracle")
    assert_bits_equal(out["log_likelihood"], o["logp_steps"], "logp vs oracle")
    assert_bits_equal(out["reward"], o["reward"], "reward vs oracle")
    assert (np.sort(out["actions"].cpu().numpy(), 1) == np.arange(N)).all()
    assert (out["actions"][:, 0].cpu().numpy() == np.repeat(np.arange(S), B)).all()      # start nodes in (s b) order


def test_full_size_cvrp500_properties():
    """BASELINE.json configs[4] at its full size (CVRP-500, batch 512, greedy, streaming decode kernel + tiled
    attention): size-independent properties -- every customer exactly once, routes within capacity (the env's own
    validity check), reward equal to the recomputed closed length, and a second run reproduces the first bit for bit."""
    import eam_rl4co_amd as ea

    N, B = 500, 512
    pol = make_policy("am_cvrp")
    env = ea.get_env("cvrp", generator_params=dict(num_loc=N), seed=1234)
    td = env.reset(batch_size=[B]).to(DEV)
    a = pol(td.clone(), env, phase="test", decode_type="greedy")
    b = pol(td.clone(), env, phase="test", decode_type="greedy")
    assert torch.equal(a["actions"], b["actions"]) and torch.equal(a["reward"], b["reward"])
    acts = a["actions"]
    srt = acts.sort(1).values
    assert (srt[:, -N:] == torch.arange(1, N + 1, device=DEV)).all() and (srt[:, :-N] == 0).all()
    env.check_solution_validity(td, acts)
    locs = td["locs"].double()
    pts = torch.cat([locs[:, :1], locs.gather(1, acts[..., None].expand(-1, -1, 2))], 1)
    length = (pts.roll(-1, 1) - pts).norm(dim=-1).sum(1)
    np.testing.assert_allclose(-a["reward"].cpu().numpy(), length.cpu().numpy(), rtol=2e-6)


# ------------------------------------------------------------------------------------------------------------
# sampling without a noise tensor: the counter-based Exp(1) generator
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("R,T,M,seed", [(3, 5, 20, 1), (7, 9, 101, 2 ** 40 + 12345), (2, 3, 7, 0), (4, 2, 128, 2 ** 63 + 5)])
def test_exp1_noise_kernel_equals_oracle(oracle, R, T, M, seed):
    from eam_rl4co_amd import ops

    got = ops.exp1_noise(seed, R, T, M, DEV)
    assert_bits_equal(got, oracle.exp1_noise(seed, R, T, M), "Exp(1) draws")
    sd = torch.tensor([0x1234567], dtype=torch.int64, device=DEV)
    assert_bits_equal(ops.exp1_noise(seed, R, T, M, DEV, seed_dev=sd), oracle.exp1_noise(seed ^ 0x1234567, R, T, M), "seed word")


def test_exp1_noise_distribution():
    from eam_rl4co_amd import ops

    x = ops.exp1_noise(77, 256, 64, 100, DEV).double()
    assert float(x.min()) > 0 and torch.isfinite(x).all()
    assert abs(float(x.mean()) - 1.0) < 5e-3 and abs(float(x.var()) - 1.0) < 2e-2          # Exp(1): mean 1, variance 1
    assert abs(float((x > 1.0).double().mean()) - np.exp(-1.0)) < 3e-3
    a, b = ops.exp1_noise(77, 8, 4, 100, DEV), ops.exp1_noise(78, 8, 4, 100, DEV)
    assert not torch.equal(a, b) and torch.equal(a, x[:8, :4].float())                      # counter-based: prefix-stable


@pytest.mark.parametrize("cfg,env_name,N,B,S", [("pomo_tsp", "tsp", 20, 5, 20), ("pomo_tsp", "tsp", 50, 3, 50),
                                                ("am_tsp", "tsp", 100, 2, 0), ("am_cvrp", "cvrp", 20, 4, 0),
                                                ("am_cvrp", "cvrp", 50, 3, 6)])
def test_seeded_sampling_equals_the_noise_tensor_path(cfg, env_name, N, B, S):
    """policy(..., decode_type=sampling) without `noise`: the draws come from (seed, row, step, node) -- inside the MFMA
    start-sharing kernel for TSP multistart, through a scratch tensor elsewhere -- and equal, bit for bit, the rollout that is
    fed the tensor eamrl_exp1_noise writes for the same seed."""
    import eam_rl4co_amd as ea
    from eam_rl4co_amd import ops

    env = ea.get_env(env_name, generator_params=dict(num_loc=N), seed=N)
    td = env.reset(batch_size=[B]).to(DEV)
    pol = make_policy(cfg)
    kw = dict(num_starts=S) if S else {}
    dt = "multistart_sampling" if S else "sampling"
    torch.manual_seed(4242)
    a = pol(td.clone(), env, phase="test", decode_type=dt, return_sum_log_likelihood=False, **kw)
    torch.manual_seed(4242)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    M = td["locs"].shape[1]
    R = B * max(S, 1)
    from eam_rl4co_amd.policy import _max_decode_steps

    # (multistart: the pre-decoder hook takes the start node, so the kernel's step t is output column t + 1)
    noise = ops.exp1_noise(seed, R, _max_decode_steps(env_name, M, 1 if S else 0), M, DEV)
    b = pol(td.clone(), env, phase="test", decode_type=dt, return_sum_log_likelihood=False, noise=noise, **kw)
    assert_bits_equal(a["actions"], b["actions"], "actions")
    assert_bits_equal(a["log_likelihood"], b["log_likelihood"], "logp")
    torch.manual_seed(4243)
    c = pol(td.clone(), env, phase="test", decode_type=dt, **kw)
    assert not torch.equal(a["actions"], c["actions"])


def test_full_size_pomo_1024x100_mfma_kernel_equals_valu_kernel():
    """BASELINE.json configs[3] at its full per-GPU size (1024 instances x 100 starts, TSP-100, POMO policy, multistart
    sampling): the MFMA start-sharing kernel with in-place noise against the register-resident VALU kernel fed the 4.1 GB
    tensor of the same draws -- two independent kernels of the canonical arithmetic, bit-identical tours and log-probs --
    plus the size-independent properties (permutations, start nodes in (s b) order)."""
    import eam_rl4co_amd as ea
    from eam_rl4co_amd import _lib, ops

    N, B, S = 100, 1024, 100
    pol = make_policy("pomo_tsp")
    env = ea.get_env("tsp", generator_params=dict(num_loc=N), seed=1234)
    td = env.reset(batch_size=[B]).to(DEV)
    torch.manual_seed(99)
    a = pol(td.clone(), env, phase="train", decode_type="multistart_sampling", num_starts=S, return_sum_log_likelihood=False)
    torch.manual_seed(99)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    noise = ops.exp1_noise(seed, B * S, N - 1, N, DEV)
    _lib.load().eamrl_debug_set(11, 1)
    try:
        b = pol(td.clone(), env, phase="train", decode_type="multistart_sampling", num_starts=S, noise=noise,
                return_sum_log_likelihood=False)
    finally:
        _lib.load().eamrl_debug_set(11, 0)
    del noise
    assert torch.equal(a["actions"], b["actions"]), "tours differ between the MFMA and the VALU kernel"
    assert torch.equal(a["log_likelihood"].view(torch.int32), b["log_likelihood"].view(torch.int32)), "log-probs differ"
    assert torch.equal(a["reward"], b["reward"])
    acts = a["actions"]
    assert (acts.sort(1).values == torch.arange(N, device=DEV)).all()
    assert (acts[:, 0] == torch.arange(S, device=DEV).repeat_interleave(B)).all()


def test_full_size_pomo_cvrp_256x100_mfma_kernel_equals_valu_kernel():
    """POMO on CVRP-100 (256 instances x 100 starts, multistart sampling, episodes of different lengths): the MFMA
    start-sharing kernel with its in-kernel CVRP state machine and in-place noise against the register-resident VALU kernel
    fed the tensor of the same draws -- bit-identical tours, log-probs, rewards and final env state -- plus feasibility."""
    import eam_rl4co_amd as ea
    from eam_rl4co_amd import _lib, ops
    from eam_rl4co_amd.policy import _max_decode_steps

    N, B, S = 100, 256, 100
    pol = make_policy("am_cvrp")
    env = ea.get_env("cvrp", generator_params=dict(num_loc=N), seed=4321)
    td = env.reset(batch_size=[B]).to(DEV)
    torch.manual_seed(77)
    a = pol(td.clone(), env, phase="train", decode_type="multistart_sampling", num_starts=S, return_sum_log_likelihood=False)
    td_a = pol._last_td
    torch.manual_seed(77)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    noise = ops.exp1_noise(seed, B * S, _max_decode_steps("cvrp", N + 1, 1), N + 1, DEV)
    _lib.load().eamrl_debug_set(14, 1)
    try:
        b = pol(td.clone(), env, phase="train", decode_type="multistart_sampling", num_starts=S, noise=noise,
                return_sum_log_likelihood=False)
    finally:
        _lib.load().eamrl_debug_set(14, 0)
    td_b = pol._last_td
    del noise
    assert torch.equal(a["actions"], b["actions"]), "tours differ between the MFMA and the VALU kernel"
    assert torch.equal(a["log_likelihood"].view(torch.int32), b["log_likelihood"].view(torch.int32)), "log-probs differ"
    assert torch.equal(a["reward"], b["reward"])
    for k in ("action_mask", "visited", "used_capacity", "current_node", "done"):
        assert torch.equal(td_a[k], td_b[k]), k
    acts = a["actions"]
    assert (acts[:, 0] == (torch.arange(S, device=DEV).repeat_interleave(B) % N) + 1).all()
    env.check_solution_validity(ea.batchify(td, S), acts)
    lens = (acts != 0).sum(1)
    assert (lens == N).all() and acts.shape[1] > N          # every customer once, depot returns in between



@pytest.mark.parametrize("env_name,N,B,S", [("cvrptw", 100, 48, 100), ("pctsp", 100, 64, 100), ("op", 100, 64, 100),
                                            ("cvrptw", 20, 7, 20), ("pctsp", 50, 5, 50), ("op", 20, 6, 20),
                                            ("sdvrp", 100, 48, 100), ("sdvrp", 20, 7, 20), ("sdvrp", 50, 5, 50), ("sdvrp", 37, 3, 30)])
def test_sibling_env_multistart_mfma_kernel_equals_valu_kernel(env_name, N, B, S):
    """Round 3: the MFMA start-sharing kernel also rolls out CVRPTW, PCTSP, OP and SDVRP multistart batches (their state machines
    run per half-wavefront inside it; SDVRP adds the dynamic embedding's rank-one terms in the canonical order).  Against the register-resident VALU start-sharing kernel (debug key 14) on the same
    multistart-sampling batch and noise field: bit-identical tours, log-probs, rewards and final env state."""
    import time

    import eam_rl4co_amd as ea
    from eam_rl4co_amd import _lib, ops
    from eam_rl4co_amd.policy import _max_decode_steps

    pol = make_policy("am_" + env_name)
    env = ea.get_env(env_name, generator_params=dict(num_loc=N), seed=4000 + N)
    torch.manual_seed(N + B)
    td = env.reset(batch_size=[B]).to(DEV)
    # forced start nodes must be feasible first actions (OP / PCTSP resample infeasible ones at random: pin them instead)
    starts = (torch.arange(S, device=DEV).repeat_interleave(B) % N) + 1
    ok = td["action_mask"].repeat(S, 1).gather(1, starts[:, None]).squeeze(1)
    first_ok = td["action_mask"][:, 1:].float().argmax(1).repeat(S) + 1
    starts = torch.where(ok, starts, first_ok)
    kw = dict(decode_type="multistart_sampling", num_starts=S, return_sum_log_likelihood=False,
              select_start_nodes_fn=lambda td_, env_, n: starts)
    M = N + 1
    noise = ops.exp1_noise(1234 + N, B * S, _max_decode_steps(env_name, M, 1), M, DEV)
    outs, tds, ms = [], [], []
    for valu in (0, 1):
        _lib.load().eamrl_debug_set(14, valu)
        try:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs.append(pol(td.clone(), env, phase="test", noise=noise, **kw))
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
            tds.append(pol._last_td)
        finally:
            _lib.load().eamrl_debug_set(14, 0)
    a, b = outs
    print(f"{env_name}-{N} x {B} x {S}: MFMA start-sharing kernel {ms[0]:.1f} ms, VALU start-sharing kernel {ms[1]:.1f} ms (first calls)")
    assert torch.equal(a["actions"], b["actions"]), "tours differ between the MFMA and the VALU kernel"
    assert torch.equal(a["log_likelihood"].view(torch.int32), b["log_likelihood"].view(torch.int32)), "log-probs differ"
    assert torch.equal(a["reward"], b["reward"])
    for k in tds[0].keys():
        if isinstance(tds[0][k], torch.Tensor):
            assert torch.equal(tds[0][k], tds[1][k]), k
