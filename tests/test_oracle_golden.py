"""CPU: the C oracle against the golden vectors captured from the reference (tests/golden/make_golden.py).

Integer / bool results (actions, masks, visited, done, step counts) must be IDENTICAL.
Float results are compared with the tolerances below; the reference's torch CPU kernels have no
defined summation order (SURVEY.md section 7 hard part 1), so float equality with them is
tolerance-level by nature:
    per-step logits / log-probs   abs 1e-5   (values are O(1..10))
    rewards (tour lengths)        rel 1e-6
    summed log-likelihood         rel 2e-6   (sum of up to ~115 terms of O(1..5))
"""
import numpy as np
import pytest

from _util import cfg_for, golden, golden_weights, instance_of

POLICY_CASES = [
    "tsp20_greedy", "tsp20_sampling", "tsp20_evaluate", "tsp20_multistart_greedy", "tsp100_greedy", "tsp100_sampling",
    "cvrp20_greedy", "cvrp20_sampling", "cvrp20_evaluate", "cvrp20_multistart_greedy", "cvrp100_greedy",
    "cvrp100_sampling", "pomo_tsp20_multistart_sampling",
    # second batch: mid sizes, graphs above 128 nodes, POMO policy on CVRP, non-default temperature / clipping
    "tsp50_greedy", "cvrp50_sampling", "tsp200_greedy", "cvrp200_greedy", "pomo_cvrp20_multistart_greedy",
    "cvrp20_sampling_temp", "tsp20_greedy_noclip",
    # third batch: top-k / top-p filtering in process_logits
    "tsp20_sampling_topk5", "tsp20_sampling_topp", "cvrp20_sampling_topk_topp", "tsp100_greedy_topk",
    # SDVRP (split delivery): dynamic embedding + partial-delivery state machine
    "sdvrp20_greedy", "sdvrp20_sampling", "sdvrp50_greedy", "sdvrp20_multistart_greedy",
    # PCTSP (prize collecting): prize-gated depot, penalty reward
    "pctsp20_greedy", "pctsp20_sampling", "pctsp50_greedy", "pctsp100_sampling", "pctsp20_multistart_greedy",
    "spctsp20_sampling", "spctsp50_greedy",          # stochastic prizes: the env collects a prize the policy does not see
    # OP (orienteering): distance-dependent mask, prize reward
    "op20_greedy", "op20_sampling", "op50_greedy", "op100_sampling", "op20_multistart_greedy",
    # CVRPTW (time windows): CVRP + clock, reachability mask, two state columns in the context
    "cvrptw20_greedy", "cvrptw20_sampling", "cvrptw50_greedy", "cvrptw100_sampling", "cvrptw20_multistart_greedy",
]


def _run(orc, fx):
    sd = golden_weights(cfg_for(fx))
    decode_type = str(fx["decode_type"])
    given = fx["actions"] if decode_type == "evaluate" else None
    ns = int(fx["num_starts"])
    if ns > 1 and "multistart" not in decode_type:
        decode_type = "multistart_" + decode_type
    return orc.policy_rollout(
        sd, str(fx["env_name"]), fx["locs"], instance_of(fx), decode_type=decode_type, num_starts=ns,
        noise=fx.get("noise"), given=given, use_graph_context=bool(fx.get("policy_kw_use_graph_context", True)),
        clip=float(fx.get("decode_kw_tanh_clipping", 10.0)), temp=float(fx.get("decode_kw_temperature", 1.0)),
        top_k=int(fx.get("decode_kw_top_k", 0)), top_p=float(fx.get("decode_kw_top_p", 0.0)),
        # OP may resample its start nodes at random (utils/ops.py:158-169): replay the recorded ones
        start_nodes=fx["actions"][:, 0] if (str(fx["env_name"]) == "op" and ns > 1) else None)


@pytest.mark.parametrize("name", POLICY_CASES)
def test_policy_rollout_matches_reference(oracle, name):
    fx = golden(name)
    if name.endswith("evaluate"):
        fx["decode_type"] = np.array("evaluate")
    out = _run(oracle, fx)
    assert out["actions"].shape == fx["actions"].shape
    assert np.array_equal(out["actions"], fx["actions"]), "tours differ from the reference"
    np.testing.assert_allclose(out["reward"], fx["reward"], rtol=1e-6, atol=0)
    # CVRPTW works on unscaled coordinates (0..150) and times (0..480): activations are ~100x larger than elsewhere and so
    # is the fp32 rounding noise between two summation orders; the tours are identical all the same
    tw = str(fx["env_name"]) == "cvrptw"
    np.testing.assert_allclose(out["logp_steps"], fx["logp_steps"], rtol=0, atol=2e-4 if tw else 1e-5)
    np.testing.assert_allclose(out["log_likelihood"], fx["log_likelihood"], rtol=2e-5 if tw else 2e-6, atol=0)


@pytest.mark.parametrize("name", ["tsp20_greedy", "cvrp20_greedy", "pomo_tsp20_multistart_sampling", "pctsp20_greedy",
                                  "op20_greedy", "cvrptw20_greedy"])
def test_encoder_and_cache_match_reference(oracle, name):
    fx = golden(name)
    sd = golden_weights(cfg_for(fx))
    env = str(fx["env_name"])
    init_h, emb = oracle.encode(sd, env, fx["locs"], instance_of(fx))
    # CVRPTW: unscaled inputs (coordinates to 150, times to 480) with these synthetic weights give activations of
    # magnitude 1e2..1e3 and near-one-hot encoder attention: an ill-conditioned network, in which two fp32 summation
    # orders differ by ~1e-4 of a tensor's scale.  The bound is relative to that scale there (tours are still identical).
    def close(got, want, atol):
        if env == "cvrptw":
            atol = 2e-4 * float(np.abs(want).max())
        np.testing.assert_allclose(got, want, rtol=0, atol=atol)

    close(init_h, fx["init_embeds"], 1e-6)
    close(emb, fx["embeddings"], 1e-5)
    use_gc = fx["graph_context"].size > 0
    cache = oracle.precompute(sd, env, emb, use_graph_context=use_gc)
    close(cache["K"], fx["glimpse_key"], 1e-5)
    close(cache["V"], fx["glimpse_val"], 1e-5)
    close(cache["L"], fx["logit_key"], 1e-5)
    if use_gc:
        close(cache["gctx"], fx["graph_context"], 1e-5)


@pytest.mark.parametrize("name", ["tsp20_greedy", "tsp100_greedy", "cvrp20_greedy", "cvrp100_greedy",
                                  "cvrp100_sampling", "tsp20_multistart_greedy"])
def test_per_step_logits_logprobs_masks(oracle, name):
    """Replay the reference's actions and compare raw decoder logits, processed log-probs and masks."""
    fx = golden(name)
    sd = golden_weights(cfg_for(fx))
    env = str(fx["env_name"])
    ns = int(fx["num_starts"])
    _, emb = oracle.encode(sd, env, fx["locs"], instance_of(fx))
    cache = oracle.precompute(sd, env, emb)
    st = oracle.State(env, fx["locs"], instance_of(fx), num_starts=ns)
    actions = fx["actions"]
    col = 0
    if ns > 1:
        st.step(actions[:, 0])
        col = 1
    kept = {int(s): i for i, s in enumerate(fx["steps_kept"])}
    for t in range(int(fx["n_decoder_steps"])):
        a, lp, logits, logprobs = oracle.decode_step(st, cache, "evaluate", given=actions[:, col + t], want_all=True)
        if t in kept:
            i = kept[t]
            ref_mask = fx["step_mask"][:, i].astype(bool)
            assert np.array_equal(st.mask.astype(bool), ref_mask), f"mask differs at step {t}"
            # the reference leaves the raw logits of masked nodes finite: compare feasible ones
            np.testing.assert_allclose(logits[ref_mask], fx["step_logits"][:, i][ref_mask], rtol=0, atol=1e-5)
            ref_lp = fx["step_logprobs"][:, i]
            assert np.array_equal(np.isneginf(logprobs), np.isneginf(ref_lp))
            np.testing.assert_allclose(logprobs[ref_mask], ref_lp[ref_mask], rtol=0, atol=1e-5)
        st.step(a)
    assert st.done.all()


@pytest.mark.parametrize("name", ["env_tsp20_random", "env_cvrp20_random", "env_cvrp100_random", "env_sdvrp20_random",
                                  "env_pctsp20_random", "env_spctsp20_random", "env_op20_random", "env_op50_random",
                                  "env_cvrptw20_random", "env_cvrptw50_random"])
def test_env_state_machine_bit_exact(oracle, name):
    fx = golden(name)
    env = str(fx["env_name"])
    if env == "tsp":
        locs, demand = fx["gen_locs"], None
    elif env == "cvrptw":
        locs = np.concatenate([fx["gen_depot"][:, None], fx["gen_locs"]], 1)
        demand = {"demand": fx["gen_demand"], "time_windows": fx["gen_time_windows"], "durations": fx["gen_durations"]}
    elif env == "op":
        import torch
        locs = np.concatenate([fx["gen_depot"][:, None], fx["gen_locs"]], 1)
        tl, td_ = torch.from_numpy(locs), torch.from_numpy(fx["gen_depot"])
        # the per-node arrival limit, with the reset's own torch expression (op/env.py:122-126)
        ml = torch.from_numpy(fx["gen_max_length"])[..., None] - (td_[..., None, :] - tl).norm(p=2, dim=-1) - 1e-6
        demand = {"prize": np.concatenate([np.zeros((locs.shape[0], 1), np.float32), fx["gen_prize"]], 1),
                  "max_length": ml.numpy()}
    elif env in ("pctsp", "spctsp"):
        locs = np.concatenate([fx["gen_depot"][:, None], fx["gen_locs"]], 1)
        pad = lambda a: np.concatenate([np.zeros((a.shape[0], 1), np.float32), a], 1)
        real = fx["gen_stochastic_prize"] if env == "spctsp" else fx["gen_deterministic_prize"]
        env = "pctsp"
        demand = {"expected_prize": fx["gen_deterministic_prize"], "real_prize": pad(real),
                  "penalty": pad(fx["gen_penalty"]), "prize_required": 1.0}
    else:
        locs = np.concatenate([fx["gen_depot"][:, None], fx["gen_locs"]], 1)
        demand = fx["gen_demand"]
    st = oracle.State(env, locs, demand)
    assert np.array_equal(st.mask.astype(bool), fx["reset_action_mask"])
    T = fx["step_action"].shape[1]
    for t in range(T):
        st.step(fx["step_action"][:, t])
        assert np.array_equal(st.mask.astype(bool), fx["step_action_mask"][:, t]), t
        assert np.array_equal(st.done.astype(bool), fx["step_done"][:, t]), t
        assert np.array_equal(st.cur, fx["step_current_node"][:, t].reshape(-1)), t
        if env == "tsp":
            assert np.array_equal(st.first, fx["step_first_node"][:, t]), t
            assert np.array_equal(st.istep, fx["step_i"][:, t].reshape(-1)), t
        elif env == "sdvrp":
            assert np.array_equal(st.rem, fx["step_demand_with_depot"][:, t]), t       # exact: min / add / sub only
            assert np.array_equal(st.used, fx["step_used_capacity"][:, t].reshape(-1)), t
        elif env == "cvrptw":
            assert np.array_equal(st.visited, fx["step_visited"][:, t]), t
            assert np.array_equal(st.used, fx["step_used_capacity"][:, t].reshape(-1)), t
            assert np.array_equal(st.time, fx["step_current_time"][:, t].reshape(-1)), t   # max / add of exact operands
        elif env == "op":
            assert np.array_equal(st.visited.astype(bool), fx["step_visited"][:, t]), t
            assert np.array_equal(st.used, fx["step_tour_length"][:, t]), t             # sqrtf(fmaf(dy,dy,dx*dx)) == torch
            assert np.array_equal(st.istep, fx["step_i"][:, t]), t
        elif env == "pctsp":
            assert np.array_equal(st.visited.astype(bool), fx["step_visited"][:, t]), t
            assert np.array_equal(st.used, fx["step_cur_total_prize"][:, t]), t         # one fp32 add per step
            assert np.array_equal(st.istep, fx["step_i"][:, t]), t
        else:
            assert np.array_equal(st.visited, fx["step_visited"][:, t]), t
            # one fp32 add + one mul per step, no reductions: exactly reproducible
            assert np.array_equal(st.used, fx["step_used_capacity"][:, t].reshape(-1)), t
    if env == "pctsp":
        reward = oracle.pctsp_reward(locs, demand["penalty"], fx["step_action"])
    elif env == "op":
        reward = oracle.op_reward(demand["prize"], fx["step_action"])
    else:
        reward = oracle.tour_length_reward(locs, fx["step_action"], with_depot=(env != "tsp"))
    np.testing.assert_allclose(reward, fx["reward"], rtol=1e-6, atol=0)
    if env == "tsp":
        assert oracle.check_tsp(fx["step_action"]) == 0
    elif env == "op":
        assert oracle.check_op(fx["step_action"], locs, demand["max_length"]) == 0
        twice = fx["step_action"].copy()
        twice[:, 1] = twice[:, 0]
        assert oracle.check_op(twice, locs, demand["max_length"]) % 1000000 == int((twice[:, 0] != 0).sum())
        far = np.tile(np.arange(1, locs.shape[1], dtype=np.int64), (locs.shape[0], 1))      # everyone: far too long
        assert oracle.check_op(far, locs, demand["max_length"]) // 1000000 == locs.shape[0]
    elif env == "pctsp":
        assert oracle.check_pctsp(fx["step_action"], demand["real_prize"]) == 0
        twice = fx["step_action"].copy()
        twice[0, 1] = twice[0, 0]                         # a customer visited twice
        assert oracle.check_pctsp(twice, demand["real_prize"]) % 1000000 == 1
        early = np.zeros_like(fx["step_action"]); early[:, 0] = fx["step_action"][:, 0]   # one customer, then home
        assert oracle.check_pctsp(early, demand["real_prize"]) // 1000000 == early.shape[0]
    elif env == "cvrp":
        assert oracle.check_cvrp(fx["step_action"], demand, 1.0) == 0
    elif env == "cvrptw":
        assert oracle.check_cvrp(fx["step_action"], demand["demand"], 1.0) == 0
        assert oracle.check_cvrptw_time(fx["step_action"], locs, demand["time_windows"], demand["durations"]) == 0
        # visiting customers in index order ignores the windows: some deadline is missed in every row
        naive = np.tile(np.arange(1, locs.shape[1], dtype=np.int64), (locs.shape[0], 1))
        assert oracle.check_cvrptw_time(naive, locs, demand["time_windows"], demand["durations"]) == locs.shape[0]
    else:
        assert (st.rem == 0).all()              # all demand delivered


def test_validity_checks_reject_bad_tours(oracle):
    fx = golden("env_cvrp20_random")
    acts = fx["step_action"].copy()
    assert oracle.check_cvrp(acts, fx["gen_demand"], 1.0) == 0
    bad = acts.copy()
    bad[0, np.nonzero(bad[0])[0][0]] = 0  # drop a customer
    assert oracle.check_cvrp(bad, fx["gen_demand"], 1.0) % 1000000 == 1
    over = np.tile(np.arange(1, 21, dtype=np.int64), (1, 1))  # one route serving everyone: over capacity
    assert oracle.check_cvrp(over, fx["gen_demand"][:1], 1.0) // 1000000 == 1
    t = golden("env_tsp20_random")["step_action"].copy()
    assert oracle.check_tsp(t) == 0
    t[1, 3] = t[1, 4]
    assert oracle.check_tsp(t) == 1


def test_defined_math_accuracy(oracle):
    """d_expf/d_logf/d_tanhf stay within a few ulp of libm on the ranges the rollout uses."""
    x = np.concatenate([np.linspace(-87, 0, 20001), np.linspace(0, 20, 4001)]).astype(np.float32)
    e, _, t = oracle.math_probe(x)
    ref_e = np.exp(x.astype(np.float64))
    sel = x <= 0
    assert np.max(np.abs(e[sel] - ref_e[sel]) / ref_e[sel]) < 4e-7
    assert np.max(np.abs(t - np.tanh(x.astype(np.float64)))) < 3e-7
    xl = np.linspace(1.0, 600.0, 30001).astype(np.float32)
    _, l, _ = oracle.math_probe(xl)
    assert np.max(np.abs(l - np.log(xl.astype(np.float64)))) < 6e-7
    assert oracle.math_probe(np.array([1.0], np.float32))[1][0] == 0.0
    assert oracle.math_probe(np.array([-np.inf, 0.0], np.float32))[0].tolist() == [0.0, 1.0]


# ---------------------------------------------------------------------------------------------------------
# evolutionary operators (the fork's EA): oracle/ea_oracle.py against outputs of the reference's own functions
# ---------------------------------------------------------------------------------------------------------
def test_ea_operators_match_reference():
    from oracle import ea_oracle as ea

    g = golden("ea_tsp_operators")
    off = ea.order_crossover_tsp(g["parents"], float(g["crossover_rate"]), g["cross_rand"], g["cross_idx"])
    np.testing.assert_array_equal(off, g["offspring"])
    mut = ea.inverse_mutate_tsp(g["offspring"], float(g["mutation_rate"]), g["mut_rand"], g["mut_idx"])
    np.testing.assert_array_equal(mut, g["mutated"])
    sel = ea.elitism_selection(g["parents"], g["fitness"], float(g["selection_rate"]))
    np.testing.assert_array_equal(sel, g["selected"])
    for row in off:      # offspring are permutations that keep their parent's first node
        assert sorted(row.tolist()) == list(range(off.shape[1]))
    np.testing.assert_array_equal(off[:, 0], g["parents"][: off.shape[0], 0])


@pytest.mark.parametrize("name", ["ea_tsp20_default", "ea_tsp20_busy", "ea_tsp50_busy"])
def test_ea_run_matches_reference(name):
    """EA.run of the reference (its operators executed with recorded draws) vs the restatement: identical
    populations; fitness within 1e-5 relative (the reference's cost is torch's tour length, ours the canonical one)."""
    from oracle import ea_oracle as ea

    g = golden(name)
    for b in range(g["locs"].shape[0]):
        pop, fit = ea.ea_run_tsp(g["locs"][b], g["init_pop"][b], int(g["num_generations"]), float(g["mutation_rate"]),
                                 float(g["crossover_rate"]), float(g["selection_rate"]), g["cross_rand"][:, b],
                                 g["cross_idx"][:, b], g["mut_rand"][:, b], g["mut_idx"][:, b])
        np.testing.assert_array_equal(pop, g["pop"][b])
        np.testing.assert_allclose(fit, g["fitness"][b], rtol=1e-5, atol=1e-5)
        assert (fit >= ea.tsp_fitness(ea.tsp_cost(g["locs"][b], g["init_pop"][b]), pop.shape[1]) - 1e-6).all()


def test_ea_population_from_single_tour():
    from oracle import ea_oracle as ea

    route = np.array([3, 0, 4, 1, 2], dtype=np.int64)
    pop = ea.generate_population_tsp(route, 7)
    assert pop[0].tolist() == route.tolist() and pop[1].tolist() == [0, 4, 1, 2, 3] and pop[5].tolist() == pop[1].tolist()
    assert pop[6].tolist() == [0, 4, 1, 2, 3]     # i % N == 1


def _cvrp_rint(g, b):
    from oracle import ea_oracle as ea

    u = {}
    S, P, O, G = g["init_mut_u"].shape[1], g["cross_u"].shape[2], g["mut_u"].shape[2], g["cross_u"].shape[0]
    for i in range(S):
        for k in range(3):
            u[(("init",), i, k)] = g["init_mut_u"][b, i, k]
    for gg in range(G):
        for p in range(P):
            u[(("cross", gg), p, 0)] = g["cross_u"][gg, b, p]
        for i in range(O):
            for k in range(3):
                u[(("mut", gg), i, k)] = g["mut_u"][gg, b, i, k]
    return ea.StructuredDraws(u)


@pytest.mark.parametrize("name", ["ea_cvrp20_default", "ea_cvrp20_busy", "ea_cvrp50_am"])
def test_ea_cvrp_run_matches_reference(name):
    """EA.run of the reference on CVRP (its operators with recorded draws, numba's float64 load accumulator emulated,
    see make_golden_ea.py) vs the restatement driven by per-slot uniforms: identical populations."""
    from oracle import ea_oracle as ea

    g = golden(name)
    for b in range(g["locs"].shape[0]):
        pop, fit = ea.ea_run_cvrp(g["locs"][b], g["demand"][b], float(g["vehicle_capacity"]), g["init_pop"][b],
                                  int(g["num_generations"]), float(g["mutation_rate"]), float(g["crossover_rate"]),
                                  float(g["selection_rate"]), g["init_mut_rand"][b], g["cross_rand"][:, b],
                                  g["mut_rand"][:, b], _cvrp_rint(g, b), top_k=bool(g["top_k"]))
        np.testing.assert_array_equal(pop, g["pop"][b])
        np.testing.assert_allclose(fit, g["fitness"][b], rtol=1e-5, atol=1e-5)
        N = g["demand"].shape[1]
        for row in pop:      # every customer exactly once, every route within the capacity
            assert sorted(x for x in row.tolist() if x) == list(range(1, N + 1))
            load = 0.0
            for x in row:
                load = 0.0 if x == 0 else load + float(g["demand"][b, x - 1])
                assert load <= float(g["vehicle_capacity"]) + 1e-5


def _prize_rint(g, b):
    from oracle import ea_oracle as ea

    u = {}
    S, P, O, G = g["init_mut_u"].shape[1], g["cross_u"].shape[2], g["mut_u"].shape[2], g["cross_u"].shape[0]
    for i in range(S):
        for k in range(2):
            u[(("init",), i, k)] = g["init_mut_u"][b, i, k]
    for gg in range(G):
        for p in range(P):
            u[(("cross", gg), p, 0)] = g["cross_u"][gg, b, p]
        for i in range(O):
            for k in range(2):
                u[(("mut", gg), i, k)] = g["mut_u"][gg, b, i, k]
    return ea.StructuredDraws(u)


@pytest.mark.parametrize("name", ["ea_pctsp20_default", "ea_pctsp20_busy", "ea_pctsp50_am"])
def test_ea_pctsp_run_matches_reference(name):
    """EA.run of the reference on PCTSP (cycle_crossover_pctsp + inverse_mutate_pctsp executed with recorded draws,
    numba's float64 accumulators emulated, see make_golden_ea.py) vs the restatement: identical populations."""
    from oracle import ea_oracle as ea

    g = golden(name)
    for b in range(g["locs"].shape[0]):
        pop, fit = ea.ea_run_pctsp(g["locs"][b], g["real_prize"][b], g["penalty"][b], g["init_pop"][b],
                                   int(g["num_generations"]), float(g["mutation_rate"]), float(g["crossover_rate"]),
                                   float(g["selection_rate"]), g["init_mut_rand"][b], g["cross_rand"][:, b],
                                   g["mut_rand"][:, b], _prize_rint(g, b), top_k=bool(g["top_k"]))
        np.testing.assert_array_equal(pop, g["pop"][b])
        np.testing.assert_allclose(fit, g["fitness"][b], rtol=1e-5, atol=1e-5)
        for row in pop:      # no customer twice, the minimum prize collected (or everything visited)
            nodes = [x for x in row.tolist() if x]
            assert len(nodes) == len(set(nodes))
            assert float(g["real_prize"][b][nodes].sum()) >= 1 - 1e-5 or len(nodes) == g["locs"].shape[1] - 1


@pytest.mark.parametrize("name", ["ea_op20_default", "ea_op20_busy", "ea_op50_busy"])
def test_ea_op_run_matches_reference(name):
    """EA.run of the reference on OP (order_crossover_op + inverse_mutate_op, recorded draws; tie-free runs with
    exactly summable prizes, see make_golden_ea.py) vs the restatement: identical populations, feasible routes."""
    from oracle import ea_oracle as ea

    g = golden(name)
    for b in range(g["locs"].shape[0]):
        pop, fit = ea.ea_run_op(g["locs"][b], g["prize"][b], g["max_length"][b], g["init_pop"][b],
                                int(g["num_generations"]), float(g["mutation_rate"]), float(g["crossover_rate"]),
                                float(g["selection_rate"]), g["init_mut_rand"][b], g["cross_rand"][:, b],
                                g["mut_rand"][:, b], _prize_rint(g, b), top_k=bool(g["top_k"]))
        np.testing.assert_array_equal(pop, g["pop"][b])
        np.testing.assert_array_equal(fit, g["fitness"][b])          # sums of multiples of 2^-20: exact in any order
        dist = ea.op_dist_matrix(g["locs"][b]).astype(np.float64)
        for row in pop:
            nodes = [x for x in row.tolist() if x]
            assert len(nodes) == len(set(nodes))
            path = [0] + row.tolist() + [0]
            assert sum(dist[a, c] for a, c in zip(path[:-1], path[1:])) <= float(g["max_length"][b][0]) + 1e-5


def test_ea_op_operators_match_reference():
    """Single calls of the reference's OP operators: ordinary parents come back unchanged from the crossover, the
    degenerate ones [customer, 0, ...] are rebuilt; rows with interior depot visits go through the mutation."""
    from oracle import ea_oracle as ea

    g = golden("ea_op_operators")
    dist = ea.op_dist_matrix(g["locs"])
    n = g["parents"].shape[0]
    u = {(("cross", 0), p, 0): g["cross_u"][p] for p in range(n // 2)}
    off = ea.order_crossover_op(g["parents"], float(g["crossover_rate"]), g["prize"], dist, g["max_length"], g["cross_rand"],
                                ea.StructuredDraws(u), ("cross", 0))
    np.testing.assert_array_equal(off, g["offspring"])
    assert (off != g["parents"]).any(-1).sum() >= 4
    u = {(("mut", 0), i, k): g["mut_u"][i, k] for i in range(n) for k in range(2)}
    mut = ea.inverse_mutate_op(g["offspring"], float(g["mutation_rate"]), g["prize"], dist, g["max_length"], g["mut_rand"],
                               ea.StructuredDraws(u), ("mut", 0))
    np.testing.assert_array_equal(mut, g["mutated"])


@pytest.mark.parametrize("name", ["tsp20_beam", "tsp20_beam5_all", "cvrp20_beam", "tsp50_beam12_all", "sdvrp20_beam", "pctsp20_beam"])
def test_beam_search_matches_reference(oracle, name):
    """decode_type="beam_search" of the reference (beam_width = num_loc or given, with and without select_best)."""
    fx = golden(name)
    bw = int(fx["decode_kw_beam_width"]) if "decode_kw_beam_width" in fx else None
    out = oracle.policy_beam_search(golden_weights(cfg_for(fx)), str(fx["env_name"]), fx["locs"], instance_of(fx),
                                    beam_width=bw, select_best=bool(fx["decode_kw_select_best"]))
    assert np.array_equal(out["actions"], fx["actions"]), "beam-search tours differ from the reference"
    np.testing.assert_allclose(out["reward"], fx["reward"], rtol=1e-6, atol=0)
    np.testing.assert_allclose(out["logp_steps"], fx["logp_steps"], rtol=0, atol=1e-5)


# ---------------------------------------------------------------------------------------------------------
# training mode: BatchNorm with batch statistics (policy.train(), nn/ops.py:45-47) -- `make_golden.py train`
# ---------------------------------------------------------------------------------------------------------
TRAIN_CASES = ["train_pomo_tsp20", "train_am_tsp20_bn", "train_am_cvrp20_bn", "train_am_tsp20_bn_multistart"]


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_train_mode_forward_matches_reference(oracle, name):
    """The reference policy in train() mode (sampling with recorded noise): same tours, log-probs within 1e-5,
    embeddings within 1e-5, running statistics after the forward within 1e-6."""
    fx = golden(name)
    sd = golden_weights(cfg_for(fx))
    ns = int(fx["num_starts"])
    out = oracle.policy_rollout(sd, str(fx["env_name"]), fx["locs"], instance_of(fx), decode_type=str(fx["decode_type"]),
                                num_starts=ns, noise=fx["noise"],
                                use_graph_context=bool(fx.get("policy_kw_use_graph_context", True)), training=True)
    assert np.array_equal(out["actions"], fx["actions"]), "tours differ from the reference"
    np.testing.assert_allclose(out["reward"], fx["reward"], rtol=1e-6, atol=0)
    np.testing.assert_allclose(out["logp_steps"], fx["logp_steps"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out["embeddings"], fx["embeddings"], rtol=0, atol=1e-5)
    nbuf = 0
    for k in fx:
        if k.startswith("buf__"):
            np.testing.assert_allclose(sd[k[5:]], fx[k], rtol=0, atol=1e-6, err_msg=k)
            nbuf += 1
    assert nbuf == (0 if name.startswith("train_pomo") else 12)     # 3 layers x 2 norms x (mean, var)


def test_batchnorm_train_is_the_defined_order(oracle):
    """orc_batchnorm_train against a float64 evaluation and torch's batch_norm (values; its own order is the contract)."""
    import torch

    rng = np.random.default_rng(5)
    x = rng.standard_normal((1000, 24)).astype(np.float32) * 3 + 1
    g, b = rng.standard_normal(24).astype(np.float32), rng.standard_normal(24).astype(np.float32)
    rm, rv = np.zeros(24, np.float32), np.ones(24, np.float32)
    y, m, v = oracle.batchnorm_train(x, g, b, rm, rv)
    trm, trv = torch.zeros(24), torch.ones(24)
    ty = torch.nn.functional.batch_norm(torch.from_numpy(x), trm, trv, torch.from_numpy(g), torch.from_numpy(b), True, 0.1, 1e-5)
    np.testing.assert_allclose(y, ty.numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(m, x.astype(np.float64).mean(0), rtol=0, atol=1e-6)
    np.testing.assert_allclose(v, x.astype(np.float64).var(0), rtol=2e-6, atol=0)
    np.testing.assert_allclose(rm, trm.numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(rv, trv.numpy(), rtol=2e-6, atol=0)


@pytest.mark.parametrize("tag", ["single", "multi", "wide"])
def test_pointer_attention_matches_reference_module(oracle, tag):
    """orc_pointer_attention against the reference's PointerAttention.forward on recorded inputs (`make_golden.py inject`)."""
    import goldweights

    fx = golden("pointer_attention")
    E = 128
    kvl = fx[f"{tag}_kvl"]
    k, v, lk = (np.ascontiguousarray(kvl[..., i * E:(i + 1) * E]) for i in range(3))
    w = goldweights.tensor_for("decoder.pointer.project_out.weight", (E, E))
    q = fx[f"{tag}_q"]
    out = oracle.pointer_attention(q, k, v, lk, fx[f"{tag}_mask"], w)
    want = fx[f"{tag}_logits"].reshape(out.shape)
    np.testing.assert_allclose(out, want, rtol=0, atol=2e-5)
