"""Full-batch audit against the REFERENCE at BASELINE.json's sizes (fixtures: tests/golden/make_audit.py).

The fixtures hold the reference's own tours / rewards on C2 (TSP-100 x 1024 greedy), C3 (CVRP-100 x 1024 sampling), C4's
per-instance shape (POMO TSP-100, 16 instances x 100 starts) and C5's graph size (CVRP-500 x 4), plus the table of the
reference's near-tie steps.  Protocol: a row may differ from the reference's tour only from a step on at which the
reference's best and second-best selection scores were closer than 1e-4 (torch's CPU summation orders are not
defined, so no independent implementation can promise more); every other row must be identical, and rewards of identical
tours agree to 1e-6 relative.  The CPU test holds the ORACLE to that, the GPU test the HIP path (which is also bit-equal
to the oracle, test_gpu_parity.py).  Match fractions at the time of writing: C2 1020 / 1024 (gaps 2.4e-7 .. 7.2e-7),
C3 1024 / 1024, C4 1600 / 1600, C5 4 / 4.
"""
import zlib

import numpy as np
import pytest
import torch

from _util import golden, golden_weights

CASES = ["audit_tsp100_b1024_greedy", "audit_cvrp100_b1024_sampling", "audit_pomo_tsp100_b16_s100", "audit_cvrp500_b4_greedy"]
TIE = 1e-4                                                # first-divergence gap bound
MIN_MATCH = {"audit_tsp100_b1024_greedy": 1016, "audit_cvrp100_b1024_sampling": 1020, "audit_pomo_tsp100_b16_s100": 1592,
             "audit_cvrp500_b4_greedy": 4}                # a regression in the arithmetic would show as a falling fraction


def instances(fx):
    """The audit's instances: the package's generator under the fixture's seed == the reference's (CRC-checked)."""
    import eam_rl4co_amd as ea

    if str(fx["torch_version"]) != torch.__version__:
        pytest.skip("audit fixtures were generated with another torch version (RNG stream may differ)")
    env_name = str(fx["env_name"])
    env = ea.get_env(env_name, generator_params=dict(num_loc=int(fx["num_loc"])), seed=int(fx["data_seed"]))
    torch.manual_seed(int(fx["data_seed"]))
    td = env.reset(batch_size=[int(fx["batch"])])
    assert zlib.crc32(td["locs"].numpy().tobytes()) == int(fx["locs_crc"]), "instances differ from the reference generator's"
    if env_name == "cvrp":
        assert zlib.crc32(td["demand"].numpy().tobytes()) == int(fx["demand_crc"])
    return env, td


def noise_shape(fx):
    """(R, T, M) of the counter-based Exp(1) field the reference run consumed (make_audit.py)."""
    S = int(fx["num_starts"])
    M = int(fx["num_loc"]) + (str(fx["env_name"]) != "tsp")
    R = int(fx["batch"]) * max(S, 1)
    T = (M - (1 if S else 0)) if str(fx["env_name"]) == "tsp" else 2 * M + 1
    return R, T, M


def audit(name, fx, actions, reward):
    """Asserts the protocol for `actions` / `reward` (numpy) against the reference fixture; returns the match count."""
    ref = fx["actions"].astype(np.int64)
    T = max(ref.shape[1], actions.shape[1])
    a, b = np.zeros((ref.shape[0], T), np.int64), np.zeros((ref.shape[0], T), np.int64)
    a[:, :ref.shape[1]], b[:, :actions.shape[1]] = ref, actions
    ne = a != b
    same = ~ne.any(1)
    first = ne.argmax(1)
    table = {(int(r), int(s)): float(g) for r, s, g in zip(fx["near_rows"], fx["near_steps"], fx["near_gaps"])}
    report = []
    for r in np.nonzero(~same)[0]:
        gap = table.get((int(r), int(first[r])))
        assert gap is not None and gap < TIE, (f"{name}: row {r} leaves the reference's tour at step {first[r]} where the "
                                               f"reference's top-2 gap is {gap} (not a near-tie)")
        report.append((int(r), int(first[r]), gap))
    n = int(same.sum())
    np.testing.assert_allclose(reward[same], fx["reward"][same], rtol=1e-6, err_msg=f"{name}: rewards of identical tours")
    print(f"{name}: {n} / {len(same)} tours identical to the reference's; near-tie divergences (row, step, gap): {report}")
    assert n >= MIN_MATCH[name], f"{name}: only {n} of {len(same)} tours match the reference"
    return n


@pytest.mark.parametrize("name", CASES)
def test_oracle_equals_reference_on_full_batches(oracle, name):
    fx = golden(name)
    env, td = instances(fx)
    env_name, S, pomo = str(fx["env_name"]), int(fx["num_starts"]), bool(fx["pomo"])
    noise = oracle.exp1_noise(int(fx["noise_seed"]), *noise_shape(fx)) if int(fx["noise_seed"]) >= 0 else None
    o = oracle.policy_rollout(golden_weights(("pomo_" if pomo else "am_") + env_name), env_name, td["locs"].numpy(),
                              td["demand"].numpy() if env_name == "cvrp" else None, decode_type=str(fx["decode_type"]),
                              num_starts=S, noise=noise, use_graph_context=not pomo)
    n = audit(name, fx, o["actions"], o["reward"])
    same = n == len(fx["reward"])
    if same:        # summed log-likelihoods: torch sums [R, T] in its own order; 1e-5 relative covers it
        np.testing.assert_allclose(o["log_likelihood"], fx["log_likelihood"], rtol=1e-5, atol=1e-4)
    # the generator's own record of where the oracle stood must still hold (same code, same machine class)
    assert n == len(fx["reward"]) - len(fx["oracle_div_rows"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_equals_reference_on_full_batches(name):
    """The product path at the audit's sizes: register-resident kernel (C2, C3), MFMA start-sharing kernel (C4), streaming
    kernel + tiled attention (C5) -- tours identical to the REFERENCE's on every row that has no near-tie."""
    from eam_rl4co_amd import ops
    from test_gpu_parity import make_policy

    fx = golden(name)
    env, td = instances(fx)
    env_name, S, pomo = str(fx["env_name"]), int(fx["num_starts"]), bool(fx["pomo"])
    pol = make_policy(("pomo_" if pomo else "am_") + env_name)
    kw = dict(decode_type=str(fx["decode_type"]))
    if S:
        kw["num_starts"] = S
    if int(fx["noise_seed"]) >= 0:
        kw["noise"] = ops.exp1_noise(int(fx["noise_seed"]), *noise_shape(fx))
    out = pol(td.to("cuda"), env, phase="test", **kw)
    audit(name, fx, out["actions"].cpu().numpy(), out["reward"].cpu().numpy())
