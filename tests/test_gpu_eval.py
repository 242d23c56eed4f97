"""GPU: the evaluation harness against fixtures recorded from the reference's own evaluators
(`tests/golden/make_golden.py eval`: rl4co/tasks/eval.py:88-297 run on one batch with the golden weights).
Bar: identical best actions, rewards within 1e-6 relative (the rewards are recomputed from the integer tours)."""
import glob
import os

import numpy as np
import pytest
import torch

from _util import GOLDEN, golden
from test_gpu_parity import DEV, make_policy

pytestmark = pytest.mark.gpu

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "eval_*.npz")))


def _evaluator(fx, env, monkeypatch):
    from eam_rl4co_amd import eval as ev

    method = str(fx["method"])
    kw = {k[len("eval_kw_"):]: (v.item() if v.ndim == 0 else v) for k, v in fx.items() if k.startswith("eval_kw_")}
    if "phi" in fx:          # the reference's random rotation angles are an input here
        kw["phi"] = torch.from_numpy(fx["phi"])
    call_kw = {}
    if method == "sampling":
        starts = torch.from_numpy(fx["start_nodes"]).permute(1, 0).reshape(-1).to(DEV)     # "(n b)" rows
        monkeypatch.setattr(ev, "sample_n_random_actions", lambda td, n: starts)
        call_kw["noise"] = torch.from_numpy(fx["noise"]).to(DEV)
    cls = {"greedy": ev.GreedyEval, "augment": ev.AugmentationEval, "sampling": ev.SamplingEval,
           "multistart_greedy": ev.GreedyMultiStartEval, "multistart_greedy_augment": ev.GreedyMultiStartAugmentEval}[method]
    return cls(env, **kw), call_kw


def test_fixture_set_is_complete():
    assert len(CASES) == 13, CASES


@pytest.mark.parametrize("name", CASES)
def test_evaluators_reproduce_the_reference(name, monkeypatch):
    import eam_rl4co_amd as ea

    fx = golden(name)
    env_name = str(fx["env_name"])
    pomo = "policy_kw_num_encoder_layers" in fx
    pol = make_policy(("pomo_" if pomo else "am_") + env_name)
    gen = {k[4:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("gen_")}
    B = gen["locs"].shape[0]
    env = ea.get_env(env_name, generator_params=dict(num_loc=gen["locs"].shape[1]))
    batch = ea.TensorDict(gen, batch_size=[B])
    eval_fn, call_kw = _evaluator(fx, env, monkeypatch)
    res = eval_fn(pol, [batch], **call_kw)
    assert np.array_equal(res["actions"].numpy(), fx["actions"]), "best actions differ from the reference"
    np.testing.assert_allclose(res["rewards"].numpy(), fx["rewards"], rtol=1e-6)


def test_evaluate_policy_defaults_follow_the_reference():
    """`augment` = symmetric rotations (not dihedral-8), auto batch size as the reference computes it."""
    import eam_rl4co_amd as ea
    from eam_rl4co_amd import eval as ev
    from eam_rl4co_amd.utils import symmetric_augmentation

    env = ea.get_env("tsp", generator_params=dict(num_loc=20))
    e = ev.AugmentationEval(env, num_augment=5)
    assert e.augmentation.augmentation is symmetric_augmentation and e.num_augment == 5
    with pytest.raises(AssertionError):
        ev.AugmentationEval(env, num_augment=5, force_dihedral_8=True)
    assert ev.get_automatic_batch_size(ev.GreedyMultiStartAugmentEval(env, num_starts=100, num_augment=8)) == 64
    assert ev.get_automatic_batch_size(ev.SamplingEval(env, samples=1280)) == 4
    assert ev.get_automatic_batch_size(ev.GreedyEval(env)) == 4096
