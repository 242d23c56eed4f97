"""Evaluation harness with the reference's classes and call pattern (rl4co/tasks/eval.py:18-410): each evaluator
feeds batches through env.reset -> policy(...) -> env.get_reward and keeps the best trajectory per instance.
Host-side orchestration only; every rollout is the native path.  Pinned by reference-generated fixtures
(tests/golden/eval_*.npz, `make_golden.py eval`)."""
from __future__ import annotations

import time

import numpy as np
import torch
from torch.utils.data import DataLoader

from .utils import StateAugmentation, batchify, gather_by_index, sample_n_random_actions, unbatchify


class EvalBase:
    """eval.py:18-85"""
    name = "base"

    def __init__(self, env, progress=False, **kwargs):
        self.env = env
        self.progress = progress

    def __call__(self, policy, dataloader, **kwargs):
        start = time.time()
        rewards_list, actions_list = [], []
        device = next(policy.parameters()).device
        with torch.inference_mode():
            for batch in dataloader:
                td = self.env.reset(batch.to(device))
                actions, rewards = self._inner(policy, td, **kwargs)
                rewards_list.append(rewards)
                actions_list.append(actions)
            rewards = torch.cat(rewards_list)
            max_len = max(a.size(-1) for a in actions_list)      # pad actions to the same length with zeros
            actions = torch.cat([torch.nn.functional.pad(a, (0, max_len - a.size(-1))) for a in actions_list], 0)
        inference_time = time.time() - start
        return {"actions": actions.cpu(), "rewards": rewards.cpu(), "inference_time": inference_time,
                "avg_reward": rewards.cpu().mean()}

    def _inner(self, policy, td):
        raise NotImplementedError("Implement in subclass")


class GreedyEval(EvalBase):
    """Greedy decoding, single trajectory (eval.py:88-104)."""
    name = "greedy"

    def _inner(self, policy, td):
        out = policy(td.clone(), self.env, decode_type="greedy", num_starts=0)
        return out["actions"], self.env.get_reward(td, out["actions"])


class AugmentationEval(EvalBase):
    """Best of N state augmentations (eval.py:107-150): random rotations / reflections ('symmetric', the default) or,
    with force_dihedral_8, POMO's 8 symmetries.  `phi`: the rotation angles to use for 'symmetric' (an input, like the
    sampling noise; fresh torch.rand draws when None)."""
    name = "augmentation"

    def __init__(self, env, num_augment=8, force_dihedral_8=False, feats=None, phi=None, **kwargs):
        super().__init__(env, kwargs.get("progress", False))
        self.augmentation = StateAugmentation(num_augment=num_augment,
                                              augment_fn="dihedral8" if force_dihedral_8 else "symmetric", feats=feats, phi=phi)

    @property
    def num_augment(self):
        return self.augmentation.num_augment

    def _inner(self, policy, td, num_augment=None):
        n = self.augmentation.num_augment if num_augment is None else num_augment
        td_init = td.clone()
        td = self.augmentation(td)
        out = policy(td.clone(), self.env, decode_type="greedy", num_starts=0)
        rewards = unbatchify(self.env.get_reward(batchify(td_init, n), out["actions"]), n)
        actions = unbatchify(out["actions"], n)
        rewards, idx = rewards.max(dim=1)
        return gather_by_index(actions, idx, dim=1), rewards


class SamplingEval(EvalBase):
    """Best of `samples` sampled trajectories (eval.py:153-204).  As in the reference the call passes
    num_starts=samples together with multisample=True, which the decoding strategy resolves to MULTISTART rollouts whose
    first action is a uniformly random feasible node with log-prob 0 (decoding.py:246-256, ops.py:239-256)."""
    name = "sampling"

    def __init__(self, env, samples, softmax_temp=None, select_best=True, temperature=1.0, top_p=0.0, top_k=0, **kwargs):
        super().__init__(env, kwargs.get("progress", False))
        self.samples, self.softmax_temp, self.temperature = samples, softmax_temp, temperature
        self.select_best, self.top_p, self.top_k = select_best, top_p, top_k

    def _inner(self, policy, td, **policy_kwargs):
        out = policy(td.clone(), self.env, decode_type="sampling", num_starts=self.samples, temperature=self.temperature,
                     top_p=self.top_p, top_k=self.top_k, multisample=True, softmax_temp=self.softmax_temp,
                     select_best=self.select_best,
                     select_start_nodes_fn=lambda td, _, n: sample_n_random_actions(td, n), **policy_kwargs)
        return out["actions"], out["reward"]


class GreedyMultiStartEval(EvalBase):
    """Best of `num_starts` greedy multistart trajectories (eval.py:207-241)."""
    name = "multistart_greedy"

    def __init__(self, env, num_starts=None, **kwargs):
        super().__init__(env, kwargs.get("progress", False))
        assert num_starts is not None, "Must specify num_starts"
        self.num_starts = num_starts

    def _inner(self, policy, td):
        td_init = td.clone()
        out = policy(td.clone(), self.env, decode_type="multistart_greedy", num_starts=self.num_starts)
        rewards = unbatchify(self.env.get_reward(batchify(td_init, self.num_starts), out["actions"]), self.num_starts)
        actions = unbatchify(out["actions"], self.num_starts)
        rewards, idx = rewards.max(dim=1)
        return gather_by_index(actions, idx, dim=1), rewards


class GreedyMultiStartAugmentEval(EvalBase):
    """Best over N augmentations x greedy multistarts (eval.py:244-297)."""
    name = "multistart_greedy_augment"

    def __init__(self, env, num_starts=None, num_augment=8, force_dihedral_8=False, feats=None, phi=None, **kwargs):
        super().__init__(env, kwargs.get("progress", False))
        assert num_starts is not None, "Must specify num_starts"
        self.num_starts = num_starts
        assert not (num_augment != 8 and force_dihedral_8), "Cannot force dihedral 8 when num_augment != 8"
        self.augmentation = StateAugmentation(num_augment=num_augment,
                                              augment_fn="dihedral8" if force_dihedral_8 else "symmetric", feats=feats, phi=phi)

    @property
    def num_augment(self):
        return self.augmentation.num_augment

    def _inner(self, policy, td, num_augment=None):
        n = self.augmentation.num_augment if num_augment is None else num_augment
        td_init = td.clone()
        td = self.augmentation(td)
        out = policy(td.clone(), self.env, decode_type="multistart_greedy", num_starts=self.num_starts)
        total = self.num_starts * n
        rewards = unbatchify(self.env.get_reward(batchify(td_init, (n, self.num_starts)), out["actions"]), total)
        actions = unbatchify(out["actions"], total)
        rewards, idx = rewards.max(dim=1)
        return gather_by_index(actions, idx, dim=1), rewards


def get_automatic_batch_size(eval_fn, start_batch_size=8192, max_batch_size=4096):
    """eval.py:300-332: the batch shrinks with the rows an instance expands to (starts / 10, augmentations, samples)."""
    batch_size = start_batch_size
    if hasattr(eval_fn, "num_starts"):
        batch_size = batch_size // max(eval_fn.num_starts // 10, 1)
    if hasattr(eval_fn, "num_augment"):
        batch_size = batch_size // eval_fn.num_augment
    if hasattr(eval_fn, "samples"):
        batch_size = batch_size // eval_fn.samples
    batch_size = max(min(batch_size, max_batch_size), 1)
    return 2 ** int(np.log2(batch_size))


def evaluate_policy(env, policy, dataset, method="greedy", batch_size=None, max_batch_size=4096, start_batch_size=8192,
                    auto_batch_size=True, samples=1280, softmax_temp=1.0, num_augment=8, force_dihedral_8=True, **kwargs):
    """eval.py:335-410"""
    num_loc = getattr(env.generator, "num_loc", None)
    methods = {
        "greedy": (GreedyEval, {}),
        "sampling": (SamplingEval, {"samples": samples, "softmax_temp": softmax_temp}),
        "multistart_greedy": (GreedyMultiStartEval, {"num_starts": num_loc}),
        "augment_dihedral_8": (AugmentationEval, {"num_augment": num_augment, "force_dihedral_8": force_dihedral_8}),
        "augment": (AugmentationEval, {"num_augment": num_augment}),
        "multistart_greedy_augment_dihedral_8": (GreedyMultiStartAugmentEval,
                                                 {"num_augment": num_augment, "force_dihedral_8": force_dihedral_8,
                                                  "num_starts": num_loc}),
        "multistart_greedy_augment": (GreedyMultiStartAugmentEval, {"num_augment": num_augment, "num_starts": num_loc}),
    }
    assert method in methods, f"Method {method} not found"
    cls, kw = methods[method]
    kw.update(kwargs)
    eval_fn = cls(env, **kw)
    if auto_batch_size:
        assert batch_size is None, "Cannot specify batch_size when auto_batch_size is True"
        batch_size = get_automatic_batch_size(eval_fn, max_batch_size=max_batch_size, start_batch_size=start_batch_size)
    loader = DataLoader(dataset, batch_size=batch_size, shuffle=False, num_workers=0, collate_fn=dataset.collate_fn)
    return eval_fn(policy, loader)
