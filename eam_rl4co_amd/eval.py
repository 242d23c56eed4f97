"""Evaluation harness with the reference's classes and call pattern (rl4co/tasks/eval.py:18-410): each evaluator
feeds batches through env.reset -> policy(...) -> env.get_reward and keeps the best trajectory per instance.
Host-side orchestration only; every rollout is the native path."""
from __future__ import annotations

import time

import torch
from torch.utils.data import DataLoader

from .utils import StateAugmentation, batchify, gather_by_index, unbatchify


class EvalBase:
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
            max_len = max(a.size(-1) for a in actions_list)
            actions = torch.cat([torch.nn.functional.pad(a, (0, max_len - a.size(-1))) for a in actions_list], 0)
        inference_time = time.time() - start
        return {"actions": actions.cpu(), "rewards": rewards.cpu(), "inference_time": inference_time,
                "avg_reward": rewards.cpu().mean()}

    def _inner(self, policy, td):
        raise NotImplementedError


class GreedyEval(EvalBase):
    """Greedy decoding, single trajectory (eval.py:88-104)."""
    name = "greedy"

    def _inner(self, policy, td):
        out = policy(td.clone(), self.env, decode_type="greedy", num_starts=0)
        return out["actions"], self.env.get_reward(td, out["actions"])


class AugmentationEval(EvalBase):
    """Best of the 8 dihedral augmentations (eval.py:107-150)."""
    name = "augmentation"

    def __init__(self, env, num_augment=8, force_dihedral_8=True, feats=None, **kwargs):
        super().__init__(env, kwargs.get("progress", False))
        self.augmentation = StateAugmentation(num_augment=num_augment, augment_fn="dihedral8", feats=feats)

    @property
    def num_augment(self):
        return self.augmentation.num_augment

    def _inner(self, policy, td, num_augment=None):
        n = self.augmentation.num_augment if num_augment is None else num_augment
        td_init = td.clone()
        out = policy(self.augmentation(td), self.env, decode_type="greedy", num_starts=0)
        rewards = unbatchify(self.env.get_reward(batchify(td_init, n), out["actions"]), n)
        actions = unbatchify(out["actions"], n)
        rewards, idx = rewards.max(dim=1)
        return gather_by_index(actions, idx, dim=1), rewards


class SamplingEval(EvalBase):
    """Best of `samples` sampled trajectories (eval.py:153-204)."""
    name = "sampling"

    def __init__(self, env, samples, softmax_temp=None, select_best=True, temperature=1.0, top_p=0.0, top_k=0, **kwargs):
        super().__init__(env, kwargs.get("progress", False))
        self.samples, self.temperature, self.select_best = samples, temperature, select_best
        self.top_p, self.top_k = top_p, top_k

    def _inner(self, policy, td):
        out = policy(td.clone(), self.env, decode_type="sampling", num_samples=self.samples, multisample=True,
                     temperature=self.temperature, top_p=self.top_p, top_k=self.top_k, select_best=self.select_best)
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
    """Best over dihedral-8 augmentations x greedy multistarts (eval.py:244-297)."""
    name = "multistart_greedy_augment"

    def __init__(self, env, num_starts=None, num_augment=8, force_dihedral_8=True, feats=None, **kwargs):
        super().__init__(env, kwargs.get("progress", False))
        assert num_starts is not None, "Must specify num_starts"
        self.num_starts = num_starts
        self.augmentation = StateAugmentation(num_augment=num_augment, augment_fn="dihedral8", feats=feats)

    @property
    def num_augment(self):
        return self.augmentation.num_augment

    def _inner(self, policy, td, num_augment=None):
        n = self.augmentation.num_augment if num_augment is None else num_augment
        td_init = td.clone()
        out = policy(self.augmentation(td), self.env, decode_type="multistart_greedy", num_starts=self.num_starts)
        total = self.num_starts * n
        rewards = unbatchify(self.env.get_reward(batchify(td_init, (n, self.num_starts)), out["actions"]), total)
        actions = unbatchify(out["actions"], total)
        rewards, idx = rewards.max(dim=1)
        return gather_by_index(actions, idx, dim=1), rewards


def evaluate_policy(env, policy, dataset, method="greedy", batch_size=None, max_batch_size=4096, samples=1280,
                    num_augment=8, force_dihedral_8=True, **kwargs):
    """eval.py:335-410 (without the automatic batch-size search: 288 GB of HBM hold any of these batches)."""
    num_loc = getattr(env.generator, "num_loc", None)
    methods = {
        "greedy": (GreedyEval, {}),
        "sampling": (SamplingEval, {"samples": samples}),
        "multistart_greedy": (GreedyMultiStartEval, {"num_starts": num_loc}),
        "augment_dihedral_8": (AugmentationEval, {"num_augment": 8, "force_dihedral_8": True}),
        "augment": (AugmentationEval, {"num_augment": num_augment, "force_dihedral_8": force_dihedral_8}),
        "multistart_greedy_augment_dihedral_8": (GreedyMultiStartAugmentEval,
                                                 {"num_augment": 8, "force_dihedral_8": True, "num_starts": num_loc}),
        "multistart_greedy_augment": (GreedyMultiStartAugmentEval,
                                      {"num_augment": num_augment, "force_dihedral_8": force_dihedral_8,
                                       "num_starts": num_loc}),
    }
    assert method in methods, f"Method {method} not found"
    cls, kw = methods[method]
    kw.update(kwargs)
    bs = batch_size or min(max_batch_size, len(dataset))
    loader = DataLoader(dataset, batch_size=bs, collate_fn=dataset.collate_fn)
    return cls(env, **kw)(policy, loader)
