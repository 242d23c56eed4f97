"""Training companion of the native rollout: gradients by teacher-forced re-evaluation.

The rollout itself (`AttentionModelPolicy.forward`) runs on the HIP kernels and produces no gradients.  For
REINFORCE / POMO / EAM the log-likelihood of the *chosen* actions is what must be differentiated; because the
actions are known after the rollout, every decode step's state (current / first node, visited set, used
capacity) is a prefix function of the action sequence, so all T steps are re-evaluated AT ONCE as dense batched
contractions with PyTorch autograd -- the same trick serves EAM's `policy(..., actions=improved)` pass
(rl4co/models/zoo/earl/model.py:179-195).  This module is the gradient path only: it is not used by inference,
benchmarks or parity tests of the rollout, and its forward values are checked against the native log-probs
(tests/test_gpu_train.py, tolerance 1e-4).

Reference call sites: rl4co/models/rl/reinforce/reinforce.py:59-106 (REINFORCE.shared_step / calculate_loss),
rl4co/models/rl/reinforce/baselines.py:57-61 (SharedBaseline), rl4co/models/zoo/pomo/model.py:89-148.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .utils import unbatchify


# ------------------------------------------------------------------------------------------------------------
# differentiable encoder + cache (same parameters as the native path)
# ------------------------------------------------------------------------------------------------------------
def _normalize(norm: nn.Module, x: torch.Tensor, training: bool) -> torch.Tensor:
    n = norm.normalizer
    if isinstance(n, nn.BatchNorm1d):   # batch statistics when training, as the reference (nn/ops.py:45-47)
        y = F.batch_norm(x.reshape(-1, x.size(-1)), n.running_mean, n.running_var, n.weight, n.bias, training,
                         n.momentum if n.momentum is not None else 0.1, n.eps)
        return y.view_as(x)
    return F.instance_norm(x.permute(0, 2, 1), weight=n.weight, bias=n.bias, eps=n.eps).permute(0, 2, 1)


def encode_autograd(policy, td):
    enc = policy.encoder
    ie = enc.init_embedding
    locs = td["locs"]
    if policy.env_name == "tsp":
        h = F.linear(locs, ie.init_embed.weight, ie.init_embed.bias)
    else:
        depot = F.linear(locs[:, :1], ie.init_embed_depot.weight, ie.init_embed_depot.bias)
        if policy.env_name == "pctsp":
            feat = torch.cat((locs[:, 1:], td["expected_prize"][..., None], td["penalty"][..., 1:, None]), -1)
        elif policy.env_name == "op":
            feat = torch.cat((locs[:, 1:], td["prize"][..., 1:, None]), -1)
        elif policy.env_name == "cvrptw":
            feat = torch.cat((locs[:, 1:], td["demand"][..., None], td["time_windows"][..., 1:, :].float(),
                              td["durations"][..., 1:, None]), -1)
        else:
            feat = torch.cat((locs[:, 1:], td["demand"][..., None]), -1)
        h = torch.cat((depot, F.linear(feat, ie.init_embed.weight, ie.init_embed.bias)), 1)
    training = policy.training
    for layer in enc.net.layers:
        mha, ffn = layer[0].module, layer[2].module
        B, N, E = h.shape
        H = mha.num_heads
        qkv = F.linear(h, mha.Wqkv.weight, mha.Wqkv.bias).view(B, N, 3, H, E // H).permute(2, 0, 3, 1, 4)
        att = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2]).permute(0, 2, 1, 3).reshape(B, N, E)
        h = _normalize(layer[1], h + F.linear(att, mha.out_proj.weight, mha.out_proj.bias), training)
        x = h
        for lin in ffn.lins[:-1]:
            x = F.relu(F.linear(x, lin.weight, lin.bias))
        h = _normalize(layer[3], h + F.linear(x, ffn.lins[-1].weight, ffn.lins[-1].bias), training)
    return h


# ------------------------------------------------------------------------------------------------------------
# per-step states from the action sequence
# ------------------------------------------------------------------------------------------------------------
def _tsp_states(actions, M, multistart):
    """-> first [R,T], cur [R,T], step0 [T] bool (placeholder context), mask [R,T,M] (True = feasible)."""
    R, T = actions.shape
    onehot = F.one_hot(actions, M).to(torch.int32)
    visited_before = torch.cumsum(onehot, 1) - onehot          # exclusive prefix
    cur = torch.cat((actions[:, :1], actions[:, :-1]), 1)       # a_{t-1}; column 0 unused at the placeholder step
    first = actions[:, :1].expand(R, T)
    return first, cur, visited_before == 0


def _cvrp_states(actions, demand_rows, vcap, M):
    """-> cur [R,T], remaining capacity [R,T] before each step, mask [R,T,M] (cvrp/env.py:68-100,132-144)."""
    R, T = actions.shape
    N = M - 1
    dev = actions.device
    visited = torch.zeros(R, M, dtype=torch.bool, device=dev)
    used = torch.zeros(R, dtype=torch.float32, device=dev)
    cur = torch.zeros(R, dtype=torch.int64, device=dev)
    lim = vcap + 1e-5
    curs, rems, masks = [], [], []
    ar = torch.arange(R, device=dev)
    for t in range(T):
        blocked = visited[:, 1:] | ((demand_rows + used[:, None]) > lim[:, None])
        depot_blocked = (cur == 0) & (~blocked).any(-1)
        masks.append(~torch.cat((depot_blocked[:, None], blocked), 1))
        curs.append(cur)
        rems.append(vcap - used)
        a = actions[:, t]
        d = demand_rows[ar, (a - 1).clamp(0, N - 1)]
        used = (used + d) * (a != 0).float()
        visited = visited.clone()
        visited[ar, a] = True
        cur = a
    return torch.stack(curs, 1), torch.stack(rems, 1), torch.stack(masks, 1)


def _pctsp_states(actions, prize_rows, prize_required):
    """-> cur [R,T], prize still to collect (clamped at 0) [R,T], mask [R,T,M] before each step
    (pctsp/env.py:64-97,156-163; context.py:194-208).  prize_rows [R, M] with a zero depot slot."""
    R, T = actions.shape
    M = prize_rows.shape[1]
    dev = actions.device
    visited = torch.zeros(R, M, dtype=torch.bool, device=dev)
    total = torch.zeros(R, dtype=torch.float32, device=dev)
    cur = torch.zeros(R, dtype=torch.int64, device=dev)
    curs, rems, masks = [], [], []
    ar = torch.arange(R, device=dev)
    for t in range(T):
        blocked = visited[:, 1:] | visited[:, :1]
        depot_blocked = (total < 1.0) & (~visited[:, 1:]).any(-1)
        masks.append(~torch.cat((depot_blocked[:, None], blocked), 1))
        curs.append(cur)
        rems.append((prize_required - total).clamp(min=0))
        a = actions[:, t]
        total = total + prize_rows[ar, a]
        visited = visited.clone()
        visited[ar, a] = True
        cur = a
    return torch.stack(curs, 1), torch.stack(rems, 1), torch.stack(masks, 1)


def _cvrptw_states(actions, demand_rows, vcap, locs_rows, tw_rows, dur_rows):
    """-> cur [R,T], free capacity [R,T], clock [R,T], mask [R,T,M] before each step (cvrptw/env.py:103-138).
    tw_rows [R,M,2] and dur_rows [R,M] as float32."""
    R, T = actions.shape
    M = locs_rows.shape[1]
    N = M - 1
    dev = actions.device
    visited = torch.zeros(R, M, dtype=torch.bool, device=dev)
    used = torch.zeros(R, dtype=torch.float32, device=dev)
    now = torch.zeros(R, dtype=torch.float32, device=dev)
    cur = torch.zeros(R, dtype=torch.int64, device=dev)
    lim = vcap + 1e-5
    curs, rems, nows, masks = [], [], [], []
    ar = torch.arange(R, device=dev)
    for t in range(T):
        blocked = visited[:, 1:] | ((demand_rows + used[:, None]) > lim[:, None])
        depot_blocked = (cur == 0) & (~blocked).any(-1)
        dist = (locs_rows[ar, cur][:, None, :] - locs_rows).norm(p=2, dim=-1)
        in_time = now[:, None] + dist <= tw_rows[..., 1]
        masks.append(~torch.cat((depot_blocked[:, None], blocked), 1) & in_time)
        curs.append(cur)
        rems.append(vcap - used)
        nows.append(now)
        a = actions[:, t]
        now = (a != 0).float() * (torch.maximum(now + dist[ar, a], tw_rows[ar, a, 0]) + dur_rows[ar, a])
        used = (used + demand_rows[ar, (a - 1).clamp(0, N - 1)]) * (a != 0).float()
        visited = visited.clone()
        visited[ar, a] = True
        cur = a
    return torch.stack(curs, 1), torch.stack(rems, 1), torch.stack(nows, 1), torch.stack(masks, 1)


def _op_states(actions, locs_rows, maxlen_rows):
    """-> cur [R,T], length still allowed to the depot's limit [R,T], mask [R,T,M] before each step
    (op/env.py:69-102,149-165; context.py:211-223).  locs_rows [R,M,2], maxlen_rows [R,M]."""
    R, T = actions.shape
    M = locs_rows.shape[1]
    dev = actions.device
    visited = torch.zeros(R, M, dtype=torch.bool, device=dev)
    length = torch.zeros(R, dtype=torch.float32, device=dev)
    cur = torch.zeros(R, dtype=torch.int64, device=dev)
    curs, rems, masks = [], [], []
    ar = torch.arange(R, device=dev)
    for t in range(T):
        here = locs_rows[ar, cur]
        exceeds = length[:, None] + (locs_rows - here[:, None, :]).norm(p=2, dim=-1) > maxlen_rows
        m = ~(visited | visited[:, :1] | exceeds)
        m[:, 0] = True
        masks.append(m)
        curs.append(cur)
        rems.append(maxlen_rows[:, 0] - length)
        a = actions[:, t]
        length = length + (locs_rows[ar, a] - here).norm(p=2, dim=-1)
        visited = visited.clone()
        visited[ar, a] = True
        cur = a
    return torch.stack(curs, 1), torch.stack(rems, 1), torch.stack(masks, 1)


def _sdvrp_states(actions, demand_rows, vcap, M):
    """-> cur [R,T], free capacity [R,T], mask [R,T,M] and remaining demand [R,T,M] before each step
    (sdvrp/env.py:58-92,137-146)."""
    R, T = actions.shape
    dev = actions.device
    rem = torch.cat((torch.zeros(R, 1, dtype=torch.float32, device=dev), demand_rows), 1)
    used = torch.zeros(R, dtype=torch.float32, device=dev)
    cur = torch.zeros(R, dtype=torch.int64, device=dev)
    curs, frees, masks, rems = [], [], [], []
    ar = torch.arange(R, device=dev)
    for t in range(T):
        blocked = (rem[:, 1:] == 0) | (used >= vcap)[:, None]
        depot_blocked = (cur == 0) & (~blocked).any(-1)
        masks.append(~torch.cat((depot_blocked[:, None], blocked), 1))
        curs.append(cur)
        frees.append(vcap - used)
        rems.append(rem)
        a = actions[:, t]
        delivered = torch.minimum(rem[ar, a], vcap - used)
        used = (used + delivered) * (a != 0).float()
        rem = rem.clone()
        rem[ar, a] = rem[ar, a] - delivered
        cur = a
    return torch.stack(curs, 1), torch.stack(frees, 1), torch.stack(masks, 1), torch.stack(rems, 1)


# ------------------------------------------------------------------------------------------------------------
def evaluate_log_likelihood(policy, td, env, actions, num_starts: int = 0, temperature=None, tanh_clipping=None,
                            chunk_rows: int = 4096):
    """Differentiable per-step log-probabilities of `actions` [R, T] (R = B or S*B rows in (s b) order; for
    multistart the first column is the start node and gets log-prob 0).  Returns logp [R, T]."""
    temperature = policy.temperature if temperature is None else temperature
    clip = policy.tanh_clipping if tanh_clipping is None else tanh_clipping
    dec = policy.decoder
    E, H = dec.embed_dim, dec.num_heads
    D = E // H
    emb = encode_autograd(policy, td)
    B, M, _ = emb.shape
    kvl = F.linear(emb, dec.project_node_embeddings.weight)
    K, V, L = kvl.chunk(3, dim=-1)
    gctx = F.linear(emb.mean(1), dec.project_fixed_context.weight) if dec.use_graph_context else None
    Wctx = dec.context_embedding.project_context.weight
    R, T = actions.shape
    S = max(int(num_starts), 1)
    assert R == S * B
    multistart = S > 1
    out = []
    # rows are processed in chunks of whole start-groups so that memory stays bounded ([rows, H, T, M] scores)
    starts_per_chunk = max(1, chunk_rows // B)
    for s0 in range(0, S, starts_per_chunk):
        s1 = min(S, s0 + starts_per_chunk)
        rows = slice(s0 * B, s1 * B)
        act = actions[rows]
        nrep = s1 - s0
        rep = lambda x: x.repeat(nrep, *([1] * (x.dim() - 1)))   # (s b) order: instance index = row % B
        embr, Kr, Vr, Lr = rep(emb), rep(K), rep(V), rep(L)
        Rc = act.shape[0]
        ar = torch.arange(Rc, device=act.device)[:, None]
        if policy.env_name == "tsp":
            first, cur, mask = _tsp_states(act, M, multistart)
            ctx_in = torch.cat((embr[ar, first], embr[ar, cur]), -1)                       # [Rc, T, 2E]
            if not multistart:   # step 0 uses the learned placeholder (context.py:118-131)
                ctx_in = torch.cat((dec.context_embedding.W_placeholder.expand(Rc, 1, 2 * E), ctx_in[:, 1:]), 1)
        elif policy.env_name == "cvrptw":
            cur, rem, now, mask = _cvrptw_states(act, rep(td["demand"]), rep(td["vehicle_capacity"].reshape(-1)),
                                                 rep(td["locs"]), rep(td["time_windows"].float()), rep(td["durations"].float()))
            ctx_in = torch.cat((embr[ar, cur], rem[..., None], now[..., None]), -1)          # [Rc, T, E+2]
        elif policy.env_name == "op":
            cur, rem, mask = _op_states(act, rep(td["locs"]), rep(td["max_length"]))
            ctx_in = torch.cat((embr[ar, cur], rem[..., None]), -1)                          # [Rc, T, E+1]
        elif policy.env_name == "pctsp":
            cur, rem, mask = _pctsp_states(act, rep(td["real_prize"]), rep(td["prize_required"].reshape(-1)))
            ctx_in = torch.cat((embr[ar, cur], rem[..., None]), -1)                          # [Rc, T, E+1]
        elif policy.env_name == "cvrp":
            cur, rem, mask = _cvrp_states(act, rep(td["demand"]), rep(td["vehicle_capacity"].reshape(-1)), M)
            ctx_in = torch.cat((embr[ar, cur], rem[..., None]), -1)                          # [Rc, T, E+1]
        else:
            cur, rem, mask, dem_t = _sdvrp_states(act, rep(td["demand"]), rep(td["vehicle_capacity"].reshape(-1)), M)
            ctx_in = torch.cat((embr[ar, cur], rem[..., None]), -1)
        q = F.linear(ctx_in, Wctx)
        if gctx is not None:
            q = q + rep(gctx)[:, None, :]
        qh = q.view(Rc, T, H, D).permute(0, 2, 1, 3)
        kh = Kr.view(Rc, M, H, D).permute(0, 2, 1, 3)
        vh = Vr.view(Rc, M, H, D).permute(0, 2, 1, 3)
        if policy.env_name == "sdvrp":
            # dynamic embedding (dynamic.py:59-78): K/V/L rows + remaining demand * projection columns.  The update is
            # rank one, so it enters as a score bias and two outer products instead of [Rc, T, M, E] tensors
            wk, wv, wl = dec.dynamic_embedding.projection.weight.view(3, E)
            qw = (qh * wk.view(1, H, 1, D)).sum(-1)                                              # [Rc, H, T]
            bias = (dem_t[:, None] * qw[..., None]) / math.sqrt(D)                               # [Rc, H, T, M]
            bias = bias.masked_fill(~mask[:, None], float("-inf"))
            att = torch.softmax(torch.matmul(qh, kh.transpose(-1, -2)) / math.sqrt(D) + bias, dim=-1)
            heads = torch.matmul(att, vh) + (att * dem_t[:, None]).sum(-1, keepdim=True) * wv.view(1, H, 1, D)
        else:
            heads = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=mask[:, None])     # [Rc, H, T, D]
        glimpse = F.linear(heads.permute(0, 2, 1, 3).reshape(Rc, T, E), dec.pointer.project_out.weight)
        logits = torch.bmm(glimpse, Lr.transpose(1, 2))
        if policy.env_name == "sdvrp":
            logits = logits + dem_t * (glimpse @ wl)[..., None]
        logits = logits / math.sqrt(E)
        if clip > 0:
            logits = torch.tanh(logits) * clip
        logits = logits.masked_fill(~mask, float("-inf")) / temperature
        logp = F.log_softmax(logits, dim=-1).gather(-1, act[..., None]).squeeze(-1)
        if multistart:   # the start node is not a decision (decoding.py:318-324)
            logp = torch.cat((torch.zeros_like(logp[:, :1]), logp[:, 1:]), 1)
            # steps t >= 1 were evaluated with the state after the start action: column t uses prefix a_{<t} as built
        out.append(logp)
    return torch.cat(out, 0)


def reinforce_loss(policy, env, td, baseline: str = "shared", num_starts: int = 0, decode_type: str = None,
                   **rollout_kwargs):
    """One REINFORCE forward: native sampled rollout (no grad) -> differentiable log-likelihood -> loss.

    baseline: "shared" (POMO: mean over the starts of an instance, needs num_starts > 1), "mean" (batch mean) or
    "no".  Returns dict(loss, reward, log_likelihood, actions).  loss = -((reward - bl) * ll).mean()
    (reinforce.py:103-106)."""
    if decode_type is None:
        decode_type = "multistart_sampling" if num_starts > 1 else "sampling"
    kw = dict(rollout_kwargs)
    if num_starts > 1:
        kw["num_starts"] = num_starts
    was_training = policy.training
    policy.eval()            # the native rollout uses running statistics
    with torch.no_grad():
        out = policy(td, env, phase="train", decode_type=decode_type, **kw)
    policy.train(was_training)
    actions, reward = out["actions"], out["reward"]
    # finished CVRP rows are padded with depot visits of probability 1: their log-prob is 0 and carries no gradient
    logp = evaluate_log_likelihood(policy, td, env, actions, num_starts=num_starts)
    ll = logp.sum(1)
    if baseline == "shared":
        assert num_starts > 1, "shared baseline needs multistart"
        r = unbatchify(reward, num_starts)
        adv = (r - r.mean(1, keepdim=True))
        loss = -(adv * unbatchify(ll, num_starts)).mean()
    elif baseline == "mean":
        loss = -((reward - reward.mean()) * ll).mean()
    else:
        loss = -(reward * ll).mean()
    return {"loss": loss, "reward": reward, "log_likelihood": ll, "actions": actions, "logp_steps": logp,
            "native_log_likelihood": out["log_likelihood"]}


def eam_loss(policy, env, td, ea, num_starts: int, improve: bool = True, draws=None, generator=None):
    """One EAM training step of the fork with the POMO (shared) baseline (rl4co/models/zoo/earl/model.py:129-247):

    1. sampled multistart rollout on the native path (no grad);
    2. the sampled tours are improved by the evolutionary operators on the GPU (`evolution_worker`; the reference
       ships them to CPU threads, model.py:166-171) and get their start column back (`_align_improved_actions`);
    3. both sets of tours are re-evaluated with autograd (`evaluate_log_likelihood`, the reference's
       `policy(..., actions=improved)`, model.py:189-195);
    4. REINFORCE with the per-instance mean over starts as baseline, over the concatenation [original; improved]
       treated as 2B instances (model.py:226-244, reinforce.py:103-106).

    Returns dict(loss, reward, improved_reward, log_likelihood, improved_log_likelihood, actions, improved_actions)."""
    from .evolution import evolution_worker
    from .utils import batchify

    S = int(num_starts)
    assert S > 1, "the EAM step uses the multistart (shared) baseline"
    was_training = policy.training
    policy.eval()
    with torch.no_grad():
        out = policy(td, env, phase="train", decode_type="multistart_sampling", num_starts=S)
    policy.train(was_training)
    actions, reward = out["actions"], out["reward"]
    ll = evaluate_log_likelihood(policy, td, env, actions, num_starts=S).sum(1)
    rs, lls = [unbatchify(reward, S)], [unbatchify(ll, S)]
    res = {"reward": reward, "log_likelihood": ll, "actions": actions}
    if improve:
        with torch.no_grad():
            improved, _ = evolution_worker(actions, td, ea, env, draws=draws, generator=generator)
            if improved.shape[-1] + 1 == actions.shape[-1]:          # _align_improved_actions
                improved = torch.cat([actions[:, :1], improved], dim=-1)
            r_imp = env.get_reward(batchify(td, S), improved)
        ll_imp = evaluate_log_likelihood(policy, td, env, improved, num_starts=S).sum(1)
        rs.append(unbatchify(r_imp, S))
        lls.append(unbatchify(ll_imp, S))
        res.update(improved_actions=improved, improved_reward=r_imp, improved_log_likelihood=ll_imp)
    r_all, ll_all = torch.cat(rs, 0), torch.cat(lls, 0)              # [B or 2B, S]
    adv = r_all - r_all.mean(1, keepdim=True)
    res["loss"] = -(adv * ll_all).mean()
    return res
