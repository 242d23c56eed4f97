"""CPU baseline for bench.py: the reference's rollout restated as the SAME SEQUENCE OF torch OPS on the host cores.

Test / measurement infrastructure (like everything under oracle/): only bench.py's `cpu_baseline` leg and
tools/validate_cpu_baseline.py import it; the product never does.

Why a second restatement next to eamrl_oracle.c: the C oracle defines the arithmetic (parity), but as a TIMING baseline it
understates the reference -- on 8 cores the reference's own PyTorch-CPU rollout of TSP-100 x 1024 takes 3.9 s, the C oracle
10.9 s.  BASELINE.md section 4 therefore asks for the reference's op sequence, not its arithmetic: this file
issues, step by step, the ATen calls the reference issues (the `cache + 0` copies of `_compute_kvl`, the head re-arrangements
around F.scaled_dot_product_attention, the per-step `.item()` / `.any()` host checks, the out-of-place mask scatter, the
TensorDict-style dict updates), with plain dicts instead of TensorDict.  tools/validate_cpu_baseline.py times it against
the shim-imported reference in the build container (profiles/r03_cpu_baseline_validation.json: within +-10 % at C1 / C2 / C3).
Values are the same up to torch's own run-to-run freedom; nothing here is used for parity.

Citations (reference files): models/common/constructive/base.py:157-275 (loop), zoo/am/encoder.py:70-91,
nn/graph/attnnet.py:16-103, nn/attention.py:66-136,224-328, nn/ops.py:32-56, nn/mlp.py:52-61, zoo/am/decoder.py:133-235,
nn/env_embeddings/context.py:50-74,105-157, utils/decoding.py:38-64,140-190,346-417, envs/routing/tsp/env.py:62-168,
envs/routing/cvrp/env.py:68-185, utils/ops.py:59-95.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

E, H = 128, 8


def _heads(x):          # "... g (h s) -> ... h g s": a strided view, as einops gives the reference
    B, G, _ = x.shape
    return x.view(B, G, H, E // H).permute(0, 2, 1, 3)


def encode(sd, env_name, td, eps=1e-5):
    """AttentionModelEncoder.forward: init embedding + GraphAttentionNetwork (BatchNorm in eval mode)."""
    pre = "encoder.init_embedding."
    if env_name == "tsp":
        h = F.linear(td["locs"], sd[pre + "init_embed.weight"], sd[pre + "init_embed.bias"])
    else:
        depot, cities = td["locs"][:, :1, :], td["locs"][:, 1:, :]
        dep = F.linear(depot, sd[pre + "init_embed_depot.weight"], sd[pre + "init_embed_depot.bias"])
        feat = torch.cat((cities, td["demand"][..., None]), -1)
        h = torch.cat((dep, F.linear(feat, sd[pre + "init_embed.weight"], sd[pre + "init_embed.bias"])), -2)
    layer = 0
    while f"encoder.net.layers.{layer}.0.module.Wqkv.weight" in sd:
        p = f"encoder.net.layers.{layer}."
        h = _norm(sd, p + "1.normalizer.", h + _mha(sd, p + "0.module.", h), eps)      # SkipConnection(MHA) -> Normalization
        h = _norm(sd, p + "3.normalizer.", h + _mlp(sd, p + "2.module.", h), eps)      # SkipConnection(MLP) -> Normalization
        layer += 1
    return h


# One function per reference module: the temporaries of a module die when it returns, as they do in the reference (their
# lifetime decides how often the host allocator hands out fresh pages for the 50 - 210 MB activations).
def _mha(sd, p, x):
    B, N, _ = x.shape
    q, k, v = F.linear(x, sd[p + "Wqkv.weight"], sd[p + "Wqkv.bias"]).view(B, N, 3, H, E // H).permute(2, 0, 3, 1, 4).unbind(0)
    out = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0)
    return F.linear(out.permute(0, 2, 1, 3).reshape(B, N, E), sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def _mlp(sd, p, x):
    x = F.relu(F.linear(x, sd[p + "lins.0.weight"], sd[p + "lins.0.bias"]))
    return F.linear(x, sd[p + "lins.1.weight"], sd[p + "lins.1.bias"])


def _norm(sd, p, x, eps):
    if p + "running_mean" in sd:      # Normalization("batch") in eval mode: BatchNorm1d over the flattened rows
        return F.batch_norm(x.view(-1, x.size(-1)), sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                            False, 0.1, eps).view(*x.size())
    return F.instance_norm(x.permute(0, 2, 1), None, None, sd[p + "weight"], sd[p + "bias"], True, 0.1, eps).permute(0, 2, 1)


def rollout(sd, env_name, td, decode_type="greedy", num_starts=0, use_graph_context=True, clip=10.0, temp=1.0):
    """ConstructivePolicy.forward in inference mode.  td: post-reset dict (TSP: locs, first_node, current_node, i, action_mask;
    CVRP: locs incl. depot, demand, current_node, used_capacity, vehicle_capacity, visited, action_mask; + done).
    -> dict(reward, log_likelihood, actions)."""
    sd = {k: torch.as_tensor(v) for k, v in sd.items()}
    td = dict(td)
    emb = encode(sd, env_name, td)
    # AttentionModelDecoder._precompute_cache
    K0, V0, L0 = F.linear(emb, sd["decoder.project_node_embeddings.weight"]).chunk(3, dim=-1)
    gctx = F.linear(emb.mean(1), sd["decoder.project_fixed_context.weight"]) if use_graph_context else 0
    Wctx, Wout = sd["decoder.context_embedding.project_context.weight"], sd["decoder.pointer.project_out.weight"]
    B, M, _ = emb.shape
    S = int(num_starts)
    pre_lp, pre_a = [], []
    if S > 1:       # multistart pre-decoder hook: the start action, then every tensor repeated S times in (s b) order
        start = (torch.arange(S).repeat_interleave(B) % (M if env_name == "tsp" else M - 1)) + (0 if env_name == "tsp" else 1)
        td = {k: (v.unsqueeze(0).expand(S, *v.shape).contiguous().view(S * v.shape[0], *v.shape[1:]) if torch.is_tensor(v) else v)
              for k, v in td.items()}
        td["action"] = start
        td = _step(env_name, td)
        pre_lp, pre_a = [torch.zeros(S * B)], [start]
    logprobs, actions = list(pre_lp), list(pre_a)
    sampling = "sampling" in decode_type
    steps = 0
    while not td["done"].all():
        tdv = td
        if S > 1:   # decoder.forward: unbatchify so that the S queries of an instance share its keys
            tdv = {k: (v.view(S, B, *v.shape[1:]).permute(1, 0, *range(2, v.dim() + 1)).contiguous() if torch.is_tensor(v) else v)
                   for k, v in td.items()}
        # ---- context embedding + query (_compute_q) ----
        if env_name == "tsp":
            if tdv["i"][(0,) * tdv["i"].dim()].item() < 1:
                ctx = sd["decoder.context_embedding.W_placeholder"][None, :].expand(B, 2 * E) if S <= 1 else \
                    sd["decoder.context_embedding.W_placeholder"][None, None, :].expand(B, S, 2 * E)
            else:
                idx = torch.stack([tdv["first_node"], tdv["current_node"]], -1).view(B, -1)
                ctx = emb.gather(1, idx[..., None].expand(B, idx.shape[1], E)).view(B, *((S, -1) if S > 1 else (-1,)))
        else:
            cur = tdv["current_node"].view(B, -1)
            cur_emb = emb.gather(1, cur[..., None].expand(B, cur.shape[1], E))
            cur_emb = cur_emb.view(B, S, E) if S > 1 else cur_emb.squeeze(1)
            ctx = torch.cat([cur_emb, tdv["vehicle_capacity"] - tdv["used_capacity"]], -1)
        q = F.linear(ctx, Wctx) + (gctx.unsqueeze(1) if (S > 1 and torch.is_tensor(gctx)) else gctx)
        q = q.unsqueeze(1) if q.ndim == 2 else q
        # ---- _compute_kvl: StaticEmbedding returns (0, 0, 0); `cache + 0` materialises three [B, M, E] tensors per step ----
        Kc, Vc, Lc = K0 + 0, V0 + 0, L0 + 0
        # ---- PointerAttention ----
        mask = tdv["action_mask"]
        am = mask.unsqueeze(1) if mask.ndim == 3 else mask.unsqueeze(1).unsqueeze(2)
        heads = F.scaled_dot_product_attention(_heads(q), _heads(Kc), _heads(Vc), attn_mask=am)
        heads = heads.permute(0, 2, 1, 3).reshape(B, q.shape[1], E)
        glimpse = F.linear(heads, Wout)
        logits = torch.bmm(glimpse, Lc.squeeze(-2).transpose(-2, -1)).squeeze(-2) / math.sqrt(E)
        assert not torch.isnan(logits).any(), "Logits contain NaNs"
        if S > 1:
            logits = logits.permute(1, 0, 2).reshape(S * B, M)
            mask = mask.permute(1, 0, 2).reshape(S * B, M)
        # ---- process_logits + selection (DecodingStrategy.step) ----
        if clip > 0:
            logits = torch.tanh(logits) * clip
        logits[~mask] = float("-inf")
        logits = logits / temp
        lp = F.log_softmax(logits, dim=-1)
        if sampling:
            sel = torch.multinomial(lp.exp(), 1).squeeze(1)
        else:
            sel = lp.argmax(dim=-1)
        assert not (~mask).gather(1, sel.unsqueeze(-1)).data.any(), "infeasible action selected"
        logprobs.append(lp.gather(1, sel[:, None]).squeeze(1))
        actions.append(sel)
        td["action"] = sel
        td = _step(env_name, td)
        steps += 1
    lps, acts = torch.stack(logprobs, 1), torch.stack(actions, 1)
    reward = _reward(env_name, td, acts)
    ll = lps.sum(1)
    assert (ll > -1000).data.all(), "Logprobs should not be -inf, check sampling procedure!"
    return {"reward": reward, "log_likelihood": ll, "actions": acts, "steps": steps}


def _step(env_name, td):
    a = td["action"]
    if env_name == "tsp":       # TSPEnv._step
        first = a if td["i"].all() == 0 else td["first_node"]
        avail = td["action_mask"].scatter(-1, a.unsqueeze(-1).expand_as(td["action_mask"]), 0)
        done = torch.sum(avail, dim=-1) == 0
        td.update({"first_node": first, "current_node": a, "i": td["i"] + 1, "action_mask": avail,
                   "reward": torch.zeros_like(done), "done": done})
        return td
    cur = a[:, None]            # CVRPEnv._step + get_action_mask
    n_loc = td["demand"].size(-1)
    sel_dem = td["demand"].gather(1, torch.clamp(cur - 1, 0, n_loc - 1))
    used = (td["used_capacity"] + sel_dem) * (cur != 0).float()
    visited = td["visited"].scatter(-1, cur, 1)
    done = visited.sum(-1) == visited.size(-1)
    td.update({"current_node": cur, "used_capacity": used, "visited": visited, "reward": torch.zeros_like(done), "done": done})
    td["action_mask"] = cvrp_mask(td)
    return td


def cvrp_mask(td):
    exceeds = td["demand"] + td["used_capacity"] > td["vehicle_capacity"] + 1e-5
    mask_loc = td["visited"][..., 1:].to(exceeds.dtype) | exceeds
    mask_depot = (td["current_node"] == 0) & ((mask_loc == 0).int().sum(-1) > 0)[:, None]
    return ~torch.cat((mask_depot, mask_loc), -1)


def _reward(env_name, td, actions):
    B = td["locs"].shape[0]
    S = actions.shape[0] // B
    locs = td["locs"]
    if env_name == "tsp":       # validity check + closed tour length
        assert (torch.arange(actions.size(1)).view(1, -1).expand_as(actions) == actions.sort(1)[0]).all(), "Invalid tour"
        ordered = locs.gather(1, actions[..., None].expand(*actions.shape, 2))
    else:
        _check_cvrp(td, actions)
        ordered = torch.cat([locs[..., 0:1, :], locs.gather(1, actions[..., None].expand(*actions.shape, 2))], 1)
    return -(torch.roll(ordered, -1, dims=-2) - ordered).norm(p=2, dim=-1).sum(-1)


def _check_cvrp(td, actions):
    B, N = td["demand"].shape
    srt = actions.sort(1)[0]
    assert (torch.arange(1, N + 1).view(1, -1).expand(B, N) == srt[:, -N:]).all() and (srt[:, :-N] == 0).all(), "Invalid tour"
    dem = torch.cat((-td["vehicle_capacity"], td["demand"]), 1)
    d = dem.gather(1, actions)
    used = torch.zeros_like(td["demand"][:, 0])
    for i in range(actions.size(1)):        # the reference's python loop over the steps
        used += d[:, i]
        used[used < 0] = 0
        assert (used <= td["vehicle_capacity"][:, 0] + 1e-5).all(), "Used more than capacity"


def reset_td(env_name, gen):
    """Post-reset state from a generator dict (TSPEnv._reset / CVRPEnv._reset + the `done` flag torchrl adds)."""
    if env_name == "tsp":
        B, N, _ = gen["locs"].shape
        z = torch.zeros(B, dtype=torch.int64)
        return {"locs": gen["locs"], "first_node": z, "current_node": z, "i": torch.zeros(B, 1, dtype=torch.int64),
                "action_mask": torch.ones(B, N, dtype=torch.bool), "done": torch.zeros(B, 1, dtype=torch.bool)}
    B, N, _ = gen["locs"].shape
    td = {"locs": torch.cat((gen["depot"][:, None, :], gen["locs"]), -2), "demand": gen["demand"],
          "current_node": torch.zeros(B, 1, dtype=torch.long), "used_capacity": torch.zeros(B, 1),
          "vehicle_capacity": torch.full((B, 1), 1.0), "visited": torch.zeros(B, N + 1, dtype=torch.uint8),
          "done": torch.zeros(B, 1, dtype=torch.bool)}
    td["action_mask"] = cvrp_mask(td)
    return td
