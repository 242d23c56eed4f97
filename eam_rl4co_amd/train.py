"""Training companion of the native rollout: gradients by teacher-forced re-evaluation.

The rollout itself runs on the HIP kernels without a graph.  For REINFORCE / POMO / EAM the log-likelihood of the
*chosen* actions is what must be differentiated; because the actions are known after the rollout, every decode
step's state (current / first node, visited set, used capacity) is a prefix function of the action sequence, so all
T steps are re-evaluated AT ONCE as dense batched contractions -- the same trick serves EAM's
`policy(..., actions=improved)` pass (rl4co/models/zoo/earl/model.py:179-195).
`AttentionModelPolicy.forward(phase="train")` attaches this re-evaluation to the `log_likelihood` it returns
(`attach_log_likelihood_grad`): the value is the native rollout's, the gradient is the re-evaluation's, so the
reference's trainers (`out["log_likelihood"]` -> loss -> backward, reinforce.py:79-106, pomo/model.py:89-112) run on
the policy unchanged.  This module is the gradient path only: it is not used by inference, benchmarks or parity tests
of the rollout, and its forward values are checked against the native log-probs (tests/test_gpu_next.py,
tests/test_gpu_train.py).

Reference call sites: rl4co/models/rl/reinforce/reinforce.py:59-106 (REINFORCE.shared_step / calculate_loss),
rl4co/models/rl/reinforce/baselines.py:57-61 (SharedBaseline), rl4co/models/zoo/pomo/model.py:89-148.
"""
from __future__ import annotations

import os

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .utils import unbatchify


# ------------------------------------------------------------------------------------------------------------
# differentiable encoder + cache (same parameters as the native path)
# ------------------------------------------------------------------------------------------------------------
class _InstanceNormFn(torch.autograd.Function):
    """InstanceNorm1d(affine) on [B, N, E] as it lies (eamrl_instance_norm_forward / _backward): torch's instance_norm needs
    [B, E, N] -- two transposed copies per call -- and runs MIOpen's spatial batch-norm kernels (0.49 ms per backward at
    1024 x 100 x 128; the POMO encoder has 12 of them)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        from . import ops

        xc = x.contiguous()
        y, mean, rstd = ops.instance_norm_forward(xc, weight, bias, eps)
        ctx.save_for_backward(xc, mean, rstd, weight)
        ctx.affine = (weight is not None and weight.requires_grad, bias is not None and bias.requires_grad)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import ops

        x, mean, rstd, weight = ctx.saved_tensors
        dx, dg, db = ops.instance_norm_backward(x, dy.contiguous(), mean, rstd, weight, need_affine_grads=any(ctx.affine))
        return dx, (dg if ctx.affine[0] else None), (db if ctx.affine[1] else None), None


class _BatchNormTrainFn(torch.autograd.Function):
    """BatchNorm1d with BATCH statistics on [rows, E] (policy.train(), nn/ops.py:45-47): forward eamrl_batchnorm_train -- the
    kernels, and therefore the bits, of the native rollout's encoder --, backward eamrl_batchnorm_backward.  The running
    statistics are not touched here (the rollout updated them once for this forward)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        from . import ops

        y = x.clone(memory_format=torch.contiguous_format)          # the kernel normalises in place; x is kept for the backward
        _, mean, var = ops.batchnorm_train_(y, weight.detach(), bias.detach(), None, None, 0.0, eps)
        ctx.save_for_backward(x.contiguous(), mean, var, weight)
        ctx.eps = eps
        ctx.affine = weight.requires_grad or bias.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import ops

        x, mean, var, weight = ctx.saved_tensors
        dx, dg, db = ops.batchnorm_backward(x, dy.contiguous(), mean, var, weight.detach(), ctx.eps, need_affine_grads=ctx.affine)
        return dx, dg, db, None


class _SmallLinearFn(torch.autograd.Function):
    """The init embeddings' Linear(2 .. 8 -> E) on [rows, K] node features: forward eamrl_linear's tiny-K kernel (the rollout's
    own init embedding), weight / bias gradient eamrl_small_linear_wgrad.  The features are inputs: no input gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from . import ops

        xc = x.contiguous()
        ctx.save_for_backward(xc)
        ctx.has_bias = bias is not None
        return ops.linear(xc, weight.detach(), None if bias is None else bias.detach())

    @staticmethod
    def backward(ctx, dy):
        from . import ops

        (x,) = ctx.saved_tensors
        dW, db = ops.small_linear_wgrad(dy.contiguous(), x, need_bias=ctx.has_bias)
        return None, dW, db


def _small_linear(x, weight, bias):
    """F.linear for the init embeddings; native where the features need no gradient (they are the instance)."""
    if (x.is_cuda and x.dtype == torch.float32 and not x.requires_grad and weight.shape[1] <= 8 and weight.is_contiguous()
            and os.environ.get("EAMRL_TORCH_INIT_EMBED", "0") != "1"):
        return _SmallLinearFn.apply(x, weight, bias)
    return F.linear(x, weight, bias)


class _LinearFn(torch.autograd.Function):
    """torch.nn.Linear (+ ReLU) of the training graph on the package's own GEMMs: forward eamrl_linear (the rollout's kernel),
    input gradient eamrl_linear on the transposed weight, weight / bias gradient eamrl_linear_wgrad.  hipBLASLt runs these fp32 shapes
    (rows = B * N, 128 .. 512 columns) at 29-40 TFLOP/s; the fp32-MFMA kernels here at 70-95."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, residual):
        from . import ops

        xc = x if x.stride(-1) == 1 else x.contiguous()
        y = ops.linear(xc, weight, bias, relu=relu, residual=None if residual is None else residual.contiguous())
        ctx.save_for_backward(xc, weight, y if relu else None)
        ctx.relu, ctx.has_bias = relu, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import ops

        x, weight, y = ctx.saved_tensors
        g = dy.contiguous()
        if ctx.relu:
            g = torch.ops.aten.threshold_backward(g, y, 0.0)
        # dx = g W as eamrl_linear on the transposed copy of W (out x in floats, a 3 us copy): the [out][in]-weight variant of
        # the GEMM kernel runs these shapes 30 % faster than the right-multiplication one (eamrl_matmul_right)
        dx = ops.linear(g, weight.t().contiguous()) if ctx.needs_input_grad[0] else None
        dW = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dW, db = ops.linear_wgrad(g, x, need_bias=ctx.has_bias)
        return dx, dW, db, None, (dy if ctx.needs_input_grad[4] else None)


def _linear(x, weight, bias=None, relu=False, residual=None):
    """[residual +] F.linear (+ F.relu) -- on the native GEMMs where they apply (CUDA fp32, both dims multiples of 128)."""
    if (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.is_contiguous()
            and weight.shape[0] % 128 == 0 and weight.shape[1] % 128 == 0 and os.environ.get("EAMRL_TORCH_LINEAR", "0") != "1"):
        return _LinearFn.apply(x, weight, bias, relu, residual)
    y = F.linear(x, weight, bias)
    y = F.relu(y) if relu else y
    return y if residual is None else residual + y


class _SelfAttentionFn(torch.autograd.Function):
    """The encoder's multi-head self-attention on the packed projection qkv [B, N, 3E] ("b s (three h d)", as Wqkv writes it):
    forward eamrl_mha_encoder (the rollout's kernel), backward eamrl_mha_encoder_backward, which recomputes the softmax from
    qkv -- nothing but qkv is kept.  torch's scaled_dot_product_attention needs [B, H, N, D] operands (permuted copies in and
    out) and its flash backward takes 0.7 ms per layer at 1024 x 100 nodes."""

    @staticmethod
    def forward(ctx, qkv, num_heads):
        from . import ops

        q = qkv.contiguous()
        ctx.save_for_backward(q)
        ctx.H = num_heads
        return ops.mha_encoder(q, num_heads)

    @staticmethod
    def backward(ctx, dout):
        from . import ops

        (q,) = ctx.saved_tensors
        return ops.mha_encoder_backward(q, dout.contiguous(), ctx.H), None


def _self_attention(qkv, B, N, E, H):
    """qkv [B, N, 3E] -> [B, N, E]"""
    from . import ops

    if (qkv.is_cuda and qkv.dtype == torch.float32 and os.environ.get("EAMRL_TORCH_ATTENTION", "0") != "1"
            and N <= 112 and ops.mha_encoder_backward_supported(N, E, H)):
        return _SelfAttentionFn.apply(qkv, H)
    q = qkv.view(B, N, 3, H, E // H).permute(2, 0, 3, 1, 4)
    return F.scaled_dot_product_attention(q[0], q[1], q[2]).permute(0, 2, 1, 3).reshape(B, N, E)


def _normalize(norm: nn.Module, x: torch.Tensor, training: bool) -> torch.Tensor:
    n = norm.normalizer
    if isinstance(n, nn.BatchNorm1d):
        # Batch statistics when training, as the reference (nn/ops.py:45-47) and as the native rollout that drew the
        # actions (ops.batchnorm_train_); the running statistics were updated by that rollout, not again here.
        x2 = x.reshape(-1, x.size(-1))
        if training and x2.is_cuda and x2.dtype == torch.float32 and n.weight is not None and n.bias is not None and x2.size(-1) % 4 == 0 \
                and os.environ.get("EAMRL_TORCH_BATCHNORM", "0") != "1":
            y = _BatchNormTrainFn.apply(x2, n.weight, n.bias, float(n.eps))
        elif training:
            y = F.batch_norm(x2, None, None, n.weight, n.bias, True, 0.0, n.eps)
        else:
            y = F.batch_norm(x2, n.running_mean, n.running_var, n.weight, n.bias, False, 0.0, n.eps)
        return y.view_as(x)
    if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and n.weight is not None and n.bias is not None
            and os.environ.get("EAMRL_TORCH_INSTANCE_NORM", "0") != "1"):
        return _InstanceNormFn.apply(x, n.weight, n.bias, float(n.eps))
    return F.instance_norm(x.permute(0, 2, 1), weight=n.weight, bias=n.bias, eps=n.eps).permute(0, 2, 1)


def init_embedding_autograd(policy, td):
    """The init embedding of the training graph [B, M, E] (differentiable w.r.t. the init-embedding weights)."""
    enc = policy.encoder
    ie = enc.init_embedding
    locs = td["locs"]
    native_init = locs.is_cuda and os.environ.get("EAMRL_TORCH_INIT_EMBED", "0") != "1"
    if policy.env_name == "tsp":
        h = _small_linear(locs, ie.init_embed.weight, ie.init_embed.bias)
    else:
        depot = _small_linear(locs[:, :1], ie.init_embed_depot.weight, ie.init_embed_depot.bias)
        if policy.env_name == "pctsp":
            feat = torch.cat((locs[:, 1:], td["expected_prize"][..., None], td["penalty"][..., 1:, None]), -1)
        elif policy.env_name == "op":
            feat = torch.cat((locs[:, 1:], td["prize"][..., 1:, None]), -1)
        elif policy.env_name == "cvrptw":
            feat = torch.cat((locs[:, 1:], td["demand"][..., None], td["time_windows"][..., 1:, :].float(),
                              td["durations"][..., 1:, None]), -1)
        else:
            feat = torch.cat((locs[:, 1:], td["demand"][..., None]), -1)
        h = torch.cat((depot, _small_linear(feat, ie.init_embed.weight, ie.init_embed.bias)), 1)
    if locs.is_cuda and not native_init:
        # (cross-check path, EAMRL_TORCH_INIT_EMBED=1) the VALUE of the rollout's own init embedding (its kernels' rounding), the
        # gradient of torch's expression above: x + (y - y) is exactly x.  By default `_small_linear` already IS the rollout's
        # kernel, so -- the Linears / attention / norms below being the rollout's kernels too -- the embeddings of this graph equal
        # the native encoder's bit for bit (test_training_graph_encoder_equals_native_encoder) and one encoder pass can serve
        # rollout and gradient.
        with torch.no_grad():
            h_native = ie(td)
        h = h_native + (h - h.detach())
    return h


def encode_autograd(policy, td):
    enc = policy.encoder
    h = init_embedding_autograd(policy, td)
    training = policy.training
    for layer in enc.net.layers:
        mha, ffn = layer[0].module, layer[2].module
        B, N, E = h.shape
        H = mha.num_heads
        att = _self_attention(_linear(h, mha.Wqkv.weight, mha.Wqkv.bias), B, N, E, H)
        h = _normalize(layer[1], _linear(att, mha.out_proj.weight, mha.out_proj.bias, residual=h), training)
        x = h
        for lin in ffn.lins[:-1]:
            x = _linear(x, lin.weight, lin.bias, relu=True)
        h = _normalize(layer[3], _linear(x, ffn.lins[-1].weight, ffn.lins[-1].bias, residual=h), training)
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
# differentiable decoder inputs and the per-chunk log-probabilities
# ------------------------------------------------------------------------------------------------------------
_STATE_KEYS = {"cvrp": ("demand", "vehicle_capacity"), "sdvrp": ("demand", "vehicle_capacity"),
               "cvrptw": ("demand", "vehicle_capacity", "locs", "time_windows", "durations"),
               "op": ("locs", "max_length"), "pctsp": ("real_prize", "prize_required"), "tsp": ()}


class shared_decoder_tensors:
    """Within this context, policy calls on the SAME instances under the SAME parameters share one differentiable
    encoder pass (`decoder_tensors`) and one native encoder + cache launch (`AttentionModelPolicy._enqueue`): the EAM
    step evaluates the sampled and the improved tours of a batch one after the other (zoo/earl/model.py:179-195 runs the
    whole policy, encoder included, twice), and both losses are summed before the one backward -- so one encoder graph
    serves both, with identical gradients up to summation order.  Opt-in, because a graph that has already been
    backpropagated through cannot serve a second backward."""

    def __init__(self, policy):
        self.policy = policy

    def __enter__(self):
        self.policy._shared_dt = {}
        return self

    def __exit__(self, *exc):
        self.policy._shared_dt = None
        return False


def _graph_key(policy, td):
    """Identity of (instances, parameters, mode): storage address, in-place version and shape of every input tensor, the sums
    of the same over the parameters (an optimizer step or load_state_dict changes them) and, in eval mode, the buffers (the
    running statistics of batch norm are read only then; in train mode every native rollout updates them in place)."""
    ins = tuple((k, v.data_ptr(), v._version, tuple(v.shape)) for k, v in sorted(td.items(), key=lambda kv: kv[0])
                if torch.is_tensor(v))
    # (walking the module tree costs 0.3 ms per call and a step asks several times: the lists of Parameter / buffer objects
    # are kept -- .to(), load_state_dict and optimizer steps keep the objects; the key is only ever compared within one step)
    lists = policy.__dict__.get("_key_tensors")
    if lists is None:
        lists = (list(policy.parameters()), list(policy.buffers()))
        policy.__dict__["_key_tensors"] = lists
    ver = ptr = 0
    for p_ in lists[0] + ([] if policy.training else lists[1]):
        ver += p_._version
        ptr += p_.data_ptr()
    return ins, ver, ptr, policy.training, torch.is_grad_enabled()


def graph_encoder_equals_native(policy, td) -> bool:
    """True where `encode_autograd` reproduces the native encoder's embeddings bit for bit AND may stand in for it: instance-norm
    layers (their training forward kernel sums in the rollout's order), every Linear, the self-attention and the norms on this
    library's kernels (none of the EAMRL_TORCH_* switches), fp32 on the GPU.  Batch-norm policies keep two passes: their graph
    runs on the same kernels since round 3 (`_BatchNormTrainFn`), but the rollout's pass is the one that updates the running
    statistics."""
    if any(os.environ.get(k, "0") == "1" for k in ("EAMRL_TORCH_LINEAR", "EAMRL_TORCH_ATTENTION", "EAMRL_TORCH_INSTANCE_NORM",
                                                  "EAMRL_SEPARATE_ENCODER_PASSES")):
        return False
    locs = td["locs"]
    enc = getattr(policy, "encoder", None)
    layers = getattr(getattr(enc, "net", None), "layers", None)
    if layers is None or not locs.is_cuda or locs.dtype != torch.float32 or locs.shape[1] > 112:
        return False
    for layer in layers:
        mha, ffn = layer[0].module, layer[2].module
        E = mha.Wqkv.weight.shape[1]
        if E != 128 or mha.num_heads != 8 or len(ffn.lins) != 2 or ffn.lins[0].weight.shape[0] % 128:
            return False
        for norm in (layer[1], layer[3]):
            n = norm.normalizer
            if not isinstance(n, nn.InstanceNorm1d) or n.weight is None or n.bias is None:
                return False
    return True


def decoder_tensors(policy, td):
    """The differentiable tensors the decode steps read: node embeddings, glimpse key / value, logit key, graph
    context (encoder + `_precompute_cache` with autograd) and the decoder's own parameters.  -> dict name -> tensor."""
    shared = getattr(policy, "_shared_dt", None)
    key = None
    if shared is not None:
        key = _graph_key(policy, td)
        ent = shared.get("graph")
        if ent is not None and ent[0] == key:
            return ent[1]
    dec = policy.decoder
    emb = encode_autograd(policy, td)
    K, V, L = _linear(emb, dec.project_node_embeddings.weight).chunk(3, dim=-1)
    t = {"emb": emb, "K": K, "V": V, "L": L, "Wctx": dec.context_embedding.project_context.weight,
         "Wout": dec.pointer.project_out.weight}
    if dec.use_graph_context:
        t["gctx"] = F.linear(emb.mean(1), dec.project_fixed_context.weight)
    if policy.env_name == "tsp":
        t["placeholder"] = dec.context_embedding.W_placeholder
    if policy.env_name == "sdvrp":
        t["dyn"] = dec.dynamic_embedding.projection.weight
    if shared is not None:
        shared["graph"] = (key, t)
    return t


def _filter_logits_(logits, act, top_k: int, top_p: float):
    """process_logits' top-k / top-p filtering (utils/decoding.py:111-137,170-176) as a fixed, non-differentiated choice of the
    entries that stay: top-k keeps an entry iff fewer than k are strictly larger, top-p sorts ascending and drops the entries
    whose cumulative probability stays <= 1 - top_p.  The chosen action always stays (it did in the rollout;
    a last-bit difference between torch's sums and the rollout kernel's at the threshold must not turn its log-prob into -inf)."""
    with torch.no_grad():
        drop = torch.zeros_like(logits, dtype=torch.bool)
        if top_k > 0:
            k = min(int(top_k), logits.shape[-1])
            kth = torch.topk(logits, k, dim=-1).values[..., -1:]
            drop |= logits < kth
        if 0.0 < top_p < 1.0:
            x = logits.masked_fill(drop, float("-inf"))
            srt, idx = torch.sort(x, dim=-1, descending=False, stable=True)
            cum = torch.softmax(srt, dim=-1).cumsum(-1)
            drop |= torch.zeros_like(drop).scatter_(-1, idx, cum <= (1.0 - top_p))
        drop.scatter_(-1, act[..., None], False)
    return logits.masked_fill(drop, float("-inf"))


def _logp_rows(env_name, t, static, act, nrep, multistart, H, temperature, clip, top_k=0, top_p=0.0):
    """Per-step log-probabilities [Rc, T] of the rows `act` (nrep whole start groups of the B instances, (s b) order).
    `t`: decoder_tensors; `static`: the instance tensors the masks are rebuilt from."""
    emb, K, V, L = t["emb"], t["K"], t["V"], t["L"]
    B, M, E = emb.shape
    D = E // H
    rep = lambda x: x.repeat(nrep, *([1] * (x.dim() - 1)))   # (s b) order: instance index = row % B
    embr, Kr, Vr, Lr = rep(emb), rep(K), rep(V), rep(L)
    Rc, T = act.shape
    ar = torch.arange(Rc, device=act.device)[:, None]
    dem_t = None
    with torch.no_grad():
        if env_name == "tsp":
            first, cur, mask = _tsp_states(act, M, multistart)
            rem = now = None
        elif env_name == "cvrptw":
            cur, rem, now, mask = _cvrptw_states(act, rep(static["demand"]), rep(static["vehicle_capacity"].reshape(-1)),
                                                 rep(static["locs"]), rep(static["time_windows"].float()),
                                                 rep(static["durations"].float()))
        elif env_name == "op":
            cur, rem, mask = _op_states(act, rep(static["locs"]), rep(static["max_length"]))
        elif env_name == "pctsp":
            cur, rem, mask = _pctsp_states(act, rep(static["real_prize"]), rep(static["prize_required"].reshape(-1)))
        elif env_name == "cvrp":
            cur, rem, mask = _cvrp_states(act, rep(static["demand"]), rep(static["vehicle_capacity"].reshape(-1)), M)
        else:
            cur, rem, mask, dem_t = _sdvrp_states(act, rep(static["demand"]), rep(static["vehicle_capacity"].reshape(-1)), M)
    if env_name == "tsp":
        ctx_in = torch.cat((embr[ar, first], embr[ar, cur]), -1)                       # [Rc, T, 2E]
        if not multistart:   # step 0 uses the learned placeholder (context.py:118-131)
            ctx_in = torch.cat((t["placeholder"].expand(Rc, 1, 2 * E), ctx_in[:, 1:]), 1)
    elif env_name == "cvrptw":
        ctx_in = torch.cat((embr[ar, cur], rem[..., None], now[..., None]), -1)          # [Rc, T, E+2]
    else:
        ctx_in = torch.cat((embr[ar, cur], rem[..., None]), -1)                          # [Rc, T, E+1]
    q = F.linear(ctx_in, t["Wctx"])
    if "gctx" in t:
        q = q + rep(t["gctx"])[:, None, :]
    qh = q.view(Rc, T, H, D).permute(0, 2, 1, 3)
    kh = Kr.view(Rc, M, H, D).permute(0, 2, 1, 3)
    vh = Vr.view(Rc, M, H, D).permute(0, 2, 1, 3)
    if env_name == "sdvrp":
        # dynamic embedding (dynamic.py:59-78): K/V/L rows + remaining demand * projection columns.  The update is
        # rank one, so it enters as a score bias and two outer products instead of [Rc, T, M, E] tensors
        wk, wv, wl = t["dyn"].view(3, E)
        qw = (qh * wk.view(1, H, 1, D)).sum(-1)                                              # [Rc, H, T]
        bias = (dem_t[:, None] * qw[..., None]) / math.sqrt(D)                               # [Rc, H, T, M]
        bias = bias.masked_fill(~mask[:, None], float("-inf"))
        att = torch.softmax(torch.matmul(qh, kh.transpose(-1, -2)) / math.sqrt(D) + bias, dim=-1)
        heads = torch.matmul(att, vh) + (att * dem_t[:, None]).sum(-1, keepdim=True) * wv.view(1, H, 1, D)
    else:
        heads = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=mask[:, None])     # [Rc, H, T, D]
    glimpse = F.linear(heads.permute(0, 2, 1, 3).reshape(Rc, T, E), t["Wout"])
    logits = torch.bmm(glimpse, Lr.transpose(1, 2))
    if env_name == "sdvrp":
        logits = logits + dem_t * (glimpse @ wl)[..., None]
    logits = logits / math.sqrt(E)
    if clip > 0:
        logits = torch.tanh(logits) * clip
    logits = logits.masked_fill(~mask, float("-inf")) / temperature
    if top_k > 0 or 0.0 < top_p < 1.0:
        logits = _filter_logits_(logits, act, top_k, top_p)
    logp = F.log_softmax(logits, dim=-1).gather(-1, act[..., None]).squeeze(-1)
    if multistart:   # the start node is not a decision (decoding.py:318-324)
        logp = torch.cat((torch.zeros_like(logp[:, :1]), logp[:, 1:]), 1)
        # steps t >= 1 were evaluated with the state after the start action: column t uses prefix a_{<t} as built
    return logp


class _ChunkedReeval(torch.autograd.Function):
    """logp = cat_c fn(c, *tensors) with activation memory bounded to one chunk: the forward keeps no graph, the
    backward recomputes each chunk with autograd and accumulates the gradients of `tensors`."""

    @staticmethod
    def forward(ctx, fn, n_chunks, *tensors):
        with torch.no_grad():
            outs = [fn(c, *tensors) for c in range(n_chunks)]
        ctx.fn, ctx.n_chunks = fn, n_chunks
        ctx.save_for_backward(*tensors)
        return torch.cat(outs, 0)

    @staticmethod
    def backward(ctx, g):
        leaves = [x.detach().requires_grad_(need) for x, need in zip(ctx.saved_tensors, ctx.needs_input_grad[2:])]
        need = [x for x in leaves if x.requires_grad]
        acc = [None] * len(need)
        off = 0
        for c in range(ctx.n_chunks):
            with torch.enable_grad():
                out = ctx.fn(c, *leaves)
            gs = torch.autograd.grad(out, need, g[off:off + out.shape[0]], allow_unused=True)
            off += out.shape[0]
            for i, gi in enumerate(gs):
                if gi is not None:
                    acc[i] = gi if acc[i] is None else acc[i].add_(gi)
        it = iter(acc)
        return (None, None) + tuple(next(it) if x.requires_grad else None for x in leaves)


# ------------------------------------------------------------------------------------------------------------
# native (HIP) re-evaluation: forward and backward of all decode steps on fp32 MFMA (csrc/reeval.hip)
# ------------------------------------------------------------------------------------------------------------
_NATIVE_ENVS = ("tsp", "cvrp", "pctsp", "op", "cvrptw", "sdvrp")


def native_reeval_supported(policy, M: int) -> bool:
    import os

    from . import ops

    dec = policy.decoder
    return (os.environ.get("EAMRL_NATIVE_REEVAL", "1") != "0" and policy.env_name in _NATIVE_ENVS
            and ops.reeval_supported(M, dec.embed_dim, dec.num_heads))


@torch.no_grad()
def replay_states(policy, td, actions, S: int, multistart: bool):
    """What the env state is before every decode step of the given action rows, in the form the re-evaluation kernels
    read it: mask bits [R, T, 4], the node(s) whose folded context rows enter the query (idxA / idxB [R, T], -1 = none) and
    the state scalars sc [NC, R, T] multiplying the state columns of project_context.  TSP is closed-form in the
    actions; the other envs replay their transition kernels step by step (eamrl_*_step_mask + eamrl_pack_mask_bits).
    -> dict(maskbits, idxA, idxB, sc, tstart, placeholder)"""
    from . import ops
    from .policy import _env_step_, state_from_td

    env_name = policy.env_name
    R, T = actions.shape
    dev = actions.device
    actions = actions.contiguous()
    if env_name == "tsp":
        M = td["locs"].shape[1]
        bits = ops.tsp_mask_bits(actions, M) if M <= ops.KEY_CHUNK else ops.tsp_mask_bits_chunked(actions, M)
        a32 = actions.to(torch.int32)
        prev = torch.cat((torch.full((R, 1), -1, dtype=torch.int32, device=dev), a32[:, :-1]), 1)       # a_{t-1}
        first = a32[:, :1].expand(R, T).contiguous()
        sc, placeholder = None, False
        if not multistart:       # step 0: the learned placeholder replaces both gathered rows (context.py:118-131)
            first = first.clone()
            first[:, 0] = -1
            sc = torch.zeros(1, R, T, dtype=torch.float32, device=dev)
            sc[0, :, 0] = 1.0
            placeholder = True
        return dict(maskbits=bits, idxA=first, idxB=prev.contiguous(), sc=sc, tstart=1 if multistart else 0,
                    placeholder=placeholder)
    B = td["action_mask"].shape[0]
    if env_name == "sdvrp":     # + the remaining demands of every step (the dynamic embedding's input)
        st = state_from_td(env_name, td, S, copy=False)
        bits, idxA, sc, rem = ops.replay_states_sdvrp(st, actions)
        return dict(maskbits=bits, idxA=idxA, idxB=None, sc=sc, tstart=1 if multistart else 0, placeholder=False, rem=rem)
    big = td["action_mask"].shape[-1] > ops.KEY_CHUNK            # key-chunked kernels: their mask layout (either way)
    if os.environ.get("EAMRL_REPLAY_LOOP", "0") != "1":         # one launch: the env's transitions replayed inside a kernel
        st = state_from_td(env_name, td, S, copy=False)         # read-only
        bits, idxA, sc = ops.replay_states(st, actions, B)
        return dict(maskbits=bits, idxA=idxA, idxB=None, sc=sc, tstart=1 if multistart else 0, placeholder=False)
    st = state_from_td(env_name, td, S)                         # (the step-by-step form, kept as the cross-check)
    NC = 2 if env_name == "cvrptw" else 1
    bits = torch.empty((R, T, -(-st.M // ops.KEY_CHUNK), 4) if big else (R, T, 4), dtype=torch.int32, device=dev)
    idxA = torch.empty(R, T, dtype=torch.int32, device=dev)
    sc = torch.empty(NC, R, T, dtype=torch.float32, device=dev)
    for t in range(T):
        (ops.pack_mask_bits_chunked_ if big else ops.pack_mask_bits_)(st.mask, bits, t)
        idxA[:, t] = st.cur
        if env_name == "pctsp":          # prize still to collect, clamped at 0 (context.py:194-208)
            sc[0, :, t] = (st.vcap - st.used).clamp_(min=0)
        else:                            # free capacity (cvrp, cvrptw) / length still allowed (op: vcap = max_length[:, 0])
            sc[0, :, t] = st.vcap - st.used
        if env_name == "cvrptw":
            sc[1, :, t] = st.time
        _env_step_(st, actions[:, t].contiguous())
    return dict(maskbits=bits, idxA=idxA, idxB=None, sc=sc, tstart=1 if multistart else 0, placeholder=False)


class _NativeReeval(torch.autograd.Function):
    """logp [R, T] = eamrl_reeval_forward(K, V, Lp, Pa, Pb, gctx, cvec | replayed states); backward = eamrl_reeval_backward."""

    @staticmethod
    def forward(ctx, K, V, Lp, Pa, Pb, gctx, cvec, meta, dyn=None):
        from . import ops

        parts = [K, V, Lp, Pa] + ([Pb] if Pb is not None else [])
        buf = torch.cat([x.detach() for x in parts], -1).contiguous()
        plan = ops.ReevalPlan(buf, Pb is not None, None if gctx is None else gctx.detach().contiguous(),
                              None if cvec is None else cvec.detach().contiguous(), meta["idxA"], meta["idxB"], meta["sc"],
                              meta["maskbits"], meta["actions"], meta["S"], meta["tstart"], meta["clip"], meta["temp"],
                              rollout_logp=meta.get("rollout_logp"), rollout_heads=meta.get("rollout_heads"),
                              rem=meta.get("rem"), dyn=None if dyn is None else dyn.detach().contiguous())
        ctx.plan = plan
        ctx.has = (Pb is not None, gctx is not None, cvec is not None, dyn is not None)
        return plan.forward()

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        dbuf, dg, dc = plan.backward(g)
        plan.heads = None           # 5 .. 10 GB at the POMO sizes: not kept for as long as the graph object lives
        E = plan.E
        sl = [dbuf[..., i * E:(i + 1) * E] for i in range(5 if ctx.has[0] else 4)]
        return (sl[0], sl[1], sl[2], sl[3], sl[4] if ctx.has[0] else None, dg if ctx.has[1] else None,
                dc if ctx.has[2] else None, None, plan.ddyn if ctx.has[3] else None)


def _evaluate_native(policy, t, td, actions, S, multistart, temperature, clip, rollout_logp=None, rollout_heads=None):
    """Differentiable log-probs [R, T]: the weight folds (Lp = L Wout, Pa / Pb = emb x halves of project_context, state
    columns) as small autograd GEMMs on the decoder tensors `t`, everything per (row, step) in the HIP kernels."""
    E = t["emb"].shape[-1]
    Wctx = t["Wctx"]
    Lp = _linear(t["L"], t["Wout"].t().contiguous())
    Pa = _linear(t["emb"], Wctx[:, :E].contiguous())
    Pb = _linear(t["emb"], Wctx[:, E:2 * E].contiguous()) if policy.env_name == "tsp" else None
    meta = replay_states(policy, td, actions, S, multistart)
    if policy.env_name == "tsp":
        cvec = F.linear(t["placeholder"][None], Wctx) if meta["placeholder"] else None            # [1, E]
    else:
        cvec = Wctx[:, E:].t()                                                                      # [NC, E] state columns
    meta.update(actions=actions.contiguous(), S=S, clip=float(clip), temp=float(temperature))
    if rollout_logp is not None:
        meta["rollout_logp"] = rollout_logp.detach().to(torch.float32).contiguous()
        if rollout_heads is not None:           # the rollout kernel's glimpse outputs of these very steps: not recomputed
            meta["rollout_heads"] = rollout_heads.detach()
    dyn = None
    if policy.env_name == "sdvrp":      # wk | wv | lw: the logit-key column folded through project_out like Lp (dynamic.py:59-78)
        wk, wv, wl = t["dyn"].view(3, E)
        dyn = torch.stack((wk, wv, wl @ t["Wout"]))
    return _NativeReeval.apply(t["K"], t["V"], Lp, Pa, Pb, t.get("gctx"), cvec, meta, dyn)


def evaluate_log_likelihood(policy, td, env, actions, num_starts: int = 0, temperature=None, tanh_clipping=None,
                            chunk_rows: int = 4096, multistart=None, checkpoint=None, native=None, rollout_logp=None,
                            rollout_heads=None, top_k: int = 0, top_p: float = 0.0):
    """Differentiable per-step log-probabilities of `actions` [R, T] (R = B or S*B rows in (s b) order).  native
    (default: where supported -- TSP / CVRP / PCTSP / OP / CVRPTW, graphs up to 112 nodes): forward and backward of the
    decode steps run in the HIP re-evaluation kernels; otherwise (and as the cross-check) PyTorch autograd ops.  multistart
    (default: num_starts > 1): the first column is the start node and gets log-prob 0; False with num_starts > 1 is
    the multi-sample layout (every column a decision).  checkpoint (default: by size): keep no activations in the
    forward and recompute chunk by chunk in the backward.  rollout_logp [R, T] (native path only): the per-step log-probs
    the rollout kernel produced for exactly these actions with this policy -- the HIP forward pass is then skipped (the
    returned values are these; the backward recovers each step's normaliser from them).  top_k / top_p: the rollout filtered
    its logits (process_logits, decoding.py:170-176): the same entries are dropped here, as a fixed mask, before the
    log-softmax (PyTorch path).  Returns logp [R, T]."""
    temperature = policy.temperature if temperature is None else temperature
    clip = policy.tanh_clipping if tanh_clipping is None else tanh_clipping
    H = policy.decoder.num_heads
    t = decoder_tensors(policy, td)
    B, M, _ = t["emb"].shape
    R, T = actions.shape
    S = max(int(num_starts), 1)
    assert R == S * B
    if multistart is None:
        multistart = S > 1
    env_name = policy.env_name
    filtering = top_k > 0 or 0.0 < top_p < 1.0
    if native is None:
        native = native_reeval_supported(policy, M) and not filtering
    if native:
        if filtering:
            raise ValueError("evaluate_log_likelihood: the native re-evaluation kernels do not filter (top_k / top_p)")
        return _evaluate_native(policy, t, td, actions, S, multistart, temperature, clip, rollout_logp, rollout_heads)
    static = {k: td[k] for k in _STATE_KEYS[env_name]}
    # rows are processed in chunks of whole start-groups so that memory stays bounded ([rows, H, T, M] scores)
    starts_per_chunk = max(1, chunk_rows // B)
    bounds = [(s0, min(S, s0 + starts_per_chunk)) for s0 in range(0, S, starts_per_chunk)]
    names = list(t)

    def chunk(c, *tensors):
        s0, s1 = bounds[c]
        return _logp_rows(env_name, dict(zip(names, tensors)), static, actions[s0 * B:s1 * B], s1 - s0, multistart, H,
                          temperature, clip, top_k, top_p)

    if checkpoint is None:      # ~6 live [rows, H, T, M] fp32 tensors per chunk if the graph is kept
        checkpoint = len(bounds) > 1 and 24.0 * R * H * T * M > 8e9
    if checkpoint and torch.is_grad_enabled():
        return _ChunkedReeval.apply(chunk, len(bounds), *[t[k] for k in names])
    return torch.cat([chunk(c, *[t[k] for k in names]) for c in range(len(bounds))], 0)


def attach_log_likelihood_grad(policy, td, env, actions, native_logp, num_starts: int = 0, multistart=None,
                               temperature=None, tanh_clipping=None):
    """[R, T] per-step log-probs whose VALUE is the native rollout's `native_logp` and whose GRADIENT is the
    re-evaluation's: native + (re - re.detach()).  This is what `AttentionModelPolicy.forward(phase="train")` returns
    under autograd, the counterpart of the reference's graph-attached logprobs (constructive/base.py:236-263)."""
    re = evaluate_log_likelihood(policy, td, env, actions, num_starts=num_starts, multistart=multistart,
                                 temperature=temperature, tanh_clipping=tanh_clipping)
    return native_logp.detach() + (re - re.detach())


def reinforce_loss(policy, env, td, baseline: str = "shared", num_starts: int = 0, decode_type: str = None,
                   **rollout_kwargs):
    """One REINFORCE forward as the reference's trainers run it (REINFORCE.shared_step + calculate_loss,
    reinforce.py:59-106; POMO.shared_step, pomo/model.py:89-112): `policy(td, env, phase="train")` under autograd ->
    loss = -((reward - bl) * log_likelihood).mean().

    baseline: "shared" (POMO: mean over the starts of an instance, needs num_starts > 1), "mean" (batch mean) or
    "no".  Returns dict(loss, reward, log_likelihood, actions, ...)."""
    if decode_type is None:
        decode_type = "multistart_sampling" if num_starts > 1 else "sampling"
    kw = dict(rollout_kwargs)
    if num_starts > 1:
        kw["num_starts"] = num_starts
    with torch.enable_grad():
        out = policy(td, env, phase="train", decode_type=decode_type, return_sum_log_likelihood=False, **kw)
    actions, reward, logp = out["actions"], out["reward"], out["log_likelihood"]
    # finished CVRP rows are padded with depot visits of probability 1: their log-prob is 0 and carries no gradient
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
            "native_log_likelihood": ll.detach()}


def eam_loss(policy, env, td, ea, num_starts: int, improve: bool = True, draws=None, generator=None, return_entropy: bool = True):
    """One EAM training step of the fork with the POMO (shared) baseline (rl4co/models/zoo/earl/model.py:129-247):

    1. sampled multistart rollout, `policy(td, env, phase="train", num_starts=S)` (log-likelihood with grad); with
       return_entropy=True also the entropy of the sampled tours, as the reference's first call asks for (model.py:150-156;
       it only logs it) -- res["entropy"];
    2. the sampled tours are improved by the evolutionary operators on the GPU (`evolution_worker`; the reference
       ships them to CPU threads, model.py:166-171) and get their start column back (`_align_improved_actions`);
    3. the improved tours are evaluated by `policy(td, env, phase="train", actions=improved)` (model.py:189-195);
    4. REINFORCE with the per-instance mean over starts as baseline, over the concatenation [original; improved]
       treated as 2B instances (model.py:226-244, reinforce.py:103-106).

    Returns dict(loss, reward, improved_reward, log_likelihood, improved_log_likelihood, actions, improved_actions)."""
    S = int(num_starts)
    assert S > 1, "the EAM step uses the multistart (shared) baseline"
    with shared_decoder_tensors(policy):        # one differentiable encoder pass for the sampled and the improved tours
        return _eam_loss(policy, env, td, ea, S, improve, draws, generator, return_entropy)


def _eam_loss(policy, env, td, ea, S, improve, draws, generator, return_entropy=False):
    from .evolution import evolution_worker

    with torch.enable_grad():
        out = policy(td, env, phase="train", decode_type="multistart_sampling", num_starts=S, return_entropy=return_entropy)
    actions, reward, ll = out["actions"], out["reward"], out["log_likelihood"]
    rs, lls = [unbatchify(reward, S)], [unbatchify(ll, S)]
    res = {"reward": reward, "log_likelihood": ll, "actions": actions}
    if return_entropy:
        res["entropy"] = out["entropy"]
    if improve:
        with torch.no_grad():
            improved, _ = evolution_worker(actions, td, ea, env, draws=draws, generator=generator)
            if improved.shape[-1] + 1 == actions.shape[-1]:          # _align_improved_actions
                improved = torch.cat([actions[:, :1], improved], dim=-1)
        with torch.enable_grad():
            out2 = policy(td, env, phase="train", actions=improved, num_starts=S, decode_type="multistart_sampling")
        r_imp, ll_imp = out2["reward"], out2["log_likelihood"]
        rs.append(unbatchify(r_imp, S))
        lls.append(unbatchify(ll_imp, S))
        res.update(improved_actions=improved, improved_reward=r_imp, improved_log_likelihood=ll_imp)
    r_all, ll_all = torch.cat(rs, 0), torch.cat(lls, 0)              # [B or 2B, S]
    adv = r_all - r_all.mean(1, keepdim=True)
    res["loss"] = -(adv * ll_all).mean()
    return res


def problem_symmetricity_loss(reward, log_likelihood, dim=1):
    """zoo/symnco/losses.py:5-15: REINFORCE with the mean over `dim` as baseline (0 when that axis has fewer than 2 entries)."""
    if reward.shape[dim] < 2:
        return 0
    return (-(reward - reward.mean(dim=dim, keepdim=True)) * log_likelihood).mean()


def solution_symmetricity_loss(reward, log_likelihood, dim=-1):
    """zoo/symnco/losses.py:18-28: the same over the last axis."""
    return problem_symmetricity_loss(reward, log_likelihood, dim)


def invariance_loss(proj_embed, num_augment):
    """zoo/symnco/losses.py:31-39: mean cosine similarity between the projected embeddings of an instance's first copy and its
    other copies -- with the reference's own grouping "(b a) ... -> b a ..." of the rows."""
    pe = proj_embed.reshape(proj_embed.shape[0] // num_augment, num_augment, *proj_embed.shape[1:])
    sim = sum(F.cosine_similarity(pe[:, 0], pe[:, i], dim=-1) for i in range(1, num_augment))
    return sim.mean()


def symeam_loss(policy, env, td, ea, num_augment: int = 4, num_starts: int = 0, alpha: float = 0.2, beta: float = 1.0,
                improve: bool = True, augment=None, draws=None, generator=None):
    """The training branch of `SymEAM.shared_step`, the fork's second trainer (rl4co/models/zoo/earl/model.py:535-660), on a
    `SymNCOPolicy`: symmetric augmentation of the batch (num_augment copies) -> sampled rollout with entropy -> (with
    probability `improve_prob`, decided by the caller: `improve`) evolutionary improvement of the sampled tours and their
    teacher-forced evaluation -> loss = L_ps + beta L_ss + alpha L_inv over [original; improved] (zoo/symnco/losses.py).
    The reference's expressions are kept as written, including which loss is switched by which count (`loss_ps` needs
    num_starts > 1, `loss_ss` num_augment > 1) and the row grouping of the invariance loss.  num_starts > 1: the improved
    tours get their start column back before the evaluation, as in `eam_loss`.
    Returns dict(loss, loss_ps, loss_ss, loss_inv, reward, log_likelihood, actions[, improved_*])."""
    from .evolution import evolution_worker
    from .utils import StateAugmentation

    n_aug, n_start = int(num_augment), int(num_starts)
    if n_aug > 1:
        td = (augment or StateAugmentation(num_augment=n_aug))(td)
    init_td = td.clone()
    kw = dict(num_starts=n_start) if n_start > 1 else {}
    with torch.enable_grad():
        out = policy(td, env, phase="train", return_entropy=True, **kw)
    res = {"reward": out["reward"], "log_likelihood": out["log_likelihood"], "actions": out["actions"], "entropy": out.get("entropy")}
    rs, lls, projs = [unbatchify(out["reward"], (n_aug, n_start))], [unbatchify(out["log_likelihood"], (n_aug, n_start))], \
        [out["proj_embeddings"]]
    if improve:
        with torch.no_grad():
            improved, _ = evolution_worker(out["actions"], init_td, ea, env, draws=draws, generator=generator)
            if improved is not None and improved.shape[-1] + 1 == out["actions"].shape[-1]:
                improved = torch.cat([out["actions"][:, :1], improved], dim=-1)
        if improved is not None:
            with torch.enable_grad():
                out2 = policy(init_td, env, phase="train", actions=improved.to(out["actions"].device), **kw)
            rs.append(unbatchify(out2["reward"], (n_aug, n_start)))
            lls.append(unbatchify(out2["log_likelihood"], (n_aug, n_start)))
            projs.append(out2["proj_embeddings"])
            res.update(improved_actions=improved, improved_reward=out2["reward"], improved_log_likelihood=out2["log_likelihood"])
    reward, ll, proj = torch.cat(rs, 0), torch.cat(lls, 0), torch.cat(projs, 0)
    loss_ps = problem_symmetricity_loss(reward, ll) if n_start > 1 else 0
    loss_ss = solution_symmetricity_loss(reward, ll) if n_aug > 1 else 0
    loss_inv = invariance_loss(proj, n_aug) if n_aug > 1 else 0
    res.update(loss=loss_ps + beta * loss_ss + alpha * loss_inv, loss_ps=loss_ps, loss_ss=loss_ss, loss_inv=loss_inv)
    return res


class PolicyGradientStep:
    """One optimizer step of the reference's REINFORCE / POMO training, data-parallel (BASELINE.json configs[3]):

        out = policy(td, env, phase="train"[, num_starts=S])            # POMO.shared_step, pomo/model.py:89-112
        loss = -((reward - baseline) * log_likelihood).mean()            # reinforce.py:103-106, baselines.py:57-61
        loss.backward() -> ONE flat all-reduce (mean over ranks) -> clip_grad_norm 1.0 -> Adam(lr 1e-4, wd 1e-6)
                                                                         # utils/trainer.py:55,72-89; configs/experiment/routing/pomo.yaml

    Each rank owns its own instances (all starts of an instance on one GPU); the gradient buffer is the only thing
    exchanged (dist.FlatGradBuffer: p.grad are views, no copy kernels around the collective).  At construction every rank
    takes rank 0's parameters and buffers (as DDP does), so the ranks may have been seeded differently; from then on every
    rank ends a step with bit-identical parameters: the reduced buffer is identical on all ranks and the update is
    deterministic.  With batch normalisation in train() mode the running statistics, which each rank updates from its own
    shard, are averaged over the ranks after the step (`sync_buffers`; DDP broadcasts rank 0's instead)."""

    def __init__(self, policy, env, num_starts: int = 0, baseline: str = None, lr: float = 1e-4, weight_decay: float = 1e-6,
                 max_grad_norm: float = 1.0, optimizer=None):
        from .dist import FlatGradBuffer, broadcast_module_state

        broadcast_module_state(policy, src=0)        # before the optimizer reads the parameters
        self.sync_buffers = any(b.is_floating_point() for b in policy.buffers())
        self.policy, self.env, self.S = policy, env, int(num_starts)
        self.baseline = baseline or ("shared" if self.S > 1 else "mean")
        self.max_grad_norm = max_grad_norm
        self.grads = FlatGradBuffer(policy)
        policy._flat_grads = self.grads
        self.optimizer = optimizer or torch.optim.Adam(policy.parameters(), lr=lr, weight_decay=weight_decay)

    def __call__(self, td, **rollout_kwargs) -> dict:
        self.grads.zero_()
        out = reinforce_loss(self.policy, self.env, td, baseline=self.baseline, num_starts=self.S, **rollout_kwargs)
        out["loss"].backward()
        if not self.grads.attached():
            raise RuntimeError("PolicyGradientStep: p.grad is no longer a view of the flat buffer (zero_grad(set_to_none)?)")
        self.grads.allreduce(average=True)
        out["grad_norm"] = self.grads.clip_(self.max_grad_norm) if self.max_grad_norm else None
        self.optimizer.step()
        if self.sync_buffers and self.policy.training:
            from .dist import allreduce_buffers

            allreduce_buffers(self.policy)
        return out
