"""torch.Tensor front-end of the C ABI: checks device / dtype / contiguity / shapes on the host (a kernel
that faults can reset the GPU host) and enqueues on torch's current HIP stream.  Tensors are plumbing only:
every computation below happens in libeamrl_hip.so."""
from __future__ import annotations

import ctypes as C
import logging
import os

import torch

from . import _lib
from ._lib import (ENV_CVRP, ENV_CVRPTW, ENV_OP, ENV_PCTSP, ENV_SDVRP, ENV_TSP, EVALUATE, GREEDY, NORM_BATCH_EVAL, NORM_INSTANCE, SAMPLE,  # noqa: F401
                   ST_INFEASIBLE, ST_NAN_LOGITS, ST_STEP_OVERRUN)

MODES = {"greedy": GREEDY, "sampling": SAMPLE, "evaluate": EVALUATE}
ENVS = {"tsp": ENV_TSP, "cvrp": ENV_CVRP, "sdvrp": ENV_SDVRP, "pctsp": ENV_PCTSP, "op": ENV_OP, "cvrptw": ENV_CVRPTW}


def _need_gpu(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: the eam_rl4co_amd rollout path runs only on an MI355X (HIP) device; "
            "there is no CPU fallback. Move the TensorDict / policy to 'cuda'.")


def _chk(t, name, dtype, shape=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a tensor")
    _need_gpu(t, name)
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _bytes(t):
    """bool tensors are passed as their uint8 storage."""
    return t.view(torch.uint8) if t.dtype == torch.bool else t


# ------------------------------------------------------------------------------------------------------
# encoder / cache
# ------------------------------------------------------------------------------------------------------
def linear(x, W, bias=None, relu=False, residual=None, out=None, w_cols=None, bn=None):
    """y = [residual +] act(x @ W[:, w_cols].T + bias), optionally followed by BatchNorm1d(eval) in the same launch
    (bn = (gamma, beta, running_mean, running_var, eps)).  x [..., in] (last dim contiguous, rows strided ok)."""
    lib = _lib.load()
    _need_gpu(x, "x")
    in_dim = x.shape[-1] if w_cols is None else w_cols[1] - w_cols[0]
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    if W.dtype != torch.float32 or W.stride(-1) != 1 or x2.dtype != torch.float32:
        raise TypeError("linear: fp32 tensors with unit inner stride required")
    Wv = W if w_cols is None else W[:, w_cols[0]:w_cols[1]]
    if Wv.shape[1] != in_dim or x2.shape[1] != in_dim:
        raise ValueError(f"linear: in_dim mismatch {tuple(x2.shape)} vs {tuple(Wv.shape)}")
    out_dim = Wv.shape[0]
    rows = x2.shape[0]
    if out is None:
        out = torch.empty(*x.shape[:-1], out_dim, device=x.device, dtype=torch.float32)
    o2 = out.reshape(-1, out.shape[-1]) if out.dim() != 2 else out
    if o2.stride(-1) != 1 or o2.shape[0] != rows or o2.shape[1] < out_dim and o2.shape[1] != out_dim:
        raise ValueError("linear: bad output buffer")
    res2 = None
    if residual is not None:
        res2 = residual.reshape(-1, residual.shape[-1])
        if res2.stride(-1) != 1 or res2.shape != (rows, out_dim) or res2.dtype != torch.float32:
            raise ValueError("linear: bad residual")
    if bias is not None:
        _chk(bias, "bias", torch.float32, (out_dim,))
    if bn is not None:
        if relu:
            raise ValueError("linear: relu and fused batch-norm are not combined")
        gamma, beta, mean, var, eps = bn
        for nm, t in (("gamma", gamma), ("beta", beta), ("running_mean", mean), ("running_var", var)):
            _chk(t, nm, torch.float32, (out_dim,))
        _lib.check(lib.eamrl_linear_bn(_ptr(x2), x2.stride(0), _ptr(Wv), Wv.stride(0), _ptr(bias), _ptr(res2),
                                       res2.stride(0) if res2 is not None else 0, _ptr(o2), o2.stride(0), rows, in_dim,
                                       out_dim, _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(var), float(eps), _stream(x)),
                   "eamrl_linear_bn")
        return out
    _lib.check(lib.eamrl_linear(_ptr(x2), x2.stride(0), _ptr(Wv), Wv.stride(0), _ptr(bias), _ptr(res2),
                                res2.stride(0) if res2 is not None else 0, _ptr(o2), o2.stride(0), rows, in_dim,
                                out_dim, int(relu), _stream(x)), "eamrl_linear")
    return out


def matmul_right(x, Wt, out=None):
    """y = x @ Wt   (Wt [in][out], contiguous)."""
    lib = _lib.load()
    _chk(Wt, "Wt", torch.float32)
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1 or x2.dtype != torch.float32:
        raise ValueError("matmul_right: bad x")
    _need_gpu(x2, "x")
    rows, in_dim = x2.shape
    if Wt.shape[0] != in_dim:
        raise ValueError("matmul_right: in_dim mismatch")
    out_dim = Wt.shape[1]
    if out is None:
        out = torch.empty(*x.shape[:-1], out_dim, device=x.device, dtype=torch.float32)
    o2 = out.reshape(-1, out.shape[-1]) if out.dim() != 2 else out
    if o2.stride(-1) != 1 or o2.shape != (rows, out_dim):
        raise ValueError("matmul_right: bad output buffer")
    _lib.check(lib.eamrl_matmul_right(_ptr(x2), x2.stride(0), _ptr(Wt), _ptr(o2), o2.stride(0), rows, in_dim, out_dim,
                                      _stream(x)), "eamrl_matmul_right")
    return out


def linear_wgrad_supported(out_dim: int, in_dim: int) -> bool:
    return _lib.load().eamrl_linear_wgrad_scratch(0, int(out_dim), int(in_dim)) >= 0


def linear_wgrad(dy, x, need_bias=True):
    """dW [out, in] = dy^T x over all rows, db [out] = column sums of dy (eamrl_linear_wgrad); dy [..., out], x [..., in]."""
    lib = _lib.load()
    dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
    for nm, t_ in (("dy", dy2), ("x", x2)):
        _need_gpu(t_, nm)
        if t_.dtype != torch.float32 or t_.stride(-1) != 1:
            raise TypeError(f"linear_wgrad: {nm} must be fp32 with unit inner stride")
    rows, out_dim = dy2.shape
    if x2.shape[0] != rows:
        raise ValueError("linear_wgrad: row mismatch")
    in_dim = x2.shape[1]
    need = lib.eamrl_linear_wgrad_scratch(rows, out_dim, in_dim)
    if need < 0:
        raise ValueError(f"linear_wgrad: shape {out_dim} x {in_dim} not supported (multiples of 128)")
    dW = torch.empty(out_dim, in_dim, dtype=torch.float32, device=dy.device)
    db = torch.empty(out_dim, dtype=torch.float32, device=dy.device) if need_bias else None
    scratch = torch.empty(max(int(need), 4), dtype=torch.float32, device=dy.device)
    _lib.check(lib.eamrl_linear_wgrad(_ptr(dy2), dy2.stride(0), _ptr(x2), x2.stride(0), rows, out_dim, in_dim, _ptr(dW), _ptr(db),
                                      _ptr(scratch), scratch.numel(), _stream(dy)), "eamrl_linear_wgrad")
    return dW, db


def augment_xy(xy, code, cs=None, offset=0.5):
    """Coordinates [B, N, 2] -> [R, N, 2] with row r = a * B + b the instance b transformed by code[r] (eamrl_augment_xy:
    0..7 dihedral variants, 8 rotation by the angle with (cos, sin) = cs[r], 9 rotation + x <-> y)."""
    lib = _lib.load()
    _chk(xy, "xy", torch.float32)
    B, N, two = xy.shape
    if two != 2:
        raise ValueError("augment_xy: coordinates must be [B, N, 2]")
    _chk(code, "code", torch.int32)
    R = code.shape[0]
    if cs is not None:
        _chk(cs, "cs", torch.float32, (R, 2))
    out = torch.empty(R, N, 2, dtype=torch.float32, device=xy.device)
    _lib.check(lib.eamrl_augment_xy(_ptr(xy), _ptr(cs), _ptr(code), _ptr(out), R, B, N, float(offset), _stream(xy)),
               "eamrl_augment_xy")
    return out


def small_linear_wgrad(dy, x, need_bias=True):
    """dW [out, K] = dy^T x, db [out] for a Linear with K <= 8 inputs (eamrl_small_linear_wgrad); dy [..., out], x [..., K]."""
    lib = _lib.load()
    dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
    for nm, t_ in (("dy", dy2), ("x", x2)):
        _need_gpu(t_, nm)
        if t_.dtype != torch.float32 or t_.stride(-1) != 1:
            raise TypeError(f"small_linear_wgrad: {nm} must be fp32 with unit inner stride")
    rows, out_dim = dy2.shape
    K = x2.shape[1]
    if x2.shape[0] != rows or not 1 <= K <= 8:
        raise ValueError("small_linear_wgrad: row mismatch or more than 8 input features")
    dW = torch.empty(out_dim, K, dtype=torch.float32, device=dy.device)
    db = torch.empty(out_dim, dtype=torch.float32, device=dy.device) if need_bias else None
    scratch = torch.empty(max(int(lib.eamrl_small_linear_wgrad_scratch(rows, out_dim)), 4), dtype=torch.float32, device=dy.device)
    _lib.check(lib.eamrl_small_linear_wgrad(_ptr(dy2), dy2.stride(0), _ptr(x2), x2.stride(0), rows, out_dim, K, _ptr(dW), _ptr(db),
                                            _ptr(scratch), scratch.numel(), _stream(dy)), "eamrl_small_linear_wgrad")
    return dW, db


def batchnorm_backward(x, dy, save_mean, save_var, gamma, eps, need_affine_grads=True):
    """Gradient of batchnorm_train_ (eamrl_batchnorm_backward): x = the normalisation's input [..., E] -> (dx, dgamma, dbeta)."""
    lib = _lib.load()
    _chk(x, "x", torch.float32)
    _chk(dy, "dy", torch.float32, tuple(x.shape))
    E = x.shape[-1]
    rows = x.numel() // E
    dx = torch.empty_like(x)
    dg = torch.empty(E, dtype=torch.float32, device=x.device) if need_affine_grads else None
    db = torch.empty(E, dtype=torch.float32, device=x.device) if need_affine_grads else None
    scratch = torch.empty(max(int(lib.eamrl_batchnorm_backward_scratch(rows, E)), 4), dtype=torch.float32, device=x.device)
    _lib.check(lib.eamrl_batchnorm_backward(_ptr(x), _ptr(dy), _ptr(save_mean), _ptr(save_var), _ptr(gamma), float(eps), rows, E,
                                            _ptr(dx), _ptr(dg), _ptr(db), _ptr(scratch), scratch.numel(), _stream(x)),
               "eamrl_batchnorm_backward")
    return dx, dg, db


def mha_encoder(qkv, num_heads):
    lib = _lib.load()
    _chk(qkv, "qkv", torch.float32)
    B, N, E3 = qkv.shape
    out = torch.empty(B, N, E3 // 3, device=qkv.device, dtype=torch.float32)
    _lib.check(lib.eamrl_mha_encoder(_ptr(qkv), _ptr(out), B, N, E3 // 3, num_heads, _stream(qkv)), "eamrl_mha_encoder")
    return out


def mha_encoder_backward_supported(N: int, E: int, H: int) -> bool:
    return bool(_lib.load().eamrl_mha_encoder_backward_supported(int(N), int(E), int(H)))


def mha_encoder_backward(qkv, dout, num_heads):
    """dqkv [B, N, 3E] of mha_encoder(qkv) for the output gradient dout [B, N, E] (eamrl_mha_encoder_backward)."""
    lib = _lib.load()
    _chk(qkv, "qkv", torch.float32)
    B, N, E3 = qkv.shape
    _chk(dout, "dout", torch.float32, (B, N, E3 // 3))
    dqkv = torch.empty_like(qkv)
    _lib.check(lib.eamrl_mha_encoder_backward(_ptr(qkv), _ptr(dout), _ptr(dqkv), B, N, E3 // 3, num_heads, _stream(qkv)),
               "eamrl_mha_encoder_backward")
    return dqkv


def normalize_(x, kind, gamma, beta, mean=None, var=None, eps=1e-5):
    lib = _lib.load()
    _chk(x, "x", torch.float32)
    B, N, E = x.shape
    for nm, t in (("gamma", gamma), ("beta", beta)):
        _chk(t, nm, torch.float32, (E,))
    if kind == NORM_BATCH_EVAL:
        _chk(mean, "running_mean", torch.float32, (E,))
        _chk(var, "running_var", torch.float32, (E,))
    _lib.check(lib.eamrl_normalize(_ptr(x), B, N, E, kind, _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(var), eps,
                                   _stream(x)), "eamrl_normalize")
    return x


def instance_norm_forward(x, gamma, beta, eps):
    """-> (y, mean, rstd): InstanceNorm1d(affine) over the nodes of x [B, N, E], kept statistics [B, E] for the backward."""
    lib = _lib.load()
    _chk(x, "x", torch.float32)
    B, N, E = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(B, E, dtype=torch.float32, device=x.device)
    rstd = torch.empty(B, E, dtype=torch.float32, device=x.device)
    _lib.check(lib.eamrl_instance_norm_forward(_ptr(x), _ptr(y), _ptr(mean), _ptr(rstd), B, N, E, _ptr(gamma), _ptr(beta),
                                               float(eps), _stream(x)), "eamrl_instance_norm_forward")
    return y, mean, rstd


def instance_norm_backward(x, dy, mean, rstd, gamma, need_affine_grads=True):
    """-> (dx, dgamma | None, dbeta | None)"""
    lib = _lib.load()
    _chk(x, "x", torch.float32)
    _chk(dy, "dy", torch.float32, tuple(x.shape))
    B, N, E = x.shape
    dx = torch.empty_like(x)
    dg = torch.zeros(E, dtype=torch.float32, device=x.device) if need_affine_grads else None
    db = torch.zeros(E, dtype=torch.float32, device=x.device) if need_affine_grads else None
    _lib.check(lib.eamrl_instance_norm_backward(_ptr(x), _ptr(dy), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(dx), _ptr(dg),
                                                _ptr(db), B, N, E, _stream(x)), "eamrl_instance_norm_backward")
    return dx, dg, db


def batchnorm_train_(x, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5):
    """BatchNorm1d with batch statistics over all rows of x [..., E], in place; running stats updated as torch does.
    -> (x, batch_mean [E], batch_var [E] (biased))."""
    lib = _lib.load()
    _chk(x, "x", torch.float32)
    E = x.shape[-1]
    rows = x.numel() // E
    for nm, t in (("gamma", gamma), ("beta", beta)):
        _chk(t, nm, torch.float32, (E,))
    if (running_mean is None) != (running_var is None):
        raise ValueError("batchnorm_train: running_mean and running_var come together")
    if running_mean is not None:
        _chk(running_mean, "running_mean", torch.float32, (E,))
        _chk(running_var, "running_var", torch.float32, (E,))
    nws = ((rows + 127) // 128) * E
    ws = torch.empty(nws + 2 * E, device=x.device, dtype=torch.float32)
    save_mean, save_var = ws[nws:nws + E], ws[nws + E:]
    _lib.check(lib.eamrl_batchnorm_train(_ptr(x), rows, E, _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                         float(momentum), float(eps), _ptr(save_mean), _ptr(save_var), _ptr(ws), nws,
                                         _stream(x)), "eamrl_batchnorm_train")
    return x, save_mean, save_var


def mean_nodes(emb):
    lib = _lib.load()
    _chk(emb, "emb", torch.float32)
    B, M, E = emb.shape
    out = torch.empty(B, E, device=emb.device, dtype=torch.float32)
    _lib.check(lib.eamrl_mean_nodes(_ptr(emb), _ptr(out), B, M, E, _stream(emb)), "eamrl_mean_nodes")
    return out


def pack_linear_weight(W, out=None):
    """torch.nn.Linear.weight [out, in] -> the packed MFMA fragment order the fused encoder reads (eamrl_pack_linear_weight)."""
    lib = _lib.load()
    _chk(W, "weight", torch.float32)
    N, K = W.shape
    if out is None:
        out = torch.empty(N * K, device=W.device, dtype=torch.float32)
    _chk(out, "packed weight", torch.float32, (N * K,))
    _lib.check(lib.eamrl_pack_linear_weight(_ptr(W), _ptr(out), N, K, _stream(W)), "eamrl_pack_linear_weight")
    return out


def encoder_fused_supported(M, E, H, ff_hidden, nlayers) -> bool:
    return bool(_lib.load().eamrl_encoder_fused_supported(int(M), int(E), int(H), int(ff_hidden), int(nlayers)))


def encoder_fused(h, layers, num_heads, ff_hidden, norm, eps, cache=None, init=None, store_hidden=True):
    """All encoder layers of every instance in one launch.  h [B, M, E]; layers: list of dicts with the 16 fields of
    struct eamrl_encoder_layer (packed weights, biases, norm parameters; running stats may be None for instance norm).
    cache = (Wc_packed, WoutT_packed, out [B, M, ld], nproj[, Wg [E, E], gctx [B, E]]): also fill the slot-major decoder
    cache and, with the last two, the graph context (struct eamrl_encoder_cache).
    init (instead of h; struct eamrl_encoder_init) = dict(feat [B, M, F], W [E, F], b, depot [B, >= 2] or None, Wd, bd,
    want_init): the init embedding is computed inside the kernel; returns (hidden or None, init embeddings or None) then.
    store_hidden=False (with init and cache): the node embeddings never leave LDS."""
    lib = _lib.load()
    if init is not None:
        feat = init["feat"]
        _chk(feat, "node features", torch.float32)
        B, M, F = feat.shape
        W = init["W"]
        E = W.shape[0]
        _chk(W, "init_embed.weight", torch.float32, (E, F))
        dev = feat.device
    else:
        _chk(h, "h", torch.float32)
        B, M, E = h.shape
        dev = h.device
    arr = (_lib.EncoderLayer * len(layers))()
    for i, d in enumerate(layers):
        for name, _ in _lib.EncoderLayer._fields_:
            t = d.get(name)
            if t is not None:
                _need_gpu(t, name)
                if t.dtype != torch.float32 or not t.is_contiguous():
                    raise TypeError(f"encoder_fused: {name} must be contiguous fp32")
            setattr(arr[i], name, _ptr(t))
    out = torch.empty(B, M, E, dtype=torch.float32, device=dev) if (store_hidden or cache is None) else None
    cstruct = None
    if cache is not None:
        Wc, WoT, buf, nproj = cache[:4]
        Wg, gctx = cache[4:6] if len(cache) >= 6 else (None, None)
        _chk(Wc, "packed cache weights", torch.float32, (nproj * E * E,))
        _chk(WoT, "packed project_out^T", torch.float32, (E * E,))
        _chk(buf, "decoder cache", torch.float32)
        if buf.dim() != 3 or buf.shape[0] != B or buf.shape[1] != M or buf.shape[2] < (nproj + 1) * E:
            raise ValueError("encoder_fused: cache buffer must be [B, M, >= (nproj + 1) * E]")
        cs = _lib.EncoderCache()
        cs.Wc, cs.WoutT, cs.out, cs.ld, cs.nproj = _ptr(Wc), _ptr(WoT), _ptr(buf), buf.shape[2], int(nproj)
        if Wg is not None:
            _chk(Wg, "project_fixed_context.weight", torch.float32, (E, E))
            _chk(gctx, "graph context", torch.float32, (B, E))
            if Wg.data_ptr() % 16:
                raise ValueError("encoder_fused: Wg must be 16-byte aligned")
            cs.Wg, cs.gctx = _ptr(Wg), _ptr(gctx)
        cstruct = C.byref(cs)
    if init is not None:
        ist = _lib.EncoderInit()
        b, depot, Wd, bd = init.get("b"), init.get("depot"), init.get("Wd"), init.get("bd")
        for name, t in (("init_embed.bias", b), ("depot", depot), ("init_embed_depot.weight", Wd), ("init_embed_depot.bias", bd)):
            if t is not None:
                _need_gpu(t, name)
                if t.dtype != torch.float32:
                    raise TypeError(f"encoder_fused: {name} must be fp32")
        if depot is not None and (depot.dim() != 2 or depot.stride(1) != 1 or depot.shape[0] != B or Wd is None or not Wd.is_contiguous()):
            raise ValueError("encoder_fused: depot must be [B, >= 2] with unit inner stride, Wd [E, 2] contiguous")
        init_out = torch.empty(B, M, E, dtype=torch.float32, device=dev) if init.get("want_init") else None
        ist.feat, ist.F, ist.W, ist.b = _ptr(feat), int(F), _ptr(W), _ptr(b)
        ist.depot, ist.depot_ld = _ptr(depot), (depot.stride(0) if depot is not None else 0)
        ist.Wd, ist.bd, ist.init_out = _ptr(Wd), _ptr(bd), _ptr(init_out)
        _lib.check(lib.eamrl_encoder_fused_init(C.byref(ist), _ptr(out), B, M, E, int(num_heads), int(ff_hidden), len(layers),
                                                int(norm), float(eps), C.cast(arr, C.c_void_p), cstruct, _stream(feat)),
                   "eamrl_encoder_fused_init")
        return out, init_out
    _lib.check(lib.eamrl_encoder_fused(_ptr(h), _ptr(out), B, M, E, int(num_heads), int(ff_hidden), len(layers), int(norm),
                                       float(eps), C.cast(arr, C.c_void_p), cstruct, _stream(h)), "eamrl_encoder_fused")
    return out


def pointer_attention(query, key, value, logit_key, attn_mask, Wout, bout=None, num_heads=8, mask_inner=True):
    """PointerAttention.forward (nn/attention.py:282-306) in one launch.  query [B, L, E] (or [B, E]); key / value /
    logit_key [B, M, E] (views with a common row stride are fine, e.g. the chunks of one projection);
    attn_mask [B, M] or [B, L, M] bool (True = feasible) or None.  -> logits [B, L, M] ([B, M] when L == 1, as the
    reference's `.squeeze(-2)` gives)."""
    lib = _lib.load()
    if query.dim() == 2:
        query = query[:, None, :]
    _chk(query, "query", torch.float32)
    B, L, E = query.shape
    M = key.shape[-2]
    ld = key.stride(-2)
    for nm, t in (("key", key), ("value", value), ("logit_key", logit_key)):
        _need_gpu(t, nm)
        if (t.dtype != torch.float32 or t.dim() != 3 or tuple(t.shape) != (B, M, E) or t.stride(-1) != 1 or t.stride(-2) != ld
                or t.stride(0) != M * ld):
            raise ValueError(f"pointer_attention: {nm} must be fp32 [B, M, E] with unit inner stride and a common row stride")
    _chk(Wout, "project_out.weight", torch.float32, (E, E))
    if bout is not None:
        _chk(bout, "project_out.bias", torch.float32, (E,))
    per_query = 0
    if attn_mask is not None:
        per_query = int(attn_mask.dim() == 3)
        _chk(attn_mask, "attn_mask", torch.bool, (B, L, M) if per_query else (B, M))
    out = torch.empty(B, L, M, device=query.device, dtype=torch.float32)
    _lib.check(lib.eamrl_pointer_attention(_ptr(query), _ptr(key), _ptr(value), _ptr(logit_key), ld,
                                           _ptr(None if attn_mask is None else _bytes(attn_mask)), per_query, _ptr(Wout),
                                           _ptr(bout), _ptr(out), B, L, M, E, int(num_heads), int(bool(mask_inner)),
                                           _stream(query)), "eamrl_pointer_attention")
    return out.squeeze(-2) if L == 1 else out


# ------------------------------------------------------------------------------------------------------
# teacher-forced re-evaluation (training gradient path)
# ------------------------------------------------------------------------------------------------------
def reeval_supported(M, E, H) -> bool:
    return bool(_lib.load().eamrl_reeval_supported(int(M), int(E), int(H)))


def pack_mask_bits_(mask, bits, t):
    """bits[:, t] <- the bit set of mask [R, M] (bool / uint8); bits [R, T, 4] int32."""
    lib = _lib.load()
    R, M = mask.shape
    _chk(_bytes(mask), "action_mask", torch.uint8, (R, M))
    _chk(bits, "mask bits", torch.int32)
    if bits.dim() != 3 or bits.shape[0] != R or bits.shape[2] != 4:
        raise ValueError("pack_mask_bits: bits must be [R, T, 4] int32")
    _lib.check(lib.eamrl_pack_mask_bits(_ptr(_bytes(mask)), _ptr(bits), R, M, bits.shape[1], int(t), _stream(mask)),
               "eamrl_pack_mask_bits")


def replay_states(st, actions, B):
    """Depot envs (cvrp, cvrptw, pctsp, op): mask bits [R, T, 4], current node idxA [R, T] and the state scalars
    sc [NC, R, T] BEFORE every step of actions [R, T], starting from the flat state `st` (not modified) -- one launch
    (eamrl_replay_states) instead of T rounds of pack-bits / copy / step."""
    lib = _lib.load()
    _chk(actions, "actions", torch.int64)
    R, T = actions.shape
    if R != st.R or st.env_name not in ("cvrp", "cvrptw", "pctsp", "op") or st.M > 1024:
        raise ValueError("replay_states: cvrp / cvrptw / pctsp / op states of at most 1024 nodes, one action row per state row")
    dev = actions.device
    NC = 2 if st.env_name == "cvrptw" else 1
    # (graphs above 112 nodes: the chunked layout of the key-chunked re-evaluation kernels)
    bits = torch.empty((R, T, -(-st.M // 112), 4) if st.M > 112 else (R, T, 4), dtype=torch.int32, device=dev)
    idxA = torch.empty(R, T, dtype=torch.int32, device=dev)
    sc = torch.empty(NC, R, T, dtype=torch.float32, device=dev)
    ss = st.struct()
    _lib.check(lib.eamrl_replay_states(ENVS[st.env_name], C.byref(ss), R, int(B), st.M, _ptr(actions), T, _ptr(bits), _ptr(idxA),
                                       _ptr(sc), _stream(actions)), "eamrl_replay_states")
    return bits, idxA, sc


KEY_CHUNK = 112        # keys per chunk of the re-evaluation kernels (graphs above 112 nodes: csrc/reeval.hip)


def tsp_mask_bits_chunked(actions, M):
    """TSP masks of every step in the layout of the key-chunked re-evaluation: int32 [R, T, nkc, 4], bit i of chunk c = node 112 c + i."""
    lib = _lib.load()
    _chk(actions, "actions", torch.int64)
    R, T = actions.shape
    nkc = -(-int(M) // KEY_CHUNK)
    bits = torch.empty(R, T, nkc, 4, dtype=torch.int32, device=actions.device)
    _lib.check(lib.eamrl_tsp_mask_bits_chunked(_ptr(actions), _ptr(bits), R, int(M), T, _stream(actions)), "eamrl_tsp_mask_bits_chunked")
    return bits


def pack_mask_bits_chunked_(mask, bits, t):
    """bits[:, t] (int32 [R, T, nkc, 4], chunked layout) <- mask [R, M] (bool / uint8)."""
    lib = _lib.load()
    R, M = mask.shape
    _chk(bits, "mask bits", torch.int32)
    nkc = -(-int(M) // KEY_CHUNK)
    if bits.dim() != 4 or bits.shape[0] != R or bits.shape[2] != nkc or bits.shape[3] != 4:
        raise ValueError("pack_mask_bits_chunked: bits must be [R, T, nkc, 4] int32")
    _lib.check(lib.eamrl_pack_mask_bits_chunked(_ptr(_bytes(mask)), _ptr(bits), R, M, bits.shape[1], int(t), _stream(mask)),
               "eamrl_pack_mask_bits_chunked")


def replay_states_sdvrp(st, actions):
    """SDVRP: as replay_states, plus the remaining demands rem [R, T, 128] (zero padded rows) before every step -- what the
    dynamic embedding multiplies (eamrl_replay_states_sdvrp)."""
    lib = _lib.load()
    _chk(actions, "actions", torch.int64)
    R, T = actions.shape
    if R != st.R or st.env_name != "sdvrp" or st.M > 1024:
        raise ValueError("replay_states_sdvrp: an sdvrp state of at most 1024 nodes, one action row per state row")
    dev = actions.device
    big = st.M > KEY_CHUNK             # the chunked layout of the key-chunked re-evaluation kernels
    nkc = -(-st.M // KEY_CHUNK)
    bits = torch.empty((R, T, nkc, 4) if big else (R, T, 4), dtype=torch.int32, device=dev)
    idxA = torch.empty(R, T, dtype=torch.int32, device=dev)
    sc = torch.empty(1, R, T, dtype=torch.float32, device=dev)
    rem = torch.zeros(R, T, nkc, 128, dtype=torch.float32, device=dev) if big else torch.empty(R, T, 128, dtype=torch.float32, device=dev)
    ss = st.struct()
    _lib.check(lib.eamrl_replay_states_sdvrp(C.byref(ss), R, st.M, _ptr(actions), T, _ptr(bits), _ptr(idxA), _ptr(sc), _ptr(rem),
                                             _stream(actions)), "eamrl_replay_states_sdvrp")
    return bits, idxA, sc, rem


def tsp_mask_bits(actions, M):
    lib = _lib.load()
    _chk(actions, "actions", torch.int64)
    R, T = actions.shape
    bits = torch.empty(R, T, 4, dtype=torch.int32, device=actions.device)
    _lib.check(lib.eamrl_tsp_mask_bits(_ptr(actions), _ptr(bits), R, int(M), T, _stream(actions)), "eamrl_tsp_mask_bits")
    return bits


class ReevalPlan:
    """Arguments of eamrl_reeval_forward / _backward (struct eamrl_reeval) kept alive between the two calls.

    buf [B, M, P*E]: the instance operands side by side -- K | V | Lp | Pa (| Pb); gctx [B, E] or None; cvec [NC, E] or
    None; idxA / idxB int32 [R, T]; sc [NC, R, T]; maskbits int32 [R, T, 4]; actions int64 [R, T]."""

    def __init__(self, buf, has_pb, gctx, cvec, idxA, idxB, sc, maskbits, actions, S, tstart, clip, temp, rollout_logp=None,
                 slots=None, E=None, want_entropy=False, rollout_heads=None, rem=None, dyn=None):
        _chk(buf, "operands", torch.float32)
        # slots: {"K", "V", "Lp", "Pa"[, "Pb"]} -> E-wide column block of buf (default: side by side in that order); with it
        # the plan reads a decoder cache (ops.DecodeCache.buf) in place
        # (a decoder cache of a large graph keeps its operands as planes [P, B, M, E] instead of side by side [B, M, P * E])
        self.planes = buf.dim() == 4
        if self.planes:
            _, self.B, self.M, width = buf.shape
            E = width
        else:
            self.B, self.M, width = buf.shape
        self.slots = slots or {n: i for i, n in enumerate(["K", "V", "Lp", "Pa"] + (["Pb"] if has_pb else []))}
        self.E = E if E is not None else width // (5 if has_pb else 4)
        self.buf, self.has_pb, self.gctx, self.cvec = buf, has_pb, gctx, cvec
        self.entropy = None
        self.want_entropy = bool(want_entropy)
        R, T = actions.shape
        if R != S * self.B:
            raise ValueError("reeval: rows must be S * B")
        _chk(actions, "actions", torch.int64, (R, T))
        _chk(idxA, "idxA", torch.int32, (R, T))
        if has_pb:
            _chk(idxB, "idxB", torch.int32, (R, T))
        # graphs above 112 nodes: key chunks (csrc/reeval.hip) -- chunked mask layout, a scratch that lives from forward to backward,
        # the forward pass always runs (its statistics feed the backward), no rollout heads
        self.nkc = -(-self.M // KEY_CHUNK) if self.M > KEY_CHUNK else 1
        self.scratch = None
        if self.nkc > 1:
            _chk(maskbits, "mask bits", torch.int32, (R, T, self.nkc, 4))
            rollout_heads = None
        else:
            _chk(maskbits, "mask bits", torch.int32, (R, T, 4))
        self.NC = 0 if cvec is None else cvec.shape[0]
        if self.NC:
            _chk(cvec, "state columns", torch.float32, (self.NC, self.E))
            _chk(sc, "state scalars", torch.float32, (self.NC, R, T))
        if gctx is not None:
            _chk(gctx, "graph context", torch.float32, (self.B, self.E))
        self.idxA, self.idxB, self.sc, self.maskbits, self.actions = idxA, idxB, sc, maskbits, actions
        self.R, self.T, self.S, self.tstart, self.clip, self.temp = R, T, int(S), int(tstart), float(clip), float(temp)
        self.nchunk = max(1, min(self.S, -(-512 // self.B)))
        # rollout_logp [R, T]: the per-step log-probs the rollout kernel produced for exactly these actions.  Then no
        # forward pass is needed for the gradient: forward() hands them back and the backward recovers the normaliser.
        # rollout_heads [R, Th, E]: every decode step's glimpse output as the rollout kernel computed it (RolloutState.heads):
        # the backward reads them instead of recomputing the glimpse (row r, step t at [r, t - tstart])
        # SDVRP: rem [R, T, 128] remaining demands per (row, step), dyn [3, E] = wk | wv | lw of the dynamic embedding
        self.rem = self.dyn = None
        if dyn is not None:
            _chk(rem, "remaining demands", torch.float32, (R, T, 128) if self.M <= KEY_CHUNK else (R, T, -(-self.M // KEY_CHUNK), 128))
            _chk(dyn, "dynamic embedding vectors", torch.float32, (3, self.E))
            self.rem, self.dyn = rem, dyn
            rollout_heads = None        # (the kernels recompute the glimpse for this env)
        self.heads = None
        if rollout_heads is not None:
            _chk(rollout_heads, "rollout heads", torch.float32)
            if rollout_heads.dim() != 3 or rollout_heads.shape[0] != R or rollout_heads.shape[2] != self.E \
                    or rollout_heads.shape[1] < T - int(tstart):
                raise ValueError("reeval: rollout heads must be [R, >= T - tstart, E]")
            self.heads = rollout_heads
        if rollout_logp is not None and self.nkc == 1:
            _chk(rollout_logp, "rollout log-probs", torch.float32, (R, T))
            self.logp, self.lse = rollout_logp, None
        else:
            self.logp = torch.empty(R, T, dtype=torch.float32, device=buf.device)
            self.lse = torch.empty(R, T, dtype=torch.float32, device=buf.device)

    def _struct(self):
        s = _lib.Reeval()
        E4 = self.B * self.M * self.E * 4 if self.planes else self.E * 4
        base = self.buf.data_ptr()
        s.K, s.V, s.Lp, s.Pa = (C.c_void_p(base + self.slots[n] * E4) for n in ("K", "V", "Lp", "Pa"))
        s.Pb = C.c_void_p(base + self.slots["Pb"] * E4) if self.has_pb else None
        s.ld = self.buf.shape[-1]
        s.gctx, s.Cvec, s.NC = _ptr(self.gctx), _ptr(self.cvec), self.NC
        s.idxA, s.idxB, s.sc = _ptr(self.idxA), _ptr(self.idxB if self.has_pb else None), _ptr(self.sc if self.NC else None)
        s.maskbits, s.actions = _ptr(self.maskbits), _ptr(self.actions)
        s.B, s.R, s.S, s.T, s.M, s.tstart, s.nchunk = self.B, self.R, self.S, self.T, self.M, self.tstart, self.nchunk
        s.clip, s.temp = self.clip, self.temp
        s.logp, s.lse = _ptr(self.logp), _ptr(self.lse)
        s.entropy = _ptr(self.entropy)
        if self.heads is not None:
            s.heads, s.heads_T = _ptr(self.heads), self.heads.shape[1]
        if self.dyn is not None:
            s.rem, s.dyn = _ptr(self.rem), _ptr(self.dyn)
        if self.nkc > 1:
            if self.scratch is None:
                n = int(_lib.load().eamrl_reeval_scratch_floats(self.R, self.T, self.M))
                self.scratch = torch.empty(n, dtype=torch.float32, device=self.buf.device)
            s.nkc, s.scratch = self.nkc, _ptr(self.scratch)
        return s

    def forward(self):
        if self.lse is None and not self.want_entropy:
            # a new tensor object: autograd makes the Function's output point at its node, and the node holds this plan --
            # returning self.logp itself would close a reference cycle that only the cyclic collector frees
            return self.logp.view_as(self.logp)
        lib = _lib.load()
        if self.want_entropy:
            self.entropy = torch.empty(self.R, self.T, dtype=torch.float32, device=self.buf.device)
            if self.lse is None:        # (the rollout's log-probs stay untouched: the kernel writes into a scratch)
                self.logp = torch.empty(self.R, self.T, dtype=torch.float32, device=self.buf.device)
        s = self._struct()
        _lib.check(lib.eamrl_reeval_forward(C.byref(s), _stream(self.buf)), "eamrl_reeval_forward")
        return self.logp.view_as(self.logp)         # (a new tensor object, see above)

    def backward(self, glogp):
        """-> (dbuf [B, M, P*E], dgctx or None, dcvec or None)"""
        if self.planes:
            raise ValueError("reeval backward: operands side by side [B, M, P * E] (a plane-layout decoder cache serves the forward only)")
        lib = _lib.load()
        glogp = glogp.contiguous()
        _chk(glogp, "grad of logp", torch.float32, (self.R, self.T))
        dev = self.buf.device
        dbuf = torch.zeros_like(self.buf)
        # (key chunks: per-chunk partials of dheads, then the per-chunk query gradients)
        dheads = torch.empty((2 * self.nkc if self.nkc > 1 else 1) * self.R, self.T, self.E, dtype=torch.float32, device=dev)
        dg = torch.zeros_like(self.gctx) if self.gctx is not None else None
        dc = torch.zeros_like(self.cvec) if self.NC else None
        s = self._struct()
        E4 = self.E * 4
        base = dbuf.data_ptr()
        s.glogp, s.dheads = _ptr(glogp), _ptr(dheads)
        s.dK, s.dV, s.dLp, s.dPa = (C.c_void_p(base + i * E4) for i in range(4))
        s.dPb = C.c_void_p(base + 4 * E4) if self.has_pb else None
        s.ldg = dbuf.shape[2]
        s.dgctx, s.dCvec = _ptr(dg), _ptr(dc)
        self.ddyn = torch.zeros_like(self.dyn) if self.dyn is not None else None      # read by the caller (SDVRP)
        s.ddyn = _ptr(self.ddyn)
        _lib.check(lib.eamrl_reeval_backward(C.byref(s), _stream(self.buf)), "eamrl_reeval_backward")
        return dbuf, dg, dc


# ------------------------------------------------------------------------------------------------------
# environment transitions
# ------------------------------------------------------------------------------------------------------
def tsp_step_(mask, first, cur, istep, action, done):
    lib = _lib.load()
    R, N = mask.shape
    _chk(mask, "action_mask", torch.bool, (R, N))
    for nm, t in (("first_node", first), ("current_node", cur), ("i", istep), ("action", action)):
        _chk(t, nm, torch.int64)
        if t.numel() != R:
            raise ValueError(f"{nm} must have {R} elements")
    _chk(done, "done", torch.bool)
    if done.numel() != R:
        raise ValueError("done must have R elements")
    _lib.check(lib.eamrl_tsp_step(_ptr(_bytes(mask)), _ptr(first), _ptr(cur), _ptr(istep), _ptr(action),
                                  _ptr(_bytes(done)), R, N, _stream(mask)), "eamrl_tsp_step")


def cvrp_mask_(visited, used, vcap, demand, cur, mask):
    lib = _lib.load()
    R, M = visited.shape
    B, N = demand.shape
    if M != N + 1 or R % B:
        raise ValueError("cvrp_mask: shape mismatch")
    _chk(visited, "visited", torch.uint8)
    _chk(mask, "action_mask", torch.bool, (R, M))
    for nm, t in (("used_capacity", used), ("vehicle_capacity", vcap)):
        _chk(t, nm, torch.float32)
        if t.numel() != R:
            raise ValueError(f"{nm} must have {R} elements")
    _chk(demand, "demand", torch.float32)
    _chk(cur, "current_node", torch.int64)
    _lib.check(lib.eamrl_cvrp_mask(_ptr(visited), _ptr(used), _ptr(vcap), _ptr(demand), _ptr(cur), _ptr(_bytes(mask)),
                                   R, B, N, _stream(mask)), "eamrl_cvrp_mask")


def cvrp_step_mask_(visited, used, vcap, demand, cur, action, mask, done):
    lib = _lib.load()
    R, M = visited.shape
    B, N = demand.shape
    if M != N + 1 or R % B:
        raise ValueError("cvrp_step: shape mismatch")
    _chk(visited, "visited", torch.uint8)
    _chk(mask, "action_mask", torch.bool, (R, M))
    for nm, t in (("used_capacity", used), ("vehicle_capacity", vcap)):
        _chk(t, nm, torch.float32)
        if t.numel() != R:
            raise ValueError(f"{nm} must have {R} elements")
    _chk(demand, "demand", torch.float32)
    for nm, t in (("current_node", cur), ("action", action)):
        _chk(t, nm, torch.int64)
        if t.numel() != R:
            raise ValueError(f"{nm} must have {R} elements")
    _chk(done, "done", torch.bool)
    _lib.check(lib.eamrl_cvrp_step_mask(_ptr(visited), _ptr(used), _ptr(vcap), _ptr(demand), _ptr(cur), _ptr(action),
                                        _ptr(_bytes(mask)), _ptr(_bytes(done)), R, B, N, _stream(mask)),
               "eamrl_cvrp_step_mask")


# ------------------------------------------------------------------------------------------------------
# reward
# ------------------------------------------------------------------------------------------------------
def sdvrp_step_mask_(rem, used, vcap, cur, action, mask, done=None):
    """SDVRPEnv._step + get_action_mask in place (sdvrp/env.py:58-92,137-146); action None: mask only."""
    lib = _lib.load()
    R, M = rem.shape
    _chk(rem, "demand_with_depot", torch.float32)
    _chk(used, "used_capacity", torch.float32, (R,))
    _chk(vcap, "vehicle_capacity", torch.float32, (R,))
    _chk(cur, "current_node", torch.int64, (R,))
    _chk(mask, "action_mask", torch.bool, (R, M))
    if action is not None:
        _chk(action, "action", torch.int64, (R,))
        _chk(done, "done", torch.bool, (R,))
    _lib.check(lib.eamrl_sdvrp_step_mask(_ptr(rem), _ptr(used), _ptr(vcap), _ptr(cur), _ptr(action), _ptr(_bytes(mask)),
                                         _ptr(_bytes(done)) if done is not None else None, R, M, _stream(mask)),
               "eamrl_sdvrp_step_mask")
    return mask


def pctsp_step_mask_(visited, prize_tot, pen_tot, prize, penalty, cur, istep, action, mask, done=None):
    """PCTSPEnv._step + get_action_mask in place (pctsp/env.py:64-97,156-163); action None: mask only."""
    lib = _lib.load()
    R, M = visited.shape
    B = R if prize is None else prize.shape[0]
    if visited.dtype not in (torch.bool, torch.uint8):
        raise TypeError("visited must be bool or uint8")
    _chk(_bytes(visited), "visited", torch.uint8, (R, M))
    _chk(prize_tot, "cur_total_prize", torch.float32, (R,))
    _chk(mask, "action_mask", torch.bool, (R, M))
    if action is not None:
        _chk(prize, "real_prize", torch.float32, (B, M))
        _chk(cur, "current_node", torch.int64, (R,))
        _chk(istep, "i", torch.int64, (R,))
        _chk(action, "action", torch.int64, (R,))
        _chk(done, "done", torch.bool, (R,))
        if pen_tot is not None:
            _chk(pen_tot, "cur_total_penalty", torch.float32, (R,))
            _chk(penalty, "penalty", torch.float32, (B, M))
    _lib.check(lib.eamrl_pctsp_step_mask(_ptr(_bytes(visited)), _ptr(prize_tot), _ptr(pen_tot), _ptr(prize), _ptr(penalty),
                                         _ptr(cur), _ptr(istep), _ptr(action), _ptr(_bytes(mask)),
                                         _ptr(_bytes(done)) if done is not None else None, R, B, M, _stream(mask)),
               "eamrl_pctsp_step_mask")
    return mask


def cvrptw_step_mask_(visited, used, vcap, demand, cur, time, locs, tw, dur, action, mask, done=None):
    """CVRPTWEnv._step + get_action_mask in place (cvrptw/env.py:103-138); action None: mask only.
    tw [B, M, 2] and dur [B, M] as float32."""
    lib = _lib.load()
    R, M = visited.shape
    B, N = demand.shape
    if N + 1 != M or R % B:
        raise ValueError("cvrptw_step: shape mismatch")
    _chk(visited, "visited", torch.uint8)
    _chk(used, "used_capacity", torch.float32, (R,))
    _chk(vcap, "vehicle_capacity", torch.float32, (R,))
    _chk(demand, "demand", torch.float32)
    _chk(cur, "current_node", torch.int64, (R,))
    _chk(time, "current_time", torch.float32, (R,))
    _chk(locs, "locs", torch.float32, (B, M, 2))
    _chk(tw, "time_windows", torch.float32, (B, M, 2))
    _chk(mask, "action_mask", torch.bool, (R, M))
    if action is not None:
        _chk(dur, "durations", torch.float32, (B, M))
        _chk(action, "action", torch.int64, (R,))
        _chk(done, "done", torch.bool, (R,))
    _lib.check(lib.eamrl_cvrptw_step_mask(_ptr(visited), _ptr(used), _ptr(vcap), _ptr(demand), _ptr(cur), _ptr(time),
                                          _ptr(locs), _ptr(tw), _ptr(dur), _ptr(action), _ptr(_bytes(mask)),
                                          _ptr(_bytes(done)) if done is not None else None, R, B, N, _stream(mask)),
               "eamrl_cvrptw_step_mask")
    return mask


def cvrptw_check_time(actions, locs, tw, dur):
    """-> device int32[2]: [0] = rows that start a service after its window closed (cvrptw/env.py:203-227)."""
    lib = _lib.load()
    _chk(actions, "actions", torch.int64)
    _chk(locs, "locs", torch.float32)
    B, M, _ = locs.shape
    _chk(tw, "time_windows", torch.float32, (B, M, 2))
    _chk(dur, "durations", torch.float32, (B, M))
    R, T = actions.shape
    bad = torch.zeros(2, device=actions.device, dtype=torch.int32)
    _lib.check(lib.eamrl_cvrptw_check_time(_ptr(actions), _ptr(locs), _ptr(tw), _ptr(dur), R, B, M, T, _ptr(bad),
                                           _stream(actions)), "eamrl_cvrptw_check_time")
    return bad


def op_step_mask_(visited, tour_len, prize_tot, prize, locs, maxlen, cur, istep, action, mask, done=None):
    """OPEnv._step + get_action_mask in place (op/env.py:69-102,149-165); action None: mask only."""
    lib = _lib.load()
    R, M = visited.shape
    _chk(locs, "locs", torch.float32)
    B = locs.shape[0]
    if tuple(locs.shape) != (B, M, 2) or R % B:
        raise ValueError("op_step: locs must be [B, M, 2] with R a multiple of B")
    _chk(_bytes(visited), "visited", torch.uint8, (R, M))
    _chk(tour_len, "tour_length", torch.float32, (R,))
    _chk(maxlen, "max_length", torch.float32, (B, M))
    _chk(cur, "current_node", torch.int64, (R,))
    _chk(mask, "action_mask", torch.bool, (R, M))
    if action is not None:
        _chk(istep, "i", torch.int64, (R,))
        _chk(action, "action", torch.int64, (R,))
        _chk(done, "done", torch.bool, (R,))
        if prize_tot is not None:
            _chk(prize_tot, "current_total_prize", torch.float32, (R,))
            _chk(prize, "prize", torch.float32, (B, M))
    _lib.check(lib.eamrl_op_step_mask(_ptr(_bytes(visited)), _ptr(tour_len), _ptr(prize_tot), _ptr(prize), _ptr(locs),
                                      _ptr(maxlen), _ptr(cur), _ptr(istep), _ptr(action), _ptr(_bytes(mask)),
                                      _ptr(_bytes(done)) if done is not None else None, R, B, M, _stream(mask)),
               "eamrl_op_step_mask")
    return mask


def op_reward(prize, actions):
    """OPEnv._get_reward (op/env.py:167-177): the collected prize."""
    lib = _lib.load()
    _chk(prize, "prize", torch.float32)
    B, M = prize.shape
    _chk(actions, "actions", torch.int64)
    R, T = actions.shape
    if R % B:
        raise ValueError("actions rows must be a multiple of the number of instances")
    out = torch.empty(R, dtype=torch.float32, device=prize.device)
    _lib.check(lib.eamrl_op_reward(_ptr(prize), _ptr(actions), _ptr(out), R, B, M, T, _stream(prize)), "eamrl_op_reward")
    return out


def op_check_solution(actions, locs, maxlen):
    """-> device int32[2]: (rows with a customer visited twice, rows longer than allowed)  (op/env.py:179-212)."""
    lib = _lib.load()
    _chk(actions, "actions", torch.int64)
    _chk(locs, "locs", torch.float32)
    B, M, _ = locs.shape
    _chk(maxlen, "max_length", torch.float32, (B, M))
    R, T = actions.shape
    bad = torch.zeros(2, device=actions.device, dtype=torch.int32)
    _lib.check(lib.eamrl_op_check_solution(_ptr(actions), _ptr(locs), _ptr(maxlen), R, B, M, T, _ptr(bad),
                                           _stream(actions)), "eamrl_op_check_solution")
    return bad


def pctsp_reward(locs, penalty, actions):
    """PCTSPEnv._get_reward (pctsp/env.py:165-187): saved penalties - (tour length + all penalties)."""
    lib = _lib.load()
    _chk(locs, "locs", torch.float32)
    B, M, _ = locs.shape
    _chk(penalty, "penalty", torch.float32, (B, M))
    _chk(actions, "actions", torch.int64)
    R, T = actions.shape
    if R % B:
        raise ValueError("actions rows must be a multiple of the number of instances")
    out = torch.empty(R, dtype=torch.float32, device=locs.device)
    _lib.check(lib.eamrl_pctsp_reward(_ptr(locs), _ptr(penalty), _ptr(actions), _ptr(out), R, B, M, T, _stream(locs)),
               "eamrl_pctsp_reward")
    return out


def tour_length_reward(locs, actions, with_depot):
    lib = _lib.load()
    _chk(locs, "locs", torch.float32)
    _chk(actions, "actions", torch.int64)
    B, M, two = locs.shape
    R, T = actions.shape
    if two != 2 or R % B:
        raise ValueError("tour_length: shape mismatch")
    out = torch.empty(R, device=locs.device, dtype=torch.float32)
    _lib.check(lib.eamrl_tour_length(_ptr(locs), _ptr(actions), _ptr(out), R, B, M, T, int(with_depot), _stream(locs)),
               "eamrl_tour_length")
    return out


def sum_logp(logp):
    lib = _lib.load()
    _need_gpu(logp, "logp")
    if logp.dtype != torch.float32 or logp.dim() != 2 or logp.stride(1) != 1:
        raise ValueError("sum_logp: [R, T] fp32 with unit inner stride required")
    R, T = logp.shape
    out = torch.empty(R, device=logp.device, dtype=torch.float32)
    _lib.check(lib.eamrl_sum_logp(_ptr(logp), logp.stride(0), _ptr(out), R, T, _stream(logp)), "eamrl_sum_logp")
    return out


MULTI_COPY_MAX = 16


def multi_copy_(pairs):
    """[(dst, src | None), ...] -> dst.copy_(src) / dst.zero_() for all of them in ONE launch per 16 segments
    (eamrl_multi_copy).  Tensors must be contiguous, on the same device, src of dst's dtype and size."""
    lib = _lib.load()
    segs = []
    for dst, src in pairs:
        _need_gpu(dst, "multi_copy dst")
        if not dst.is_contiguous():
            raise ValueError("multi_copy: dst must be contiguous")
        if src is not None:
            if src.dtype != dst.dtype or src.numel() != dst.numel() or not src.is_contiguous() or src.device != dst.device:
                raise ValueError("multi_copy: src must be a contiguous tensor of dst's dtype, size and device")
        if dst.numel():
            segs.append((dst, src))
    for i in range(0, len(segs), MULTI_COPY_MAX):
        part = segs[i:i + MULTI_COPY_MAX]
        n = len(part)
        srcs = (C.c_void_p * n)(*[None if s_ is None else s_.data_ptr() for _, s_ in part])
        dsts = (C.c_void_p * n)(*[d.data_ptr() for d, _ in part])
        nbytes = (C.c_int64 * n)(*[d.numel() * d.element_size() for d, _ in part])
        _lib.check(lib.eamrl_multi_copy(n, srcs, dsts, nbytes, _stream(part[0][0])), "eamrl_multi_copy")


def rollout_finish(env_name, locs, actions, logp=None, demand=None, vcap=None, want_reward=True, bad=None):
    """Reward, summed log-likelihood and validity counters of TSP / CVRP tours in one launch (eamrl_rollout_finish):
    bit-identical to tour_length_reward / sum_logp / check_solution.  bad: int32[2] device tensor to add to, or None.
    -> (reward [R] | None, ll [R] | None)."""
    lib = _lib.load()
    _chk(locs, "locs", torch.float32)
    _chk(actions, "actions", torch.int64)
    B, M, _ = locs.shape
    R, T = actions.shape
    if R % B:
        raise ValueError("rollout_finish: rows must be a multiple of the number of instances")
    reward = torch.empty(R, device=locs.device, dtype=torch.float32) if want_reward else None
    ll, ld = None, 0
    if logp is not None:
        _need_gpu(logp, "logp")
        if logp.dtype != torch.float32 or logp.shape != actions.shape or logp.stride(1) != 1:
            raise ValueError("rollout_finish: logp must be [R, T] fp32 with unit inner stride")
        ll, ld = torch.empty(R, device=locs.device, dtype=torch.float32), logp.stride(0)
    vc = None
    if env_name == "cvrp" and bad is not None:
        _chk(demand, "demand", torch.float32, (B, M - 1))
        vc = vcap.reshape(-1).contiguous()
        if vc.numel() != R:
            vc = vc.repeat(R // vc.numel())
        _chk(vc, "vehicle_capacity", torch.float32, (R,))
    _lib.check(lib.eamrl_rollout_finish(ENVS[env_name], _ptr(locs), _ptr(actions), _ptr(logp), ld, _ptr(demand), _ptr(vc),
                                        _ptr(reward), _ptr(ll), _ptr(bad), R, B, M, T, _stream(locs)), "eamrl_rollout_finish")
    return reward, ll


def check_solution(env_name, actions, demand=None, vcap=None, num_loc=None):
    """-> device int32[2]: (invalid tours, over-capacity rows); sdvrp: (rows with demand left, double depot visits)."""
    lib = _lib.load()
    _chk(actions, "actions", torch.int64)
    R, T = actions.shape
    bad = torch.zeros(2, device=actions.device, dtype=torch.int32)
    if env_name == "tsp":
        N, B = (T if num_loc is None else num_loc), R
        _lib.check(lib.eamrl_check_solution(ENV_TSP, _ptr(actions), None, None, R, B, N, T, _ptr(bad), _stream(actions)),
                   "eamrl_check_solution")
    elif env_name == "pctsp":       # demand = real_prize [B, N+1]
        _chk(demand, "real_prize", torch.float32)
        B, M = demand.shape
        _lib.check(lib.eamrl_check_solution(ENV_PCTSP, _ptr(actions), _ptr(demand), None, R, B, M - 1, T, _ptr(bad),
                                            _stream(actions)), "eamrl_check_solution")
    else:
        _chk(demand, "demand", torch.float32)
        B, N = demand.shape
        vc = vcap.reshape(-1).contiguous()
        if vc.numel() != R:
            vc = vc.repeat(R // vc.numel())
        _chk(vc, "vehicle_capacity", torch.float32, (R,))
        _lib.check(lib.eamrl_check_solution(ENVS[env_name], _ptr(actions), _ptr(demand), _ptr(vc), R, B, N, T, _ptr(bad),
                                            _stream(actions)), "eamrl_check_solution")
    return bad


# ------------------------------------------------------------------------------------------------------
# decode
# ------------------------------------------------------------------------------------------------------
def beam_topk(logprobs, parent, B: int, beam_width: int):
    """Per instance the beam_width best (beam, node) candidates of a beam-search step.
    -> node [R] i64, beam [R] i32, cum [R] f32, step_logp [R] f32 with R = beam_width * B (rows "(w b)")."""
    lib = _lib.load()
    R, M = logprobs.shape
    if R != beam_width * B:
        raise ValueError("beam_topk: rows must be beam_width * B")
    _chk(logprobs, "logprobs", torch.float32)
    _chk(parent, "parent", torch.float32, (R,))
    dev = logprobs.device
    node = torch.empty(R, dtype=torch.int64, device=dev)
    beam = torch.empty(R, dtype=torch.int32, device=dev)
    cum = torch.empty(R, dtype=torch.float32, device=dev)
    slp = torch.empty(R, dtype=torch.float32, device=dev)
    _lib.check(lib.eamrl_beam_topk(_ptr(logprobs), _ptr(parent), B, beam_width, M, _ptr(node), _ptr(beam), _ptr(cum),
                                   _ptr(slp), _stream(logprobs)), "eamrl_beam_topk")
    return node, beam, cum, slp


def ea_num_pairs(selection_rate: float, pop_size: int) -> int:
    """Crossover pairs per generation of EA.run (elites // 2; int(rate*S) elites, 0 -> all, S <= 2 -> all)."""
    if pop_size <= 2:
        ne = pop_size
    else:
        ne = int(selection_rate * pop_size)
        ne = pop_size if ne <= 0 else min(ne, pop_size)
    return ne // 2


def ea_tsp_run_(locs, pop, num_generations, mutation_rate, crossover_rate, selection_rate, cross_rand, cross_idx,
                mut_rand, mut_idx):
    """In place on pop [B, S, N] (int64); returns fitness [B, S].  Draw tensors: see include/eamrl.h."""
    lib = _lib.load()
    _chk(locs, "locs", torch.float32)
    _chk(pop, "pop", torch.int64)
    B, S, N = pop.shape
    if locs.shape != (B, N, 2):
        raise ValueError(f"ea_tsp_run: locs {tuple(locs.shape)} does not match pop {tuple(pop.shape)}")
    P = ea_num_pairs(selection_rate, S)
    G = int(num_generations)
    if G > 0 and P > 0:
        _chk(cross_rand, "cross_rand", torch.float64, (G, B, P))
        _chk(cross_idx, "cross_idx", torch.int32, (G, B, P, 2))
        _chk(mut_rand, "mut_rand", torch.float64, (G, B, 2 * P))
        _chk(mut_idx, "mut_idx", torch.int32, (G, B, 2 * P, 2))
    fitness = torch.empty(B, S, device=pop.device, dtype=torch.float32)
    nul = lambda t: _ptr(t) if (G > 0 and P > 0) else None
    _lib.check(lib.eamrl_ea_tsp_run(_ptr(locs), _ptr(pop), _ptr(fitness), B, S, N, G, float(mutation_rate),
                                    float(crossover_rate), float(selection_rate), nul(cross_rand), nul(cross_idx),
                                    nul(mut_rand), nul(mut_idx), _stream(pop)), "eamrl_ea_tsp_run")
    return fitness


def ea_cvrp_run_(locs, demand, vcap, pop, num_generations, mutation_rate, crossover_rate, selection_rate, top_k,
                 init_mut_rand, init_mut_u, cross_rand, cross_u, mut_rand, mut_u):
    """In place on pop [B, S, L] (int64 action rows, 0 = depot); returns fitness [B, S].  See include/eamrl.h."""
    lib = _lib.load()
    _chk(pop, "pop", torch.int64)
    B, S, L = pop.shape
    N = demand.shape[-1]
    _chk(locs, "locs", torch.float32, (B, N + 1, 2))
    _chk(demand, "demand", torch.float32, (B, N))
    _chk(vcap, "vcap", torch.float32, (B,))
    P = ea_num_pairs(selection_rate, S)
    G = int(num_generations)
    _chk(init_mut_rand, "init_mut_rand", torch.float64, (B, S))
    _chk(init_mut_u, "init_mut_u", torch.float64, (B, S, 3))
    if G > 0 and P > 0:
        _chk(cross_rand, "cross_rand", torch.float64, (G, B, P))
        _chk(cross_u, "cross_u", torch.float64, (G, B, P))
        _chk(mut_rand, "mut_rand", torch.float64, (G, B, 2 * P))
        _chk(mut_u, "mut_u", torch.float64, (G, B, 2 * P, 3))
    fitness = torch.empty(B, S, device=pop.device, dtype=torch.float32)
    nul = lambda t: _ptr(t) if (G > 0 and P > 0) else None
    _lib.check(lib.eamrl_ea_cvrp_run(_ptr(locs), _ptr(demand), _ptr(vcap), _ptr(pop), _ptr(fitness), B, S, N, L, G,
                                     float(mutation_rate), float(crossover_rate), float(selection_rate), int(bool(top_k)),
                                     _ptr(init_mut_rand), _ptr(init_mut_u), nul(cross_rand), nul(cross_u),
                                     nul(mut_rand), nul(mut_u), _stream(pop)), "eamrl_ea_cvrp_run")
    return fitness


def ea_prize_run_(env_name, locs, prize, aux, pop, num_generations, mutation_rate, crossover_rate, selection_rate, top_k,
                  init_mut_rand, init_mut_u, cross_rand, cross_u, mut_rand, mut_u):
    """PCTSP / OP EA.run, in place on pop [B, S, L] (int64 action rows, 0 = depot); returns fitness [B, S].
    prize / aux [B, N+1] with the depot first (aux: PCTSP penalty, OP max_length).  See include/eamrl.h."""
    lib = _lib.load()
    if env_name not in ("pctsp", "op"):
        raise ValueError("ea_prize_run: env must be pctsp or op")
    _chk(pop, "pop", torch.int64)
    B, S, L = pop.shape
    M = locs.shape[1]
    _chk(locs, "locs", torch.float32, (B, M, 2))
    _chk(prize, "prize", torch.float32, (B, M))
    _chk(aux, "penalty / max_length", torch.float32, (B, M))
    P = ea_num_pairs(selection_rate, S)
    G = int(num_generations)
    _chk(init_mut_rand, "init_mut_rand", torch.float64, (B, S))
    _chk(init_mut_u, "init_mut_u", torch.float64, (B, S, 2))
    if G > 0 and P > 0:
        _chk(cross_rand, "cross_rand", torch.float64, (G, B, P))
        if env_name == "op":
            _chk(cross_u, "cross_u", torch.float64, (G, B, P))
        _chk(mut_rand, "mut_rand", torch.float64, (G, B, 2 * P))
        _chk(mut_u, "mut_u", torch.float64, (G, B, 2 * P, 2))
    fitness = torch.empty(B, S, device=pop.device, dtype=torch.float32)
    nul = lambda t: _ptr(t) if (G > 0 and P > 0 and t is not None) else None
    _lib.check(lib.eamrl_ea_prize_run(ENVS[env_name], _ptr(locs), _ptr(prize), _ptr(aux), _ptr(pop), _ptr(fitness), B, S,
                                      M - 1, L, G, float(mutation_rate), float(crossover_rate), float(selection_rate),
                                      int(bool(top_k)), _ptr(init_mut_rand), _ptr(init_mut_u), nul(cross_rand),
                                      nul(cross_u), nul(mut_rand), nul(mut_u), _stream(pop)), "eamrl_ea_prize_run")
    return fitness


class DecodeCache:
    """Device-resident decoder cache (struct eamrl_cache).

    Two layouts of the same E-wide per-node rows (the kernels take base pointers and a row stride, so both work):
      * slot-major, one [B, M, ld] fp32 buffer with the rows of a node side by side -- graphs up to 128 nodes, where
        the register-resident kernel reads everything once:
            TSP : K | V | L | P_first | P_current | Lp        (ld = 6E)
            CVRP: K | V | L | P_current | Lp                  (ld = 5E)
      * plane-major, [P, B, M, E] (ld = E) -- larger graphs, where the streaming kernel re-reads one kind of row per
        stage and step: a stage then walks one dense plane instead of 512 bytes out of every 2.5-3 KB.
    K, V, L are AttentionModelDecoder's glimpse_key / glimpse_val / logit_key
    (zoo/am/decoder.py:206-235); P_* and Lp are the weight folds described in DESIGN.md.
    """

    def __init__(self, env_name, buf, cvec, gctx, embeddings, num_heads, dyn=None, embed_dim=None):
        self.env_name, self.buf, self.cvec, self.gctx = env_name, buf, cvec, gctx
        self.dyn = dyn          # sdvrp: [3, E] dynamic-embedding vectors (key, value, folded logit key)
        self.node_embeddings = embeddings      # None when the fused encoder kept them in LDS (nobody asked for them)
        self.E = embeddings.shape[-1] if embeddings is not None else int(embed_dim)
        self.H = num_heads
        self.slots = slot_map(env_name)
        self.planes = buf.dim() == 4
        if self.planes:
            P, self.B, self.M, self.ld = buf.shape
            assert P == len(self.slots) and self.ld == self.E
        else:
            self.B, self.M, self.ld = buf.shape
            assert self.ld == len(self.slots) * self.E

    def view(self, name):
        i = self.slots[name]
        return self.buf[i] if self.planes else self.buf[..., i * self.E:(i + 1) * self.E]

    # the reference's vocabulary (PrecomputedCache fields, zoo/am/decoder.py:22-41)
    @property
    def glimpse_key(self):
        return self.view("K")

    @property
    def glimpse_val(self):
        return self.view("V")

    @property
    def logit_key(self):
        return self.view("L")

    @property
    def graph_context(self):
        return self.gctx if self.gctx is not None else 0

    def struct(self):
        c = _lib.Cache()
        base, step = self.buf.data_ptr(), (self.B * self.M * self.E * 4 if self.planes else self.E * 4)
        c.K = C.c_void_p(base + self.slots["K"] * step)
        c.V = C.c_void_p(base + self.slots["V"] * step)
        c.Lp = C.c_void_p(base + self.slots["Lp"] * step)
        c.Pa = C.c_void_p(base + self.slots["Pa"] * step)
        c.Pb = C.c_void_p(base + self.slots["Pb"] * step) if "Pb" in self.slots else None
        c.cvec, c.gctx = _ptr(self.cvec), _ptr(self.gctx)
        c.ld, c.B, c.M, c.E, c.H = self.ld, self.B, self.M, self.E, self.H
        c.dyn = _ptr(self.dyn)
        return c


def slot_map(env_name):
    names = ["K", "V", "L", "Pa", "Pb", "Lp"] if env_name == "tsp" else ["K", "V", "L", "Pa", "Lp"]
    return {n: i for i, n in enumerate(names)}


class RolloutState:
    """Flat per-row state tensors (struct eamrl_state) for R = S*B rows."""

    def __init__(self, env_name, R, M, device, demand=None):
        self.env_name, self.R, self.M = env_name, R, M
        i64 = dict(dtype=torch.int64, device=device)
        self.first = torch.zeros(R, **i64)
        self.cur = torch.zeros(R, **i64)
        self.istep = torch.zeros(R, **i64)
        self.done = torch.zeros(R, dtype=torch.bool, device=device)
        self.mask = torch.ones(R, M, dtype=torch.bool, device=device)
        self.used = self.vcap = self.visited = self.rem = None
        self.demand = demand
        self.locs = None                                                       # op, cvrptw: node coordinates [B, M, 2]
        self.time = self.tw = self.dur = None                                  # cvrptw: clock [R], windows, service times
        if env_name == "cvrptw":
            self.time = torch.zeros(R, dtype=torch.float32, device=device)
        if env_name in ("cvrp", "sdvrp", "pctsp", "op", "cvrptw"):
            self.used = torch.zeros(R, dtype=torch.float32, device=device)     # pctsp: cur_total_prize; op: tour_length
            self.vcap = torch.ones(R, dtype=torch.float32, device=device)      # pctsp: prize_required; op: max_length[:, 0]
        if env_name in ("cvrp", "pctsp", "op", "cvrptw"):
            self.visited = torch.zeros(R, M, dtype=torch.uint8, device=device)
        if env_name == "sdvrp":
            self.rem = torch.zeros(R, M, dtype=torch.float32, device=device)   # demand_with_depot

    def reorder_(self, idx):
        """Rows taken from rows `idx` (beam search: every beam continues the state of its parent beam).  vcap is
        per instance and the row order keeps r % B, so it needs no reordering, nor does demand."""
        for name in ("first", "cur", "istep", "done", "mask", "used", "visited", "rem", "time"):
            v = getattr(self, name)
            if v is not None:
                setattr(self, name, v.index_select(0, idx).contiguous())

    def struct(self):
        s = _lib.State()
        s.first, s.cur, s.istep = _ptr(self.first), _ptr(self.cur), _ptr(self.istep)
        s.used, s.vcap, s.demand = _ptr(self.used), _ptr(self.vcap), _ptr(self.demand)
        s.mask, s.done = _ptr(_bytes(self.mask)), _ptr(_bytes(self.done))
        s.visited = _ptr(None if self.visited is None else _bytes(self.visited))
        s.rem = _ptr(getattr(self, "rem", None))
        s.locs = _ptr(getattr(self, "locs", None))
        s.time, s.tw, s.dur = (_ptr(getattr(self, k, None)) for k in ("time", "tw", "dur"))
        s.heads_out = _ptr(getattr(self, "heads_out", None))       # [R, t_max, E] or None (eamrl_state.heads_out)
        return s


def _validate_state(st: RolloutState, cache: DecodeCache):
    R, M = st.R, st.M
    if M != cache.M or R % cache.B:
        raise ValueError("decode: state / cache shape mismatch")
    _chk(st.mask, "action_mask", torch.bool, (R, M))
    _chk(st.cur, "current_node", torch.int64, (R,))
    _chk(st.done, "done", torch.bool, (R,))
    if st.env_name == "tsp":
        _chk(st.first, "first_node", torch.int64, (R,))
        _chk(st.istep, "i", torch.int64, (R,))
    else:
        _chk(st.used, "used_capacity", torch.float32, (R,))
        _chk(st.vcap, "vehicle_capacity", torch.float32, (R,))
        if st.env_name == "sdvrp":
            _chk(st.rem, "demand_with_depot", torch.float32, (R, M))
            _chk(cache.dyn, "dynamic embedding vectors", torch.float32, (3, cache.E))
        elif st.env_name == "pctsp":
            _chk(_bytes(st.visited), "visited", torch.uint8, (R, M))
            _chk(st.demand, "real_prize", torch.float32, (cache.B, M))
            _chk(st.istep, "i", torch.int64, (R,))
        elif st.env_name == "cvrptw":
            _chk(st.visited, "visited", torch.uint8, (R, M))
            _chk(st.demand, "demand", torch.float32, (cache.B, M - 1))
            _chk(st.time, "current_time", torch.float32, (R,))
            _chk(st.locs, "locs", torch.float32, (cache.B, M, 2))
            _chk(st.tw, "time_windows", torch.float32, (cache.B, M, 2))
            _chk(st.dur, "durations", torch.float32, (cache.B, M))
            _chk(cache.cvec, "context state columns", torch.float32, (2 * cache.E,))
        elif st.env_name == "op":
            _chk(_bytes(st.visited), "visited", torch.uint8, (R, M))
            _chk(st.demand, "max_length", torch.float32, (cache.B, M))
            _chk(st.locs, "locs", torch.float32, (cache.B, M, 2))
            _chk(st.istep, "i", torch.int64, (R,))
        else:
            _chk(st.visited, "visited", torch.uint8, (R, M))
            _chk(st.demand, "demand", torch.float32, (cache.B, M - 1))


def decode_step(st: RolloutState, cache: DecodeCache, mode="greedy", noise=None, given=None, clip=10.0, temp=1.0,
                fuse_env_step=False, want_logprobs=False, want_logits=False, status=None, top_k=0, top_p=0.0):
    """One decode step for all rows.  -> (action [R], logp [R], logprobs [R,M] | None, logits [R,M] | None)."""
    lib = _lib.load()
    _validate_state(st, cache)
    R, M, dev = st.R, st.M, st.mask.device
    action = torch.empty(R, dtype=torch.int64, device=dev)
    logp = torch.empty(R, dtype=torch.float32, device=dev)
    lps = torch.empty(R, M, dtype=torch.float32, device=dev) if want_logprobs else None
    lgs = torch.empty(R, M, dtype=torch.float32, device=dev) if want_logits else None
    if status is None:
        status = torch.zeros(1, dtype=torch.int32, device=dev)
    if noise is not None:
        _chk(noise, "noise", torch.float32, (R, M))
    if given is not None:
        _chk(given, "given actions", torch.int64, (R,))
    cs, ss = cache.struct(), st.struct()
    _lib.check(lib.eamrl_am_decode_step(ENVS[st.env_name], C.byref(cs), C.byref(ss), R, MODES[mode], _ptr(noise),
                                        _ptr(given), float(clip), float(temp), int(top_k), float(top_p),
                                        int(fuse_env_step), _ptr(action),
                                        _ptr(logp), _ptr(lps), _ptr(lgs), _ptr(status), _stream(st.mask)),
               "eamrl_am_decode_step")
    return action, logp, lps, lgs, status


def exp1_noise(seed: int, R: int, T: int, M: int, device="cuda", seed_dev=None):
    """[R, T, M] Exp(1) draws of the library's counter-based generator (eamrl_exp1_noise): what a seeded rollout uses.
    seed_dev: optional int64 device tensor [1] XOR-ed into the seed on the device."""
    lib = _lib.load()
    out = torch.empty(R, T, M, dtype=torch.float32, device=device)
    _need_gpu(out, "noise")
    _lib.check(lib.eamrl_exp1_noise(C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), _ptr(seed_dev), _ptr(out), R, T, M,
                                    _stream(out)), "eamrl_exp1_noise")
    return out


def _rollout_outputs(R, t_max, dev):
    """actions [R, t_max] i64, logps [R, t_max] f32 and flags int32[4] = (steps, status, bad0, bad1) as views of ONE
    zero-filled buffer (one fill launch instead of three)."""
    n = R * t_max
    buf = torch.zeros(n * 12 + 16, dtype=torch.uint8, device=dev)
    actions = buf[:n * 8].view(torch.int64).view(R, t_max)
    logps = buf[n * 8:n * 12].view(torch.float32).view(R, t_max)
    flags = buf[n * 12:].view(torch.int32)
    return actions, logps, flags


HEADS_CAPTURE_MAX_BYTES = int(os.environ.get("EAMRL_HEADS_CAPTURE_MAX_BYTES", 48 << 30))   # 5 - 11 GB at the POMO sizes
_heads_skip_logged = False


def _capture_heads(st, cache, cs, R, t_max, want_heads):
    """Training: a [R, t_max, E] buffer for the steps' glimpse outputs where the kernel chosen for this shape writes it
    (eamrl_state.heads_out; st.heads afterwards, None otherwise).  The backward allocates a second buffer of the same size
    (dheads), so the capture is taken only while twice its size fits into half of the free device memory and under
    EAMRL_HEADS_CAPTURE_MAX_BYTES; otherwise -- and if the allocation fails -- the backward recomputes the glimpse (logged
    once: it is a slower path, not an error)."""
    global _heads_skip_logged
    st.heads = st.heads_out = None
    if not want_heads or st.env_name == "sdvrp" or not _lib.load().eamrl_rollout_rng_native(ENVS[st.env_name], C.byref(cs), R):
        return          # (SDVRP: its re-evaluation recomputes the glimpse -- the dynamic embedding's terms are not in the capture)
    need = R * int(t_max) * cache.E * 4
    free = torch.cuda.mem_get_info(st.mask.device)[0] + torch.cuda.memory_reserved(st.mask.device) - torch.cuda.memory_allocated(st.mask.device)
    reason = None
    if need > HEADS_CAPTURE_MAX_BYTES:
        reason = f"{need / 2**30:.1f} GiB > EAMRL_HEADS_CAPTURE_MAX_BYTES"
    elif 2 * need > free // 2:
        reason = f"2 x {need / 2**30:.1f} GiB would take more than half of the {free / 2**30:.1f} GiB free"
    else:
        try:
            st.heads = st.heads_out = torch.empty(R, int(t_max), cache.E, dtype=torch.float32, device=st.mask.device)
        except torch.cuda.OutOfMemoryError:
            reason = f"allocation of {need / 2**30:.1f} GiB failed"
    if reason and not _heads_skip_logged:
        _heads_skip_logged = True
        logging.getLogger(__name__).warning(
            "rollout: glimpse outputs of the decode steps are not kept for the backward (%s); the re-evaluation recomputes "
            "them (logits backward 18 instead of 11 ms at 1024 x 100 x 100)", reason)


def rollout(st: RolloutState, cache: DecodeCache, mode="greedy", noise=None, given=None, clip=10.0, temp=1.0,
            t_max=None, top_k=0, top_p=0.0, seed=None, seed_dev=None, return_flags=False, want_heads=False):
    """Whole decode loop in one launch.  -> (actions [R,t_max], logps [R,t_max], info int32[2] = (steps, status)).
    Sampling takes its Exp(1) noise from `noise` [R, T, M] or, with `seed` (XOR the device word `seed_dev`), from the counter-based
    generator -- in place where the kernel supports it (TSP multistart), else through a scratch tensor of the same draws.
    return_flags: info is int32[4] = (steps, status, 0, 0), the last two free for the caller's validity counters."""
    lib = _lib.load()
    _validate_state(st, cache)
    R, M, dev = st.R, st.M, st.mask.device
    if t_max is None:
        t_max = {"tsp": M, "cvrp": 2 * M + 1, "sdvrp": 3 * M + 1, "pctsp": M + 1, "op": M + 1, "cvrptw": 2 * M + 1}[st.env_name]
    if mode == "sampling" and noise is None:
        if seed is None:
            raise ValueError("rollout: sampling needs `noise` or `seed`")
        if top_k or (0.0 < top_p < 1.0):
            noise = exp1_noise(seed, R, int(t_max), M, dev, seed_dev)      # the filtering (streaming) kernel reads a tensor
        else:
            actions, logps, info = _rollout_outputs(R, int(t_max), dev)
            cs = cache.struct()
            _capture_heads(st, cache, cs, R, t_max, want_heads)
            ss = st.struct()
            st.heads_out = None
            native = lib.eamrl_rollout_rng_native(ENVS[st.env_name], C.byref(cs), R)
            scratch = None if native else torch.empty(R, t_max, M, dtype=torch.float32, device=dev)
            _lib.check(lib.eamrl_am_rollout_seeded(ENVS[st.env_name], C.byref(cs), C.byref(ss), R,
                                                   C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), _ptr(seed_dev), _ptr(scratch),
                                                   float(clip),
                                                   float(temp), int(t_max), _ptr(actions), _ptr(logps),
                                                   C.c_void_p(info.data_ptr()), C.c_void_p(info.data_ptr() + 4),
                                                   _stream(st.mask)), "eamrl_am_rollout_seeded")
            return actions, logps, (info if return_flags else info[:2])
    t_given = 0
    if noise is not None:
        _chk(noise, "noise", torch.float32)
        if noise.shape[0] != R or noise.shape[2] != M or noise.shape[1] < t_max:
            if noise.shape[0] != R or noise.shape[2] != M:
                raise ValueError("noise must be [R, T, M]")
            t_max = noise.shape[1]
        elif noise.shape[1] != t_max:
            t_max = noise.shape[1]
    if given is not None:
        _chk(given, "given actions", torch.int64)
        if given.dim() != 2 or given.shape[0] != R:
            raise ValueError("given actions must be [R, T]")
        t_given = given.shape[1]
        if noise is None:
            t_max = min(t_max, t_given) if st.env_name != "tsp" else t_max
    actions, logps, info = _rollout_outputs(R, int(t_max), dev)          # info: [steps, status, -, -]
    cs = cache.struct()
    _capture_heads(st, cache, cs, R, t_max, want_heads and not (top_k or (0.0 < top_p < 1.0)))
    ss = st.struct()
    st.heads_out = None
    steps_ptr = C.c_void_p(info.data_ptr())
    status_ptr = C.c_void_p(info.data_ptr() + 4)
    _lib.check(lib.eamrl_am_rollout(ENVS[st.env_name], C.byref(cs), C.byref(ss), R, MODES[mode], _ptr(noise),
                                    _ptr(given), t_given, float(clip), float(temp), int(top_k), float(top_p), int(t_max),
                                    _ptr(actions),
                                    _ptr(logps), steps_ptr, status_ptr, _stream(st.mask)), "eamrl_am_rollout")
    return actions, logps, (info if return_flags else info[:2])


def raise_on_status(status: int):
    """The reference's in-loop asserts, checked once per rollout instead of once per step."""
    if status & ST_NAN_LOGITS:
        raise AssertionError("Logits contain NaNs")                       # nn/attention.py:303-304
    if status & ST_INFEASIBLE:
        raise AssertionError("infeasible action selected")               # utils/decoding.py:397-399
