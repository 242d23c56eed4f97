"""ctypes front-end of the CPU oracle (test infrastructure -- see eamrl_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
Arrays are numpy; weights are a {state_dict key: np.float32 array} mapping using the
reference's key names (tests/golden/state_dict_contract.json).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ENV_TSP, ENV_CVRP, ENV_SDVRP, ENV_PCTSP, ENV_OP, ENV_CVRPTW = 0, 1, 2, 3, 4, 5
GREEDY, SAMPLE, EVALUATE = 0, 1, 2
MODES = {"greedy": GREEDY, "sampling": SAMPLE, "evaluate": EVALUATE}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _cpu_has_fma() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    fl = line.split()
                    return "fma" in fl and "avx2" in fl
    except OSError:
        pass
    return False


def lib():
    global _LIB
    if _LIB is None:
        name = "liboracle.so" if _cpu_has_fma() else "liboracle_generic.so"
        path = os.path.join(_HERE, "build", name)
        try:
            build()                     # make: a no-op when the library is newer than the C source
        except (OSError, subprocess.CalledProcessError):
            if not os.path.exists(path):
                raise
        _LIB = C.CDLL(path)
        _LIB.orc_lane_tree.restype = C.c_float
        _LIB.orc_check_tsp.restype = C.c_long
        _LIB.orc_check_cvrp.restype = C.c_long
    return _LIB


def usable_cpus() -> int:
    """CPUs this process may actually use: affinity mask and cgroup quota, not the machine's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, q // int(f.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def set_threads(n: int) -> int:
    return int(lib().orc_set_threads(C.c_int(int(n))))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


# ---------------------------------------------------------------------------------------------
# primitive wrappers
# ---------------------------------------------------------------------------------------------
def math_probe(x):
    x = _f32(x).ravel()
    e, l, t = (np.empty_like(x) for _ in range(3))
    lib().orc_math_probe(_p(x), _p(e), _p(l), _p(t), C.c_long(x.size))
    return e, l, t


def lane_tree(v):
    v = _f32(v).ravel()
    return np.float32(lib().orc_lane_tree(_p(v), C.c_int(v.size)))


def linear(x, W, b=None, relu=False):
    x = _f32(x)
    W = _f32(W)
    rows = int(np.prod(x.shape[:-1]))
    y = np.empty(x.shape[:-1] + (W.shape[0],), np.float32)
    bb = None if b is None else _f32(b)
    lib().orc_linear(_p(x), _p(W), _p(bb), _p(y), C.c_long(rows), C.c_int(W.shape[1]), C.c_int(W.shape[0]),
                     C.c_int(int(relu)))
    return y


def matmul_right(x, Wt):
    x = _f32(x)
    Wt = _f32(Wt)
    rows = int(np.prod(x.shape[:-1]))
    y = np.empty(x.shape[:-1] + (Wt.shape[1],), np.float32)
    lib().orc_matmul_right(_p(x), _p(Wt), _p(y), C.c_long(rows), C.c_int(Wt.shape[0]), C.c_int(Wt.shape[1]))
    return y


def mha_encoder(qkv, H):
    qkv = _f32(qkv)
    B, N, E3 = qkv.shape
    out = np.empty((B, N, E3 // 3), np.float32)
    lib().orc_mha_encoder(_p(qkv), _p(out), C.c_long(B), C.c_int(N), C.c_int(E3 // 3), C.c_int(H))
    return out


def batchnorm_eval(x, gamma, beta, mean, var, eps=1e-5):
    x = _f32(x).copy()
    E = x.shape[-1]
    lib().orc_batchnorm_eval(_p(x), C.c_long(x.size // E), C.c_int(E), _p(_f32(gamma)), _p(_f32(beta)),
                             _p(_f32(mean)), _p(_f32(var)), C.c_float(eps))
    return x


def batchnorm_train(x, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5):
    """-> (y, batch_mean, batch_var); running_mean / running_var (float32 arrays) are updated in place."""
    x = _f32(x).copy()
    E = x.shape[-1]
    sm, sv = np.empty(E, np.float32), np.empty(E, np.float32)
    lib().orc_batchnorm_train(_p(x), C.c_long(x.size // E), C.c_int(E), _p(_f32(gamma)), _p(_f32(beta)),
                              _p(running_mean), _p(running_var), C.c_float(momentum), C.c_float(eps), _p(sm), _p(sv))
    return x, sm, sv


def instancenorm(x, gamma, beta, eps=1e-5):
    x = _f32(x).copy()
    B, N, E = x.shape
    lib().orc_instancenorm(_p(x), C.c_long(B), C.c_int(N), C.c_int(E), _p(_f32(gamma)), _p(_f32(beta)),
                           C.c_float(eps))
    return x


def pointer_attention(query, key, value, logit_key, mask, Wout, bout=None, num_heads=8, mask_inner=True):
    """query [B, L, E]; key / value / logit_key [B, M, E]; mask [B, M] or [B, L, M] (True = feasible) or None -> [B, L, M]."""
    query, key, value, logit_key = (_f32(x) for x in (query, key, value, logit_key))
    B, L, E = query.shape
    M = key.shape[1]
    out = np.empty((B, L, M), np.float32)
    mk = None if mask is None else _u8(np.asarray(mask).astype(np.uint8))
    lib().orc_pointer_attention(_p(query), _p(key), _p(value), _p(logit_key), _p(mk),
                                C.c_int(int(mask is not None and np.asarray(mask).ndim == 3)), _p(_f32(Wout)),
                                _p(None if bout is None else _f32(bout)), _p(out), C.c_long(B), C.c_int(L), C.c_int(M),
                                C.c_int(E), C.c_int(num_heads), C.c_int(int(mask_inner)))
    return out


def exp1_noise(seed, R, T, M):
    """[R, T, M] Exp(1) draws of the counter-based generator (what the kernels compute in place for (seed, row, step, node))."""
    out = np.empty((R, T, M), np.float32)
    lib().orc_exp1_noise(C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), _p(out), C.c_long(R), C.c_int(T), C.c_int(M))
    return out


def mean_nodes(emb):
    emb = _f32(emb)
    B, M, E = emb.shape
    out = np.empty((B, E), np.float32)
    lib().orc_mean_nodes(_p(emb), _p(out), C.c_long(B), C.c_int(M), C.c_int(E))
    return out


# ---------------------------------------------------------------------------------------------
# encoder + cache (AttentionModelEncoder.forward, AttentionModelDecoder._precompute_cache)
# ---------------------------------------------------------------------------------------------
def _kind(env_name):
    """SPCTSP = PCTSP whose real_prize is the stochastic one (rl4co/envs/routing/spctsp/env.py)."""
    return {"spctsp": "pctsp"}.get(env_name, env_name)


def encode(sd, env_name, locs, demand=None, num_heads=8, training=False):
    """-> (init_embeds, embeddings).  training=True: BatchNorm layers use batch statistics and update the running
    statistics inside `sd` in place (policy.train(), nn/ops.py:45-47).  locs: TSP [B,N,2]; CVRP [B,N+1,2] with depot first.
    PCTSP: `demand` is the dict of instance tensors (expected_prize [B,N], real_prize / penalty [B,N+1], prize_required);
    node features = (x, y, expected prize, penalty)  [nn/env_embeddings/init.py:227-257]."""
    pre = "encoder.init_embedding."
    env_name = _kind(env_name)
    if env_name == "tsp":
        h = linear(locs, sd[pre + "init_embed.weight"], sd[pre + "init_embed.bias"])
    else:
        depot = linear(locs[:, :1], sd[pre + "init_embed_depot.weight"], sd[pre + "init_embed_depot.bias"])
        if env_name == "pctsp":
            extra = [_f32(demand["expected_prize"])[..., None], _f32(demand["penalty"])[:, 1:, None]]
        elif env_name == "op":        # (x, y, prize)  [nn/env_embeddings/init.py:260-286]
            extra = [_f32(demand["prize"])[:, 1:, None]]
        elif env_name == "cvrptw":    # (x, y, demand, window start, window end, service time)  [init.py:141-157]
            extra = [_f32(demand["demand"])[..., None], _f32(demand["time_windows"])[:, 1:, :],
                     _f32(demand["durations"])[:, 1:, None]]
        else:
            extra = [_f32(demand)[..., None]]
        feat = np.concatenate([_f32(locs[:, 1:])] + extra, -1)
        cust = linear(feat, sd[pre + "init_embed.weight"], sd[pre + "init_embed.bias"])
        h = np.concatenate([depot, cust], 1)
    init_h = h.copy()
    layer = 0
    while f"encoder.net.layers.{layer}.0.module.Wqkv.weight" in sd:
        p = f"encoder.net.layers.{layer}."
        qkv = linear(h, sd[p + "0.module.Wqkv.weight"], sd[p + "0.module.Wqkv.bias"])
        att = mha_encoder(qkv, num_heads)
        h = h + linear(att, sd[p + "0.module.out_proj.weight"], sd[p + "0.module.out_proj.bias"])
        h = _norm(sd, p + "1.normalizer.", h, training)
        f = linear(h, sd[p + "2.module.lins.0.weight"], sd[p + "2.module.lins.0.bias"], relu=True)
        h = h + linear(f, sd[p + "2.module.lins.1.weight"], sd[p + "2.module.lins.1.bias"])
        h = _norm(sd, p + "3.normalizer.", h, training)
        layer += 1
    return init_h, h


def _norm(sd, p, h, training=False):
    if p + "running_mean" in sd and training:
        rm, rv = (np.ascontiguousarray(sd[p + k], dtype=np.float32) for k in ("running_mean", "running_var"))
        y, _, _ = batchnorm_train(h, sd[p + "weight"], sd[p + "bias"], rm, rv)
        sd[p + "running_mean"], sd[p + "running_var"] = rm, rv
        return y
    if p + "running_mean" in sd:
        return batchnorm_eval(h, sd[p + "weight"], sd[p + "bias"], sd[p + "running_mean"], sd[p + "running_var"])
    return instancenorm(h, sd[p + "weight"], sd[p + "bias"])


def precompute(sd, env_name, emb, use_graph_context=True):
    """K/V/L cache plus the folded tensors of DESIGN.md (Pa/Pb, cvec, Lp) and the graph context."""
    env_name = _kind(env_name)
    emb = _f32(emb)
    E = emb.shape[-1]
    kvl = linear(emb, sd["decoder.project_node_embeddings.weight"])
    K, V, L = (np.ascontiguousarray(kvl[..., i * E:(i + 1) * E]) for i in range(3))
    Wctx = _f32(sd["decoder.context_embedding.project_context.weight"])
    out = {"K": K, "V": V, "L": L, "Lp": matmul_right(L, sd["decoder.pointer.project_out.weight"])}
    if env_name == "tsp":
        out["Pa"] = linear(emb, np.ascontiguousarray(Wctx[:, :E]))
        out["Pb"] = linear(emb, np.ascontiguousarray(Wctx[:, E:]))
        out["cvec"] = linear(_f32(sd["decoder.context_embedding.W_placeholder"])[None], Wctx)[0]
    else:
        out["Pa"] = linear(emb, np.ascontiguousarray(Wctx[:, :E]))
        out["Pb"] = None
        # state columns of project_context: capacity (and, CVRPTW, the current time) -> [E] or [2, E]
        out["cvec"] = np.ascontiguousarray(Wctx[:, E:].T.reshape(-1)) if env_name == "cvrptw" else np.ascontiguousarray(Wctx[:, E])
    out["gctx"] = linear(mean_nodes(emb), sd["decoder.project_fixed_context.weight"]) if use_graph_context else None
    if env_name == "sdvrp":      # SDVRPDynamicEmbedding: Linear(1, 3E, bias=False); the logit part folded like Lp
        w = _f32(sd["decoder.dynamic_embedding.projection.weight"]).reshape(3, E)
        lw = matmul_right(np.ascontiguousarray(w[2:3]), sd["decoder.pointer.project_out.weight"])[0]
        out["dyn"] = np.ascontiguousarray(np.stack([w[0], w[1], lw]).astype(np.float32))
    return out


# ---------------------------------------------------------------------------------------------
# env state + decode + rollout
# ---------------------------------------------------------------------------------------------
class State:
    """Per-row rollout state in the oracle's flat layout (R rows over Binst instances)."""

    def __init__(self, env_name, locs, demand=None, vehicle_capacity=1.0, num_starts=0):
        env_name = _kind(env_name)
        self.env_name = env_name
        self.env = {"tsp": ENV_TSP, "cvrp": ENV_CVRP, "sdvrp": ENV_SDVRP, "pctsp": ENV_PCTSP, "op": ENV_OP,
                    "cvrptw": ENV_CVRPTW}[env_name]
        self.rem = None
        self.oplocs = self.tw = self.dur = self.time = None
        if env_name == "cvrptw":     # CVRP state + clock; time windows as floats (exact: they are small integers)
            self.tw = _f32(demand["time_windows"])
            self.dur = _f32(demand["durations"])
            demand = demand["demand"]
        op_maxlen = None
        if env_name == "op":         # used = tour_length, vcap = max_length[:, 0] per row, demand = max_length [B, M]
            op_maxlen, demand, vehicle_capacity = _f32(demand["max_length"]), _f32(demand["max_length"]), 0.0
        if env_name == "pctsp":      # used = cur_total_prize, vcap = prize_required, demand = real_prize (depot slot 0)
            demand, vehicle_capacity = demand["real_prize"], float(np.asarray(demand["prize_required"]).reshape(-1)[0])
        self.locs = _f32(locs)
        self.Binst, self.M = self.locs.shape[:2]
        S = max(int(num_starts), 1)
        R = self.R = self.Binst * S
        self.first = np.zeros(R, np.int64)
        self.cur = np.zeros(R, np.int64)
        self.istep = np.zeros(R, np.int64)
        self.done = np.zeros(R, np.uint8)
        self.used = np.zeros(R, np.float32)
        self.vcap = np.full(R, vehicle_capacity, np.float32)
        if env_name == "tsp":
            self.demand = None
            self.visited = None
            self.mask = np.ones((R, self.M), np.uint8)
        elif env_name == "op":
            self.demand = op_maxlen
            self.oplocs = self.locs
            self.vcap = np.ascontiguousarray(np.tile(op_maxlen[:, 0], S))
            self.visited = np.zeros((R, self.M), np.uint8)
            self.mask = np.empty((R, self.M), np.uint8)
            lib().orc_op_mask(_p(self.visited), _p(self.used), _p(self.cur), _p(self.oplocs), _p(self.demand), _p(self.mask),
                              C.c_long(R), C.c_long(self.Binst), C.c_int(self.M))
        elif env_name == "pctsp":
            self.demand = _f32(demand)
            self.visited = np.zeros((R, self.M), np.uint8)
            self.mask = np.empty((R, self.M), np.uint8)
            lib().orc_pctsp_mask(_p(self.visited), _p(self.used), _p(self.mask), C.c_long(R), C.c_int(self.M))
        elif env_name == "sdvrp":        # remaining demand per row, depot slot 0 (sdvrp/env.py:94-118)
            self.demand = _f32(demand)
            self.visited = None
            d = np.concatenate([np.zeros((self.Binst, 1), np.float32), self.demand], 1)
            self.rem = np.ascontiguousarray(np.tile(d, (S, 1)))
            self.mask = np.empty((R, self.M), np.uint8)
            lib().orc_sdvrp_mask(_p(self.rem), _p(self.used), _p(self.vcap), _p(self.cur), _p(self.mask), C.c_long(R),
                                 C.c_int(self.M))
        elif env_name == "cvrptw":
            self.demand = _f32(demand)
            self.oplocs = self.locs
            self.time = np.zeros(R, np.float32)
            self.visited = np.zeros((R, self.M), np.uint8)
            self.mask = np.empty((R, self.M), np.uint8)
            lib().orc_cvrptw_mask(_p(self.visited), _p(self.used), _p(self.vcap), _p(self.demand), _p(self.cur),
                                  _p(self.time), _p(self.oplocs), _p(self.tw), _p(self.mask), C.c_long(R),
                                  C.c_long(self.Binst), C.c_int(self.M - 1))
        else:
            self.demand = _f32(demand)
            self.visited = np.zeros((R, self.M), np.uint8)
            self.mask = np.empty((R, self.M), np.uint8)
            lib().orc_cvrp_mask(_p(self.visited), _p(self.used), _p(self.vcap), _p(self.demand), _p(self.cur),
                                _p(self.mask), C.c_long(R), C.c_long(self.Binst), C.c_int(self.M - 1))

    def step(self, action):
        a = _i64(action)
        if self.env == ENV_TSP:
            lib().orc_tsp_step(_p(self.mask), _p(self.first), _p(self.cur), _p(self.istep), _p(a), _p(self.done),
                               C.c_long(self.R), C.c_int(self.M))
        elif self.env == ENV_CVRPTW:
            lib().orc_cvrptw_step(_p(self.visited), _p(self.used), _p(self.vcap), _p(self.demand), _p(self.cur),
                                  _p(self.time), _p(self.oplocs), _p(self.tw), _p(self.dur), _p(a), _p(self.mask),
                                  _p(self.done), C.c_long(self.R), C.c_long(self.Binst), C.c_int(self.M - 1))
        elif self.env == ENV_OP:
            lib().orc_op_step(_p(self.visited), _p(self.used), None, None, _p(self.oplocs), _p(self.demand), _p(self.cur),
                              _p(self.istep), _p(a), _p(self.mask), _p(self.done), C.c_long(self.R), C.c_long(self.Binst),
                              C.c_int(self.M))
        elif self.env == ENV_PCTSP:
            lib().orc_pctsp_step(_p(self.visited), _p(self.used), None, _p(self.demand), None, _p(self.cur), _p(self.istep),
                                 _p(a), _p(self.mask), _p(self.done), C.c_long(self.R), C.c_long(self.Binst),
                                 C.c_int(self.M))
        elif self.env == ENV_SDVRP:
            lib().orc_sdvrp_step(_p(self.rem), _p(self.used), _p(self.vcap), _p(self.cur), _p(a), _p(self.mask),
                                 _p(self.done), C.c_long(self.R), C.c_int(self.M))
        else:
            lib().orc_cvrp_step(_p(self.visited), _p(self.used), _p(self.vcap), _p(self.demand), _p(self.cur),
                                _p(a), _p(self.mask), _p(self.done), C.c_long(self.R), C.c_long(self.Binst),
                                C.c_int(self.M - 1))


def decode_step(st: State, cache, mode="greedy", noise=None, given=None, clip=10.0, temp=1.0, num_heads=8,
                want_all=False, top_k=0, top_p=0.0):
    R, M = st.R, st.M
    E = cache["K"].shape[-1]
    act = np.empty(R, np.int64)
    lp = np.empty(R, np.float32)
    logits = np.empty((R, M), np.float32) if want_all else None
    logprobs = np.empty((R, M), np.float32) if want_all else None
    nz = None if noise is None else _f32(noise)
    gv = None if given is None else _i64(given)
    rc = lib().orc_decode_step(
        C.c_int(st.env), C.c_long(R), C.c_long(st.Binst), C.c_int(M), C.c_int(E), C.c_int(num_heads),
        _p(cache["K"]), _p(cache["V"]), _p(cache["Lp"]), _p(cache["Pa"]), _p(cache["Pb"]), _p(cache["cvec"]),
        _p(cache["gctx"]), _p(st.first), _p(st.cur), _p(st.istep), _p(st.used), _p(st.vcap), _p(st.mask),
        _p(st.rem), _p(cache.get("dyn")), _p(st.time),
        C.c_int(MODES[mode]), _p(nz), _p(gv), C.c_float(clip), C.c_float(temp), C.c_int(int(top_k)), C.c_float(top_p),
        _p(act), _p(lp), _p(logits), _p(logprobs))
    if rc == -1:
        raise AssertionError("Logits contain NaNs")
    if rc == -2:
        raise AssertionError("infeasible action selected")
    return (act, lp, logits, logprobs) if want_all else (act, lp)


def rollout(st: State, cache, mode="greedy", noise=None, given=None, clip=10.0, temp=1.0, num_heads=8, t_max=None,
            top_k=0, top_p=0.0):
    """Decode loop until every row is done.  -> (actions [R,T], logp [R,T])."""
    R, M = st.R, st.M
    E = cache["K"].shape[-1]
    if t_max is None:
        t_max = M if st.env == ENV_TSP else (2 * M + 1 if st.env == ENV_CVRP else 3 * M + 1)
    if noise is not None:
        noise = _f32(noise)
        t_max = noise.shape[1]
    tg = 0
    if given is not None:
        given = _i64(given)
        tg = given.shape[1]
        t_max = max(t_max, tg) if noise is None else t_max
    actions = np.zeros((R, t_max), np.int64)
    logps = np.zeros((R, t_max), np.float32)
    T = lib().orc_rollout(
        C.c_int(st.env), C.c_long(R), C.c_long(st.Binst), C.c_int(M), C.c_int(E), C.c_int(num_heads),
        _p(cache["K"]), _p(cache["V"]), _p(cache["Lp"]), _p(cache["Pa"]), _p(cache["Pb"]), _p(cache["cvec"]),
        _p(cache["gctx"]), _p(st.first), _p(st.cur), _p(st.istep), _p(st.used), _p(st.vcap), _p(st.demand),
        _p(st.mask), _p(st.visited), _p(st.done), _p(st.rem), _p(cache.get("dyn")), _p(st.oplocs), _p(st.tw), _p(st.dur),
        _p(st.time), C.c_int(MODES[mode]), _p(noise),
        _p(given), C.c_int(tg),
        C.c_float(clip), C.c_float(temp), C.c_int(int(top_k)), C.c_float(top_p), C.c_int(t_max), _p(actions), _p(logps))
    if T == -1:
        raise AssertionError("Logits contain NaNs")
    if T == -2:
        raise AssertionError("infeasible action selected")
    return np.ascontiguousarray(actions[:, :T]), np.ascontiguousarray(logps[:, :T])


def tour_length_reward(locs, actions, with_depot, binst=None):
    locs = _f32(locs)
    actions = _i64(actions)
    R, T = actions.shape
    Binst, M = locs.shape[:2]
    out = np.empty(R, np.float32)
    lib().orc_tour_length(_p(locs), _p(actions), _p(out), C.c_long(R), C.c_long(Binst), C.c_int(M), C.c_int(T),
                          C.c_int(int(with_depot)))
    return out


def pctsp_reward(locs, penalty, actions):
    locs, penalty, actions = _f32(locs), _f32(penalty), _i64(actions)
    R, T = actions.shape
    Binst, M = locs.shape[:2]
    out = np.empty(R, np.float32)
    lib().orc_pctsp_reward(_p(locs), _p(penalty), _p(actions), _p(out), C.c_long(R), C.c_long(Binst), C.c_int(M), C.c_int(T))
    return out


def op_reward(prize, actions):
    prize, actions = _f32(prize), _i64(actions)
    R, T = actions.shape
    out = np.empty(R, np.float32)
    lib().orc_op_reward(_p(prize), _p(actions), _p(out), C.c_long(R), C.c_long(prize.shape[0]), C.c_int(prize.shape[1]),
                        C.c_int(T))
    return out


def check_op(actions, locs, max_length):
    actions, locs, max_length = _i64(actions), _f32(locs), _f32(max_length)
    lib().orc_check_op.restype = C.c_long
    return int(lib().orc_check_op(_p(actions), _p(locs), _p(max_length), C.c_long(actions.shape[0]), C.c_long(locs.shape[0]),
                                  C.c_int(locs.shape[1]), C.c_int(actions.shape[1])))


def check_cvrptw_time(actions, locs, time_windows, durations):
    actions, locs, tw, dur = _i64(actions), _f32(locs), _f32(time_windows), _f32(durations)
    lib().orc_check_cvrptw_time.restype = C.c_long
    return int(lib().orc_check_cvrptw_time(_p(actions), _p(locs), _p(tw), _p(dur), C.c_long(actions.shape[0]),
                                           C.c_long(locs.shape[0]), C.c_int(locs.shape[1]), C.c_int(actions.shape[1])))


def check_pctsp(actions, real_prize):
    actions, real_prize = _i64(actions), _f32(real_prize)
    lib().orc_check_pctsp.restype = C.c_long
    return int(lib().orc_check_pctsp(_p(actions), _p(real_prize), C.c_long(actions.shape[0]), C.c_long(real_prize.shape[0]),
                                     C.c_int(real_prize.shape[1]), C.c_int(actions.shape[1])))


def sum_logp(logp):
    logp = _f32(logp)
    R, T = logp.shape
    out = np.empty(R, np.float32)
    lib().orc_sum_logp(_p(logp), _p(out), C.c_long(R), C.c_int(T))
    return out


def check_tsp(actions):
    actions = _i64(actions)
    return int(lib().orc_check_tsp(_p(actions), C.c_long(actions.shape[0]), C.c_int(actions.shape[1])))


def check_cvrp(actions, demand, vcap):
    actions = _i64(actions)
    demand = _f32(demand)
    R, T = actions.shape
    vc = _f32(np.broadcast_to(np.asarray(vcap, np.float32).reshape(-1), (R,)))
    return int(lib().orc_check_cvrp(_p(actions), _p(demand), _p(vc), C.c_long(R), C.c_long(demand.shape[0]),
                                    C.c_int(demand.shape[1]), C.c_int(T)))


def policy_rollout(sd, env_name, locs, demand=None, decode_type="greedy", num_starts=0, noise=None, given=None,
                   use_graph_context=True, clip=10.0, temp=1.0, num_heads=8, top_k=0, top_p=0.0, start_nodes=None,
                   training=False):
    """ConstructivePolicy.forward restated on the oracle: encoder, cache, (multistart hook), loop, reward.

    locs for CVRP already include the depot at index 0 (post-reset layout).
    Returns dict(actions, logp_steps, log_likelihood, reward, steps).
    """
    env_name = _kind(env_name)
    _, emb = encode(sd, env_name, locs, demand, num_heads, training=training)
    cache = precompute(sd, env_name, emb, use_graph_context)
    multistart = "multistart" in decode_type and num_starts > 1
    st = State(env_name, locs, demand, num_starts=num_starts if multistart else 0)
    mode = "evaluate" if given is not None else ("greedy" if "greedy" in decode_type else "sampling")
    pre_a, pre_lp = [], []
    if multistart:
        # select_start_nodes: row j = s*B + b starts at node s (TSP) / s+1 (CVRP)   [utils/ops.py:133-169]
        B = st.Binst
        nloc = st.M if env_name == "tsp" else st.M - 1          # depot cannot be a start node (utils/ops.py:120-130)
        start = (np.repeat(np.arange(num_starts), B) % nloc + (0 if env_name == "tsp" else 1)).astype(np.int64)
        if start_nodes is not None:      # OP resamples its start nodes when some are infeasible (utils/ops.py:158-169)
            start = _i64(start_nodes)
        if given is not None:
            start, given = _i64(given[:, 0]), np.ascontiguousarray(given[:, 1:])
        st.step(start)
        pre_a, pre_lp = [start[:, None]], [np.zeros((st.R, 1), np.float32)]
    acts, lps = rollout(st, cache, mode, noise=noise, given=given, clip=clip, temp=temp, num_heads=num_heads,
                        top_k=top_k, top_p=top_p)
    actions = np.concatenate(pre_a + [acts], 1)
    logp = np.concatenate(pre_lp + [lps], 1)
    if env_name == "pctsp":
        reward = pctsp_reward(locs, demand["penalty"], actions)
    elif env_name == "op":
        reward = op_reward(demand["prize"], actions)
    else:
        reward = tour_length_reward(locs, actions, with_depot=(env_name != "tsp"))
    return {"actions": actions, "logp_steps": logp, "log_likelihood": sum_logp(logp), "reward": reward,
            "steps": acts.shape[1], "embeddings": emb, "cache": cache}


def policy_beam_search(sd, env_name, locs, demand=None, beam_width=None, select_best=True, use_graph_context=True,
                       clip=10.0, temp=1.0, num_heads=8):
    """decode_type="beam_search" restated (BeamSearch, rl4co/utils/decoding.py:468-608): beams start at the multistart
    nodes; each step keeps per instance the beam_width best (beam, node) continuations by cumulative log-prob
    (descending, ties to the lower beam*M + node), every new beam continues its parent's state; tours are recovered by
    backtracking the parent pointers.  Returns dict(actions, logp_steps, log_likelihood, reward)."""
    env_name = _kind(env_name)
    _, emb = encode(sd, env_name, locs, demand, num_heads)
    cache = precompute(sd, env_name, emb, use_graph_context)
    B, M = locs.shape[:2]
    nloc = M if env_name == "tsp" else M - 1
    BW = nloc if beam_width is None else int(beam_width)
    st = State(env_name, locs, demand, num_starts=BW)
    R = st.R
    start = (np.repeat(np.arange(BW), B) % nloc + (0 if env_name == "tsp" else 1)).astype(np.int64)
    st.step(start)
    inst = np.tile(np.arange(B), BW)
    actions, step_lps, parents = [start], [np.zeros(R, np.float32)], [np.zeros(R, np.int64)]
    parent_lp = np.zeros(R, np.float32)
    while not st.done.all():
        assert len(actions) <= 3 * M + 1, "beam search exceeded the maximum number of steps"
        _, _, _, logprobs = decode_step(st, cache, "greedy", clip=clip, temp=temp, num_heads=num_heads, want_all=True)
        cand = (logprobs + parent_lp[:, None]).astype(np.float32)                     # [R, M]
        flat = cand.reshape(BW, B, M).transpose(1, 0, 2).reshape(B, BW * M)          # [B, BW*M], index w*M + n
        node = np.empty(R, np.int64); beam = np.empty(R, np.int64); new_parent = np.empty(R, np.float32)
        for b in range(B):
            order = np.lexsort((np.arange(BW * M), -flat[b].astype(np.float64)))[:BW]    # value desc, index asc
            for k, c in enumerate(order):
                node[k * B + b], beam[k * B + b], new_parent[k * B + b] = c % M, c // M, flat[b, c]
        idx = inst + beam * B
        slp = logprobs[idx, node]
        for name in ("first", "cur", "istep", "done", "mask", "used", "visited", "rem", "time"):
            v = getattr(st, name)
            if v is not None:
                setattr(st, name, np.ascontiguousarray(v[idx]))
        st.step(node)
        actions.append(node); step_lps.append(slp.astype(np.float32)); parents.append(beam)
        parent_lp = new_parent
    acts = np.stack(actions, 1); lps = np.stack(step_lps, 1)
    T = acts.shape[1]
    cur_parent = parents[-1]
    seq, seq_lp = [acts[:, -1]], [lps[:, -1]]
    for k in range(T - 2, -1, -1):
        idx = inst + cur_parent * B
        seq.append(acts[idx, k]); seq_lp.append(lps[idx, k])
        cur_parent = parents[k][idx]
    actions_out = np.ascontiguousarray(np.stack(seq[::-1], 1))
    logp = np.ascontiguousarray(np.stack(seq_lp[::-1], 1))
    if env_name == "pctsp":
        reward = pctsp_reward(locs, demand["penalty"], actions_out)
    else:
        reward = tour_length_reward(locs, actions_out, with_depot=(env_name != "tsp"))
    if select_best:
        best = reward.reshape(BW, B).argmax(0)           # first maximum, as torch.max
        flat_idx = np.arange(B) + best * B
        actions_out, logp, reward = actions_out[flat_idx], logp[flat_idx], reward[flat_idx]
    return {"actions": np.ascontiguousarray(actions_out), "logp_steps": np.ascontiguousarray(logp),
            "log_likelihood": sum_logp(np.ascontiguousarray(logp)), "reward": reward}
