"""GPU: the fused per-instance encoder kernel (eamrl_encoder_fused: all layers of an instance in one workgroup, activations
in LDS, every Linear on v_mfma_f32_16x16x4_f32) is bit-identical to the layer-by-layer launches and to the oracle."""
import numpy as np
import pytest
import torch

from _util import golden_weights, instance_from_td
from test_gpu_parity import DEV, assert_bits_equal, make_policy

pytestmark = pytest.mark.gpu


def _encode(pol, td, fused, monkeypatch):
    monkeypatch.setenv("EAMRL_FUSED_ENCODER", "1" if fused else "0")
    with torch.no_grad():
        h, init_h = pol.encoder(td)
    return h, init_h


@pytest.mark.parametrize("cfg,env_name,N,B", [
    ("am_tsp", "tsp", 5, 3), ("am_tsp", "tsp", 16, 2), ("am_tsp", "tsp", 17, 2), ("am_tsp", "tsp", 20, 7), ("am_tsp", "tsp", 32, 3),
    ("am_tsp", "tsp", 33, 3), ("am_tsp", "tsp", 50, 5), ("am_tsp", "tsp", 64, 2), ("am_tsp", "tsp", 65, 2), ("am_tsp", "tsp", 100, 9),
    ("am_tsp", "tsp", 112, 2), ("am_cvrp", "cvrp", 20, 4), ("am_cvrp", "cvrp", 100, 5), ("am_cvrp", "cvrp", 111, 2),
    ("pomo_tsp", "tsp", 20, 3), ("pomo_tsp", "tsp", 100, 4), ("pomo_cvrp", "cvrp", 50, 3), ("am_pctsp", "pctsp", 30, 2),
])
def test_fused_encoder_is_bit_identical(oracle, monkeypatch, cfg, env_name, N, B):
    import eam_rl4co_amd as ea
    from eam_rl4co_amd import ops

    env = ea.get_env(env_name, generator_params=dict(num_loc=N), seed=N)
    torch.manual_seed(N * 31 + B)
    td_cpu = env.reset(batch_size=[B])
    td = td_cpu.to(DEV)
    pol = make_policy(cfg)
    M = td["locs"].shape[1]
    assert ops.encoder_fused_supported(M, 128, 8, 512, len(pol.encoder.net.layers))
    h_f, init_f = _encode(pol, td, True, monkeypatch)
    h_u, init_u = _encode(pol, td, False, monkeypatch)
    assert_bits_equal(init_f, init_u, "init embeddings")
    assert_bits_equal(h_f, h_u, f"embeddings fused vs layer-by-layer (M={M})")
    init_o, h_o = oracle.encode(golden_weights(cfg), env_name, td_cpu["locs"].numpy(), instance_from_td(env_name, td_cpu))
    assert_bits_equal(h_f, h_o, "embeddings vs oracle")


def test_fused_encoder_limits_and_fallbacks(monkeypatch):
    import eam_rl4co_amd as ea
    from eam_rl4co_amd import ops

    assert not ops.encoder_fused_supported(113, 128, 8, 512, 3)
    assert not ops.encoder_fused_supported(100, 64, 8, 512, 3)
    assert not ops.encoder_fused_supported(100, 128, 8, 512, 9)
    env = ea.get_env("tsp", generator_params=dict(num_loc=120), seed=1)
    td = env.reset(batch_size=[2]).to(DEV)
    pol = make_policy("am_tsp")
    h, _ = pol.encoder(td)                   # 120 nodes: layer-by-layer path
    assert h.shape == (2, 120, 128) and torch.isfinite(h).all()
    assert pol.encoder.net._fused_layers(torch.empty(2, 120, 128, device=DEV)) is None
    pol.train()                              # batch statistics: not fused
    assert pol.encoder.net._fused_layers(torch.empty(2, 20, 128, device=DEV)) is None
    pol.eval()
    assert pol.encoder.net._fused_layers(torch.empty(2, 20, 128, device=DEV)) is not None


def test_fused_encoder_follows_weight_updates(monkeypatch):
    """The packed weights are re-packed in place when a parameter changes."""
    import eam_rl4co_amd as ea

    env = ea.get_env("tsp", generator_params=dict(num_loc=20), seed=2)
    td = env.reset(batch_size=[4]).to(DEV)
    pol = make_policy("am_tsp")
    h0, _ = _encode(pol, td, True, monkeypatch)
    ptr = pol.encoder.net._packed[(0, "W1")][1].data_ptr()
    with torch.no_grad():
        for p in pol.parameters():
            p.mul_(1.01)
    h1, _ = _encode(pol, td, True, monkeypatch)
    h1u, _ = _encode(pol, td, False, monkeypatch)
    assert_bits_equal(h1, h1u, "after a weight update")
    assert not torch.equal(h0, h1) and pol.encoder.net._packed[(0, "W1")][1].data_ptr() == ptr


@pytest.mark.parametrize("cfg,env_name,N,B", [("am_tsp", "tsp", 20, 5), ("am_tsp", "tsp", 100, 6), ("am_cvrp", "cvrp", 20, 4),
                                              ("am_cvrp", "cvrp", 100, 3), ("pomo_tsp", "tsp", 50, 3), ("am_sdvrp", "sdvrp", 33, 2),
                                              ("am_op", "op", 64, 2), ("am_cvrptw", "cvrptw", 20, 3)])
def test_fused_cache_is_bit_identical(monkeypatch, cfg, env_name, N, B):
    """The decoder cache written by the fused kernel (projections of the LDS-resident embeddings + Lp) == the GEMM launches."""
    import eam_rl4co_amd as ea

    env = ea.get_env(env_name, generator_params=dict(num_loc=N), seed=N + 1)
    torch.manual_seed(N * 7 + B)
    td = env.reset(batch_size=[B]).to(DEV)
    pol = make_policy(cfg)
    M = td["locs"].shape[1]
    with torch.no_grad():
        spec = pol.decoder._fused_cache_spec(B, M, DEV)
        h, _ = pol.encoder(td, cache_spec=spec)
        assert spec["filled"]
        fused = pol.decoder._precompute_cache(h, prefilled=spec)
        ref = pol.decoder._precompute_cache(h)
    assert fused.buf.data_ptr() == spec["buf"].data_ptr() and ref.buf.data_ptr() != spec["buf"].data_ptr()
    assert_bits_equal(fused.buf, ref.buf, "slot-major cache")
    out_f = pol(td.clone(), env, phase="test", decode_type="greedy")
    monkeypatch.setenv("EAMRL_FUSED_CACHE", "0")
    monkeypatch.setenv("EAMRL_FUSED_ENCODER", "0")
    out_u = pol(td.clone(), env, phase="test", decode_type="greedy")
    assert_bits_equal(out_f["actions"], out_u["actions"], "actions")
    assert_bits_equal(out_f["log_likelihood"], out_u["log_likelihood"], "ll")


@pytest.mark.parametrize("cfg,env_name,N,B", [("am_tsp", "tsp", 20, 5), ("am_tsp", "tsp", 100, 6), ("am_tsp", "tsp", 112, 2),
                                              ("am_cvrp", "cvrp", 20, 4), ("am_cvrp", "cvrp", 100, 3), ("pomo_tsp", "tsp", 50, 3),
                                              ("am_sdvrp", "sdvrp", 33, 2), ("am_pctsp", "pctsp", 30, 3), ("am_op", "op", 64, 2),
                                              ("am_cvrptw", "cvrptw", 20, 3)])
def test_fused_init_embedding_and_skipped_hidden_store(monkeypatch, cfg, env_name, N, B):
    """eamrl_encoder_fused_init: the init embedding computed inside the fused kernel (features -> LDS) gives the same init
    embeddings, node embeddings, decoder cache and graph context, bit for bit, as the init-embedding launch + the fused kernel
    on its output; and a rollout whose embeddings never leave LDS (want_hidden=False) the same tours and log-likelihoods."""
    import eam_rl4co_amd as ea

    env = ea.get_env(env_name, generator_params=dict(num_loc=N), seed=N + 3)
    torch.manual_seed(N * 11 + B)
    td = env.reset(batch_size=[B]).to(DEV)
    pol = make_policy(cfg)
    M = td["locs"].shape[1]
    with torch.no_grad():
        spec_a = pol.decoder._fused_cache_spec(B, M, DEV)
        h_a, init_a = pol.encoder(td, cache_spec=spec_a)                               # fused init, everything stored
        spec_n = pol.decoder._fused_cache_spec(B, M, DEV)
        h_n, init_n = pol.encoder(td, cache_spec=spec_n, want_hidden=False, want_init=False)
        monkeypatch.setenv("EAMRL_FUSED_INIT", "0")
        spec_b = pol.decoder._fused_cache_spec(B, M, DEV)
        h_b, init_b = pol.encoder(td, cache_spec=spec_b)                               # init-embedding launch + fused kernel
    assert spec_a["filled"] and spec_b["filled"] and spec_n["filled"]
    assert h_n is None and init_n is None
    assert_bits_equal(init_a, init_b, "init embeddings")
    assert_bits_equal(h_a, h_b, "embeddings")
    assert_bits_equal(spec_a["buf"], spec_b["buf"], "decoder cache")
    assert_bits_equal(spec_n["buf"], spec_b["buf"], "decoder cache (embeddings kept in LDS)")
    if spec_b.get("gctx") is not None:
        assert_bits_equal(spec_a["gctx"], spec_b["gctx"], "graph context")
        assert_bits_equal(spec_n["gctx"], spec_b["gctx"], "graph context (embeddings kept in LDS)")
    out_old = pol(td.clone(), env, phase="test", decode_type="greedy", return_hidden=True, return_init_embeds=True)
    monkeypatch.delenv("EAMRL_FUSED_INIT")
    out_new = pol(td.clone(), env, phase="test", decode_type="greedy")                 # default: nothing but the cache leaves
    out_hid = pol(td.clone(), env, phase="test", decode_type="greedy", return_hidden=True, return_init_embeds=True)
    for o in (out_new, out_hid):
        assert_bits_equal(o["actions"], out_old["actions"], "actions")
        assert_bits_equal(o["log_likelihood"], out_old["log_likelihood"], "ll")
        assert_bits_equal(o["reward"], out_old["reward"], "reward")
    assert_bits_equal(out_hid["init_embeds"], out_old["init_embeds"], "returned init embeddings")
    assert_bits_equal(out_hid["hidden"].node_embeddings, out_old["hidden"].node_embeddings, "returned embeddings")
