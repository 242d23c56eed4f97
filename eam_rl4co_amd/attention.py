"""The reference's two attention injection points, served by the HIP library.

  * `PointerAttention` -- drop-in for rl4co/models/nn/attention.py:224-328 with the same constructor, parameter name
    (`project_out`) and call signature, so `AttentionModelDecoder(pointer=PointerAttention(E, H))` of the UNMODIFIED
    reference decoder computes its logits with one `eamrl_pointer_attention` launch per step
    (rl4co/models/zoo/am/decoder.py:82,109-124).
  * `scaled_dot_product_attention` -- an `sdpa_fn(q, k, v, attn_mask=None, dropout_p=0.0)` for the encoder's
    MultiHeadAttention (rl4co/models/nn/attention.py:93-100,126-135), on `eamrl_mha_encoder`.

Both are inference-path functions (no autograd graph): like the rollout, gradients come from the re-evaluation
(train.py).  The fused rollout kernels do not go through these; they exist so that the reference's own modules can
call into the HIP path through the seams the reference offers.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


class PointerAttention(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int, mask_inner: bool = True, out_bias: bool = False,
                 check_nan: bool = True, sdpa_fn="default", **kwargs):
        super().__init__()
        if callable(sdpa_fn):
            raise NotImplementedError("PointerAttention on MI355X computes its inner attention in the same launch: a custom "
                                      "sdpa_fn cannot be injected")
        self.num_heads, self.mask_inner, self.check_nan = num_heads, mask_inner, check_nan
        self.project_out = nn.Linear(embed_dim, embed_dim, bias=out_bias)

    @torch.no_grad()
    def forward(self, query, key, value, logit_key, attn_mask=None):
        """query [B, L, E]; key / value / logit_key [B, S, E]; attn_mask [B, S] or [B, L, S], True = may attend.
        -> logits [B, L, S] (squeezed to [B, S] when L == 1, as the reference)."""
        logits = ops.pointer_attention(query.contiguous(), key, value, logit_key,
                                       None if attn_mask is None else attn_mask.contiguous(), self.project_out.weight,
                                       self.project_out.bias, self.num_heads, self.mask_inner)
        if self.check_nan:
            assert not torch.isnan(logits).any(), "Logits contain NaNs"
        return logits


@torch.no_grad()
def scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p: float = 0.0, is_causal: bool = False):
    """q, k, v [B, H, N, D] (the head views of one packed Wqkv output) -> [B, H, N, D]; unmasked self-attention only
    (what the AM encoder uses: attnnet.py:94-103 asserts mask is None)."""
    if attn_mask is not None or is_causal or dropout_p:
        raise NotImplementedError("the MI355X encoder attention is unmasked, non-causal and without dropout")
    B, H, N, D = q.shape
    if k.shape != q.shape or v.shape != q.shape:
        raise NotImplementedError("self-attention only (q, k, v of one shape)")
    qkv = torch.stack((q, k, v), 0).permute(1, 3, 0, 2, 4).reshape(B, N, 3 * H * D).contiguous()   # "b s (three h d)"
    out = ops.mha_encoder(qkv, H)                                                                    # [B, N, H*D]
    return out.view(B, N, H, D).permute(0, 2, 1, 3)
