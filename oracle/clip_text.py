"""Oracle: CLIP text transformer as the reference's frozen text embedder runs it (fp32, CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, over a flat ``{state_dict key: tensor}`` mapping ``w`` with HF ``CLIPTextModel`` key names:
  * reference ``models/diffusion_prior.py:29-55`` (``FrozenCLIPEmbedder``): tokens padded to ``max_length=77`` ->
    ``self.transformer(input_ids=tokens).last_hidden_state`` of ``CLIPTextModel("openai/clip-vit-large-patch14")``
    (``:48-53``); the tokenizer (``:49-51``) is host string work outside this path - the oracle takes token ids;
  * the HF ``transformers`` modules behind that call, with the text config of clip-vit-large-patch14 (hidden 768,
    12 layers, 12 heads, ffn 3072, 77 positions, vocab 49408, ``quick_gelu``, layer_norm_eps 1e-5):
    CLIPTextEmbeddings (token + position), 12 pre-LN CLIPEncoderLayer blocks under a causal mask (no padding mask:
    the reference passes ``input_ids`` only), ``final_layer_norm``.
The algorithm lives in a dependency (``transformers``, not vendored by the reference), which IS importable here:
the restatement is pinned against ``transformers.CLIPTextModel`` itself on the same seeded weights, live in
tests/test_oracle_golden.py and through the committed tests/golden/clip_text.npz (made by
``python tests/golden/make_golden.py clip_text``).  Pretrained CLIP weights are not available offline, so the
parity inputs are random-init weights of that architecture.
"""
import torch
import torch.nn.functional as F

HIDDEN = 768
HEADS = 12
LAYERS = 12
MAX_POS = 77
EPS = 1e-5


def quick_gelu(x):
    """transformers activations.QuickGELUActivation: x * sigmoid(1.702 x)."""
    return x * torch.sigmoid(1.702 * x)


def clip_text_forward(w, input_ids, layers=LAYERS, heads=HEADS):
    """(B, T) int64 -> last_hidden_state (B, T, hidden): CLIPTextTransformer.forward as called at
    models/diffusion_prior.py:52-53."""
    p = "text_model."
    B, T = input_ids.shape
    h = F.embedding(input_ids, w[p + "embeddings.token_embedding.weight"]) \
        + w[p + "embeddings.position_embedding.weight"][:T]
    C = h.shape[-1]
    d = C // heads
    causal = torch.full((T, T), float("-inf")).triu(1)
    for l in range(layers):
        q = f"{p}encoder.layers.{l}."
        x = F.layer_norm(h, (C,), w[q + "layer_norm1.weight"], w[q + "layer_norm1.bias"], EPS)
        qh = F.linear(x, w[q + "self_attn.q_proj.weight"], w[q + "self_attn.q_proj.bias"]) * d ** -0.5
        kh = F.linear(x, w[q + "self_attn.k_proj.weight"], w[q + "self_attn.k_proj.bias"])
        vh = F.linear(x, w[q + "self_attn.v_proj.weight"], w[q + "self_attn.v_proj.bias"])
        qh, kh, vh = (t.view(B, T, heads, d).transpose(1, 2) for t in (qh, kh, vh))
        a = torch.softmax(qh @ kh.transpose(-1, -2) + causal, dim=-1) @ vh
        a = a.transpose(1, 2).reshape(B, T, C)
        h = h + F.linear(a, w[q + "self_attn.out_proj.weight"], w[q + "self_attn.out_proj.bias"])
        x = F.layer_norm(h, (C,), w[q + "layer_norm2.weight"], w[q + "layer_norm2.bias"], EPS)
        x = quick_gelu(F.linear(x, w[q + "mlp.fc1.weight"], w[q + "mlp.fc1.bias"]))
        h = h + F.linear(x, w[q + "mlp.fc2.weight"], w[q + "mlp.fc2.bias"])
    return F.layer_norm(h, (C,), w[p + "final_layer_norm.weight"], w[p + "final_layer_norm.bias"], EPS)
