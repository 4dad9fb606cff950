"""Seeded random-init weights for every network on the hot path.

There is no network for checkpoints, so tests and ``bench.py`` use random-init
weights of the reference architectures.  Every generator returns a flat
``{state_dict key: fp32 CPU tensor}`` mapping whose key names and shapes are the
reference's ``state_dict`` names, so a real checkpoint can
be fed to the same loaders.  Values come from a ``torch.Generator`` on the CPU and
are therefore identical in this container and on the GPU box.

Scales follow ``torch.nn`` defaults (U(-1/sqrt(fan_in), 1/sqrt(fan_in))) except
where the reference zero-initialises a layer (``vertice_map_r``
models/faceformer.py:156-157, EMOTE ``decoder`` FaceFormerDecoder.py:1037-1038):
a zero layer would make parity tests vacuous, so those get the default scale too.
"""
import math

import torch


class _Init:
    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.w = {}

    def uniform(self, name, shape, bound):
        self.w[name] = (torch.rand(shape, generator=self.g) * 2 - 1) * bound

    def normal(self, name, shape, std=1.0, mean=0.0):
        self.w[name] = torch.randn(shape, generator=self.g) * std + mean

    def linear(self, prefix, out_f, in_f, bias=True, scale=1.0):
        b = scale / math.sqrt(in_f)
        self.uniform(prefix + ".weight", (out_f, in_f), b)
        if bias:
            self.uniform(prefix + ".bias", (out_f,), b)

    def conv1d(self, prefix, out_c, in_c, k, bias=True, scale=1.0):
        b = scale / math.sqrt(in_c * k)
        self.uniform(prefix + ".weight", (out_c, in_c, k), b)
        if bias:
            self.uniform(prefix + ".bias", (out_c,), b)

    def norm(self, prefix, dim, bias=True, wname="weight"):
        self.normal(prefix + "." + wname, (dim,), 0.1, 1.0)
        if bias:
            self.normal(prefix + ".bias", (dim,), 0.1)


# ----------------------------------------------------------------------------- wav2vec2
W2V_CONV_KERNEL = (10, 3, 3, 3, 3, 2, 2)


def make_wav2vec2_weights(seed=0):
    """HF ``Wav2Vec2Model`` state_dict names (models/lib/wav2vec.py:76 subclasses it)."""
    I = _Init(seed)
    I.normal("masked_spec_embed", (768,), 1.0)
    for i, k in enumerate(W2V_CONV_KERNEL):
        cin = 1 if i == 0 else 512
        # gain sqrt(3)*1.3: keeps the GELU conv stack's activations O(1) through 7 layers
        I.conv1d(f"feature_extractor.conv_layers.{i}.conv", 512, cin, k, bias=False, scale=2.2)
    I.norm("feature_extractor.conv_layers.0.layer_norm", 512)
    I.norm("feature_projection.layer_norm", 512)
    I.linear("feature_projection.projection", 768, 512)
    I.uniform("encoder.pos_conv_embed.conv.bias", (768,), 1 / math.sqrt(48 * 128))
    I.normal("encoder.pos_conv_embed.conv.parametrizations.weight.original0", (1, 1, 128), 0.2, 1.5)
    I.normal("encoder.pos_conv_embed.conv.parametrizations.weight.original1", (768, 48, 128), 0.02)
    I.norm("encoder.layer_norm", 768)
    for l in range(12):
        p = f"encoder.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            I.linear(p + "attention." + n, 768, 768, scale=1.7)
        I.norm(p + "layer_norm", 768)
        I.linear(p + "feed_forward.intermediate_dense", 3072, 768, scale=1.7)
        I.linear(p + "feed_forward.output_dense", 768, 3072, scale=1.7)
        I.norm(p + "final_layer_norm", 768)
    return I.w


# ----------------------------------------------------------------------------- EMOTE head + FLINT
def make_emote_weights(seed=1, n_identities=32, n_shape=300):
    """EMOTE sequence encoder/decoder + FLINT motion-prior decoder.

    Key names follow the inferno module tree under ``TalkingHeadBase``:
    ``sequence_encoder.linear`` (SequenceEncoders.py:180-197),
    ``sequence_decoder.obj_vector.map`` (FaceFormerDecoder.py:261-268),
    ``sequence_decoder.bert_decoder.layers.0.*`` (:995-1002),
    ``sequence_decoder.decoder`` (:1034), ``sequence_decoder.squasher_2.linear`` (:967-985),
    ``sequence_decoder.motion_prior.motion_decoder.*`` (L2lMotionPrior.py:361-495).
    """
    I = _Init(seed)
    I.linear("sequence_encoder.linear", 128, 768)
    cond = 8 + 3 + n_identities + n_shape            # bertprior_wild.yaml style_embedding
    I.linear("sequence_decoder.obj_vector.map", 128, cond)
    _tel(I, "sequence_decoder.bert_decoder.layers.0", 128, 128)
    I.linear("sequence_decoder.decoder", 256, 128)
    I.linear("sequence_decoder.squasher_2.linear", 256, 256 * 8)
    m = "sequence_decoder.motion_prior.motion_decoder."
    # ConvTranspose1d weight layout is (in, out, k)
    I.uniform(m + "expander.0.0.weight", (256, 256, 5), 1.7 / math.sqrt(256 * 5 / 2))
    I.uniform(m + "expander.0.0.bias", (256,), 1 / math.sqrt(256 * 5))
    _bn(I, m + "expander.0.2", 256)
    for i in (1, 2):
        I.conv1d(m + f"expander.{i}.0", 256, 256, 5, scale=1.7)
        _bn(I, m + f"expander.{i}.2", 256)
    I.linear(m + "decoder_linear_embedding", 256, 256)
    _tel(I, m + "decoder_transformer.layers.0", 256, 384)
    I.conv1d(m + "cross_smooth_layer", 53, 256, 5)
    return I.w


def _tel(I, p, d, ff):
    """torch.nn.TransformerEncoderLayer parameter names."""
    I.uniform(p + ".self_attn.in_proj_weight", (3 * d, d), 1.7 / math.sqrt(d))
    I.uniform(p + ".self_attn.in_proj_bias", (3 * d,), 1 / math.sqrt(d))
    I.linear(p + ".self_attn.out_proj", d, d, scale=1.7)
    I.linear(p + ".linear1", ff, d, scale=1.7)
    I.linear(p + ".linear2", d, ff, scale=1.7)
    I.norm(p + ".norm1", d)
    I.norm(p + ".norm2", d)


def _bn(I, p, c):
    I.normal(p + ".weight", (c,), 0.1, 1.0)
    I.normal(p + ".bias", (c,), 0.1)
    I.normal(p + ".running_mean", (c,), 0.1)
    I.w[p + ".running_var"] = torch.rand((c,), generator=I.g) * 0.5 + 0.75
    I.w[p + ".num_batches_tracked"] = torch.tensor(100, dtype=torch.long)


# ----------------------------------------------------------------------------- FaceFormer decoder
def make_faceformer_weights(seed=2, feature_dim=64, vertice_dim=53):
    """models/faceformer.py:138-158 parameter names (decoder part only)."""
    I = _Init(seed)
    D = feature_dim
    I.linear("audio_feature_map", D, 768)
    I.linear("vertice_map", D, vertice_dim)
    p = "transformer_decoder.layers.0"
    for a in ("self_attn", "multihead_attn"):
        I.uniform(f"{p}.{a}.in_proj_weight", (3 * D, D), 1.7 / math.sqrt(D))
        I.uniform(f"{p}.{a}.in_proj_bias", (3 * D,), 1 / math.sqrt(D))
        I.linear(f"{p}.{a}.out_proj", D, D, scale=1.7)
    I.linear(p + ".linear1", 2 * D, D, scale=1.7)
    I.linear(p + ".linear2", D, 2 * D, scale=1.7)
    for n in ("norm1", "norm2", "norm3"):
        I.norm(f"{p}.{n}", D)
    I.linear("vertice_map_r", vertice_dim, D)        # reference zero-inits (:156-157); see module doc
    I.normal("obj_embedding", (1, D), 0.5)           # reference: zeros (:153)
    return I.w


# ----------------------------------------------------------------------------- aligner + diffusion prior
def make_prior_weights(seed=3, dim=128, depth=6, dim_head=64, heads=8, ff_mult=4, h=4096, n_blocks=4,
                       in_dim=768):
    """``InstructDiffusionPrior`` state_dict names:
    ``voxel2clip.*`` = BrainNetwork (models/diffusion_prior.py:58-117),
    ``net.*`` = VersatileDiffusionPriorNetwork (:169-313) with dalle2 submodule names."""
    I = _Init(seed)
    v = "voxel2clip."
    I.linear(v + "lin0.0", h, in_dim)
    I.norm(v + "lin0.1", h)
    for b in range(n_blocks):
        I.linear(v + f"mlp.{b}.0", h, h, scale=1.5)
        I.norm(v + f"mlp.{b}.1", h)
    I.linear(v + "lin1", dim, h)
    I.norm(v + "projector.0", dim)
    I.linear(v + "projector.2", 2048, dim)
    I.norm(v + "projector.3", 2048)
    I.linear(v + "projector.5", 2048, 2048)
    I.norm(v + "projector.6", 2048)
    I.linear(v + "projector.8", dim, 2048)

    n = "net."
    # to_time_embeds = Sequential(Sequential(SinusoidalPosEmb, MLP), Rearrange); MLP.net =
    # Sequential(Seq(Linear,SiLU,Identity), Seq(Linear,SiLU,Identity), Linear): hidden = 2*dim
    hid = 2 * dim
    I.linear(n + "to_time_embeds.0.1.net.0.0", hid, dim)
    I.linear(n + "to_time_embeds.0.1.net.1.0", hid, hid)
    I.linear(n + "to_time_embeds.0.1.net.2", dim, hid)
    I.normal(n + "learned_query", (1, dim), dim ** -0.5)
    I.normal(n + "null_brain_embeds", (1, dim), 1.0)
    I.normal(n + "null_image_embed", (1, dim), 1.0)
    c = n + "causal_transformer."
    I.normal(c + "rel_pos_bias.relative_attention_bias.weight", (32, heads), 1.0)
    inner = dim_head * heads
    ffi = ff_mult * dim
    for l in range(depth):
        a = c + f"layers.{l}.0."
        I.norm(a + "norm", dim, bias=False, wname="g")
        I.normal(a + "null_kv", (2, dim_head), 1.0)
        I.linear(a + "to_q", inner, dim, bias=False)
        I.linear(a + "to_kv", 2 * dim_head, dim, bias=False)
        I.linear(a + "to_out.0", dim, inner, bias=False)
        I.norm(a + "to_out.1", dim, bias=False, wname="g")
        f = c + f"layers.{l}.1."
        I.norm(f + "0", dim, bias=False, wname="g")
        I.linear(f + "1", 2 * ffi, dim, bias=False)
        I.linear(f + "5", dim, ffi, bias=False)
    I.norm(c + "norm", dim, bias=False, wname="g")
    I.linear(c + "project_out", dim, dim, bias=False)
    return I.w


def make_flame_basis(seed=4, n_vertices=5023, n_shape=300, n_exp=50):
    """Synthetic FLAME model with the buffer names, shapes and kinematic tree of the licensed ``generic_model.pkl``
    (absent; DecaFLAME.py:60-85): v_template (V,3), shapedirs (V,3,n_shape+n_exp), posedirs (36, V*3),
    J_regressor (5,V) (rows sum to 1, sparse support like the real one), lbs_weights (V,5) (rows sum to 1),
    parents [-1,0,1,1,1] (global, neck, jaw, left eye, right eye).  Magnitudes follow the real model's scale
    (head ~0.2 m, millimetre blend shapes)."""
    g = torch.Generator().manual_seed(seed)
    V = n_vertices
    r = lambda *s: torch.randn(*s, generator=g)
    v_template = r(V, 3) * 0.08
    shapedirs = r(V, 3, n_shape + n_exp) * 1.5e-3
    posedirs = r(36, V * 3) * 2e-3
    Jr = torch.zeros(5, V)
    for j in range(5):
        idx = torch.randperm(V, generator=g)[:40]
        w = torch.rand(40, generator=g)
        Jr[j, idx] = w / w.sum()
    lw = torch.rand(V, 5, generator=g) ** 4
    lw = lw / lw.sum(1, keepdim=True)
    return {"v_template": v_template, "shapedirs": shapedirs, "posedirs": posedirs, "J_regressor": Jr,
            "lbs_weights": lw, "parents": torch.tensor([-1, 0, 1, 1, 1], dtype=torch.long)}


# ----------------------------------------------------------------------------- CLIP text encoder (the text branch)
def make_clip_text_weights(seed=5, vocab=49408, hidden=768, ffn=3072, layers=12, max_pos=77):
    """HF ``CLIPTextModel`` state_dict names for openai/clip-vit-large-patch14's text tower
    (``FrozenCLIPEmbedder`` models/diffusion_prior.py:29-55)."""
    I = _Init(seed)
    I.normal("text_model.embeddings.token_embedding.weight", (vocab, hidden), 0.02)
    I.normal("text_model.embeddings.position_embedding.weight", (max_pos, hidden), 0.01)
    for l in range(layers):
        p = f"text_model.encoder.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            I.linear(p + "self_attn." + n, hidden, hidden, scale=1.7)
        I.norm(p + "layer_norm1", hidden)
        I.linear(p + "mlp.fc1", ffn, hidden, scale=1.7)
        I.linear(p + "mlp.fc2", hidden, ffn, scale=1.7)
        I.norm(p + "layer_norm2", hidden)
    I.norm("text_model.final_layer_norm", hidden)
    return I.w
