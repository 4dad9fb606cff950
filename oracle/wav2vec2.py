"""Oracle: wav2vec2-base audio encoder with 50->25 Hz resampling (fp32, CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, over a flat ``{state_dict key: tensor}`` mapping ``w``:
  * reference ``models/lib/wav2vec.py:80-156`` (Wav2Vec2Model.forward: CNN ->
    linear_interpolation -> feature_projection -> encoder), and its EMOTE twin
    ``inferno/models/temporal/AudioEncoders.py:16-24,27-128``;
  * the HF ``transformers`` wav2vec2-base modules those wrappers subclass
    (config = ``Wav2Vec2Config()`` defaults: conv_dim 512x7, kernel
    (10,3,3,3,3,2,2), stride (5,2,2,2,2,2,2), no conv bias, GroupNorm on conv
    layer 0, GELU(erf), 12 post-LN layers, 12 heads, ffn 3072, pos-conv k=128
    groups=16 with weight-norm over dim 2, layer_norm_eps 1e-5).
Pinned by tests/golden/wav2vec2_*.npz (made by tests/golden/make_golden.py from
the reference import).
"""
import math

import torch
import torch.nn.functional as F

CONV_KERNEL = (10, 3, 3, 3, 3, 2, 2)
CONV_STRIDE = (5, 2, 2, 2, 2, 2, 2)
CONV_DIM = 512
HIDDEN = 768
HEADS = 12
FFN = 3072
LAYERS = 12
POS_K = 128
POS_GROUPS = 16
LN_EPS = 1e-5


def conv_out_lengths(n_samples):
    """HF ``_get_feat_extract_output_lengths``: floor((L - k)/s) + 1 per layer."""
    lens = []
    L = n_samples
    for k, s in zip(CONV_KERNEL, CONV_STRIDE):
        L = (L - k) // s + 1
        lens.append(L)
    return lens


def feature_extractor(w, x, return_all=False):
    """HF Wav2Vec2FeatureEncoder: x (B, N) -> (B, 512, L6).

    models/lib/wav2vec.py:97 (``self.feature_extractor(input_values)``).
    Layer 0: Conv1d(1,512,10,s5,no bias) -> GroupNorm(512 groups, eps 1e-5,
    affine) -> GELU.  Layers 1-6: Conv1d(512,512,k,s,no bias) -> GELU.
    """
    h = x[:, None, :]
    outs = []
    for i, (k, s) in enumerate(zip(CONV_KERNEL, CONV_STRIDE)):
        h = F.conv1d(h, w[f"feature_extractor.conv_layers.{i}.conv.weight"], stride=s)
        if i == 0:
            h = F.group_norm(h, CONV_DIM,
                             w["feature_extractor.conv_layers.0.layer_norm.weight"],
                             w["feature_extractor.conv_layers.0.layer_norm.bias"], eps=1e-5)
        h = F.gelu(h)
        outs.append(h)
    return outs if return_all else h


def resample_length(L50, mode="int"):
    """Output length of the 50->25 Hz resample.

    ``int``: models/lib/wav2vec.py:69-71 (``int(L/50*25)``);
    ``ceil``: inferno AudioEncoders.py:19-20 (``int(math.ceil(L/50*25))``).
    """
    seq_len = L50 / float(50)
    return int(seq_len * 25) if mode == "int" else int(math.ceil(seq_len * 25))


def linear_interpolation(feat, output_len):
    """models/lib/wav2vec.py:67-73: (B, L, C) -> (B, output_len, C),
    ``F.interpolate(mode='linear', align_corners=True)`` along time."""
    y = F.interpolate(feat.transpose(1, 2), size=output_len, align_corners=True, mode="linear")
    return y.transpose(1, 2)


def pos_conv_weight(w):
    """Fold the weight-norm parametrisation (dim=2): W = g * v / ||v||_{(0,1)}.
    HF Wav2Vec2PositionalConvEmbedding (``weight_norm(self.conv, dim=2)``)."""
    g = w["encoder.pos_conv_embed.conv.parametrizations.weight.original0"]  # (1,1,K)
    v = w["encoder.pos_conv_embed.conv.parametrizations.weight.original1"]  # (768,48,K)
    norm = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    return g * v / norm


def _ln(x, w, pre):
    return F.layer_norm(x, (x.shape[-1],), w[pre + ".weight"], w[pre + ".bias"], LN_EPS)


def _lin(x, w, pre):
    return F.linear(x, w[pre + ".weight"], w[pre + ".bias"])


def encoder_layer(w, i, h):
    """HF Wav2Vec2EncoderLayer (post-LN), eager attention, no mask."""
    p = f"encoder.layers.{i}."
    B, T, C = h.shape
    d = C // HEADS
    q = _lin(h, w, p + "attention.q_proj").view(B, T, HEADS, d).transpose(1, 2)
    k = _lin(h, w, p + "attention.k_proj").view(B, T, HEADS, d).transpose(1, 2)
    v = _lin(h, w, p + "attention.v_proj").view(B, T, HEADS, d).transpose(1, 2)
    a = torch.softmax(torch.matmul(q, k.transpose(2, 3)) * d ** -0.5, dim=-1)
    o = torch.matmul(a, v).transpose(1, 2).reshape(B, T, C)
    h = h + _lin(o, w, p + "attention.out_proj")
    h = _ln(h, w, p + "layer_norm")
    f = _lin(F.gelu(_lin(h, w, p + "feed_forward.intermediate_dense")), w, p + "feed_forward.output_dense")
    h = _ln(h + f, w, p + "final_layer_norm")
    return h


def encoder(w, h, return_all=False):
    """HF Wav2Vec2Encoder: pos-conv(+GELU) residual, LN, 12 layers.
    models/lib/wav2vec.py:142-148."""
    pc = F.conv1d(h.transpose(1, 2), pos_conv_weight(w), w["encoder.pos_conv_embed.conv.bias"],
                  padding=POS_K // 2, groups=POS_GROUPS)
    pc = F.gelu(pc[:, :, :-1]).transpose(1, 2)           # SamePad drops the last step (even k)
    h = _ln(h + pc, w, "encoder.layer_norm")
    hs = [h]
    for i in range(LAYERS):
        h = encoder_layer(w, i, h)
        hs.append(h)
    return hs if return_all else h


def forward(w, input_values, frame_num=None, length_mode="int", return_intermediates=False):
    """models/lib/wav2vec.py:80-156 in eval mode, attention_mask=None.

    input_values (B, N) fp32 (already normalised) -> last_hidden_state (B, T, 768).
    ``frame_num`` plays the role of ``frame_num`` (wav2vec.py:88,108) /
    ``desired_output_length`` (AudioEncoders.py:47,60).
    """
    feats = feature_extractor(w, input_values)                       # (B,512,L)
    h = feats.transpose(1, 2)
    T = frame_num if frame_num is not None else resample_length(h.shape[1], length_mode)
    h25 = linear_interpolation(h, T)                                 # wav2vec.py:108
    hp = _lin(_ln(h25, w, "feature_projection.layer_norm"), w, "feature_projection.projection")  # :120
    out = encoder(w, hp)
    if return_intermediates:
        return {"conv": feats, "interp": h25, "proj": hp, "last_hidden_state": out}
    return out


def normalize_audio(x, joint=False, eps=1e-7):
    """HF Wav2Vec2FeatureExtractor.zero_mean_unit_var_norm.

    Per clip (dataset/data_loader.py:289-290 behaviour) or, with ``joint``, over
    all B*L samples at once -- the quirk of AudioEncoders.py:170-178 where a
    (B, L) tensor is treated as ONE un-batched input by the HF processor.
    """
    x = x.to(torch.float32)
    if joint:
        return (x - x.mean()) / torch.sqrt(x.var(unbiased=False) + eps)
    m = x.mean(dim=-1, keepdim=True)
    v = x.var(dim=-1, unbiased=False, keepdim=True)
    return (x - m) / torch.sqrt(v + eps)
