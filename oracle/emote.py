"""Oracle: EMOTE sequence encoder/decoder head + FLINT motion-prior decoder (fp32, CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates (paths under third_party/inferno/inferno/models/):
  * temporal/SequenceEncoders.py:180-197     LinearSequenceEncoder (768 -> 128)
  * talkinghead/FaceFormerDecoder.py:598-682 FeedForwardDecoder.forward / _style (style_op "add")
  * talkinghead/FaceFormerDecoder.py:64-268  EmotionCondition._gather_condition + LinearEmotionCondition
  * talkinghead/FaceFormerDecoder.py:987-1224 BertPriorDecoder (_decode with post_bug_fix=True,
    _apply_motion_prior zero-padding to a multiple of the latent frame size, squash_after)
  * talkinghead/FaceFormerDecoder.py:967-985 StackLinearSquash
  * temporal/motion_prior/L2lMotionPrior.py:361-495 L2lDecoder (ConvTranspose expander, replicate-padded
    convs + repeat_interleave, Linear, TransformerEncoderLayer with ALiBi-future bias, smoothing conv)
  * temporal/motion_prior/MotionPrior.py:316-329 decompose_sequential_output (exp 50 | jaw 3)
  * temporal/TransformerMasking.py:80-98 init_alibi_biased_mask_future
Configuration: talkinghead_conf/model/sequence_decoder/bertprior_wild.yaml (feature_dim 128, nhead 8,
1 layer, gelu, squash_after stack_linear, post_bug_fix, no PE, no temporal bias) and
motion_prior_conf l2l_decoder.yaml / l2l_sizes.yaml (dim 256, ff 384, 8 heads, quant_factor 3,
alibi_future).  BatchNorm runs in eval mode (FaceFormerDecoder.py:1061-1067 keeps the prior in eval).

Pinned (tests/test_oracle_golden.py::test_emote_oracle_matches_reference, tests/golden/emote.npz) against the
reference's OWN classes run unmodified in the build container - LinearSequenceEncoder, LinearEmotionCondition,
BertPriorDecoder.forward/_style/_decode/_apply_motion_prior, StackLinearSquash, MotionPrior.decoding_step and
L2lDecoder - to 2e-6 (tests/golden/make_golden.py: functional stand-ins for omegaconf/munch/pytorch_lightning,
and the MultiheadAttention fast path of torch >= 1.12 switched off: the reference pins torch 1.9, which has none,
and the fast path mishandles L2lDecoder's (B*H,T,T) float mask).  The masks are also pinned against the importable
TransformerMasking.py (tests/golden/masks.npz).
"""
import math

import torch
import torch.nn.functional as F

LATENT_FRAME = 8          # 2 ** quant_factor
N_EXP, N_JAW = 50, 3


def get_slopes(n):
    """TransformerMasking.py:46-56 (ALiBi head slopes)."""
    def pow2(n):
        start = 2 ** (-2 ** -(math.log2(n) - 3))
        return [start * start ** i for i in range(n)]
    if math.log2(n).is_integer():
        return pow2(n)
    c = 2 ** math.floor(math.log2(n))
    return pow2(c) + get_slopes(2 * c)[0::2][:n - c]


def alibi_future_mask(num_heads, T):
    """TransformerMasking.py:80-98: symmetric -slope_h * |i-j|, nothing masked out."""
    slopes = torch.tensor(get_slopes(num_heads), dtype=torch.float32)
    i = torch.arange(T)
    d = (i[:, None] - i[None, :]).abs().to(torch.float32)
    return -slopes[:, None, None] * d[None]


def faceformer_biased_mask(num_heads, T, period):
    """models/faceformer.py:51-72 == TransformerMasking.py:101-120: causal, -slope*floor((i-j)/period)."""
    slopes = torch.tensor(get_slopes(num_heads), dtype=torch.float32)
    i = torch.arange(T)
    d = (i[:, None] - i[None, :])
    bias = -slopes[:, None, None] * (d // period).clamp(min=0).to(torch.float32)[None]
    return bias.masked_fill((d < 0)[None], float("-inf"))


def mha(w, p, x, mem, nhead, mask=None):
    """torch.nn.MultiheadAttention (batch_first) with packed in_proj; mask (H,Tq,Tk) float or bool."""
    B, Tq, d = x.shape
    Tk = mem.shape[1]
    Wi, bi = w[p + ".in_proj_weight"], w[p + ".in_proj_bias"]
    q = F.linear(x, Wi[:d], bi[:d]).view(B, Tq, nhead, d // nhead).transpose(1, 2)
    k = F.linear(mem, Wi[d:2 * d], bi[d:2 * d]).view(B, Tk, nhead, d // nhead).transpose(1, 2)
    v = F.linear(mem, Wi[2 * d:], bi[2 * d:]).view(B, Tk, nhead, d // nhead).transpose(1, 2)
    s = torch.matmul(q, k.transpose(2, 3)) * (d // nhead) ** -0.5
    if mask is not None:
        if mask.dtype == torch.bool:
            s = s.masked_fill(mask, float("-inf"))
        else:
            s = s + mask
    o = torch.matmul(torch.softmax(s, -1), v).transpose(1, 2).reshape(B, Tq, d)
    return F.linear(o, w[p + ".out_proj.weight"], w[p + ".out_proj.bias"])


def _ln(w, p, x):
    return F.layer_norm(x, (x.shape[-1],), w[p + ".weight"], w[p + ".bias"], 1e-5)


def encoder_layer(w, p, x, nhead, mask=None, act=F.gelu):
    """torch.nn.TransformerEncoderLayer, norm_first=False (post-LN), eval mode."""
    x = _ln(w, p + ".norm1", x + mha(w, p + ".self_attn", x, x, nhead, mask))
    f = F.linear(act(F.linear(x, w[p + ".linear1.weight"], w[p + ".linear1.bias"])),
                 w[p + ".linear2.weight"], w[p + ".linear2.bias"])
    return _ln(w, p + ".norm2", x + f)


def style_condition(w, expr_onehot, intensity_onehot, identity_onehot, shape):
    """EmotionCondition._gather_condition + LinearEmotionCondition.map (FaceFormerDecoder.py:186-268):
    cat[expr(8), intensity(3), identity(N), shape(300)] over T, then Linear -> (B,T,128)."""
    T = expr_onehot.shape[1]
    shp = shape[:, None, :].expand(shape.shape[0], T, shape.shape[1])
    cond = torch.cat([expr_onehot.float(), intensity_onehot.float(), identity_onehot.float(), shp], -1)
    return F.linear(cond, w["sequence_decoder.obj_vector.map.weight"], w["sequence_decoder.obj_vector.map.bias"])


def _bn(w, p, x_bct):
    return F.batch_norm(x_bct, w[p + ".running_mean"], w[p + ".running_var"], w[p + ".weight"], w[p + ".bias"],
                        False, 0.0, 1e-5)


def flint_decoder(w, z, return_pre_conv=False):
    """L2lDecoder.forward (L2lMotionPrior.py:460-495): z (B, Tl, 256) -> (B, 8*Tl, 53)."""
    m = "sequence_decoder.motion_prior.motion_decoder."
    x = z.permute(0, 2, 1)
    x = F.conv_transpose1d(x, w[m + "expander.0.0.weight"], w[m + "expander.0.0.bias"], stride=2, padding=2,
                           output_padding=1)
    x = _bn(w, m + "expander.0.2", F.leaky_relu(x, 0.2)).permute(0, 2, 1)
    for i in (1, 2):
        xp = F.pad(x.permute(0, 2, 1), (2, 2), mode="replicate")
        y = F.conv1d(xp, w[m + f"expander.{i}.0.weight"], w[m + f"expander.{i}.0.bias"])
        x = _bn(w, m + f"expander.{i}.2", F.leaky_relu(y, 0.2)).permute(0, 2, 1)
        x = x.repeat_interleave(2, dim=1)
    x = F.linear(x, w[m + "decoder_linear_embedding.weight"], w[m + "decoder_linear_embedding.bias"])
    T = x.shape[1]
    x = encoder_layer(w, m + "decoder_transformer.layers.0", x, 8, alibi_future_mask(8, T))
    if return_pre_conv:
        return x
    y = F.conv1d(x.permute(0, 2, 1), w[m + "cross_smooth_layer.weight"], w[m + "cross_smooth_layer.bias"], padding=2)
    return y.permute(0, 2, 1)


def forward(w, audio_feature, style_emb, return_intermediates=False):
    """TalkingHeadBase.forward after forward_audio (TalkingHeadBase.py:531-553) with an external
    style embedding: audio_feature (B,T,768), style_emb (B,1,128) or (B,T,128) ->
    predicted_exp (B,T,50), predicted_jaw (B,T,3)."""
    B, T, _ = audio_feature.shape
    h = F.linear(audio_feature, w["sequence_encoder.linear.weight"], w["sequence_encoder.linear.bias"])
    styled = h + style_emb                                                       # _style, op "add"
    d = encoder_layer(w, "sequence_decoder.bert_decoder.layers.0", styled, 8)    # _decode
    d = F.linear(d, w["sequence_decoder.decoder.weight"], w["sequence_decoder.decoder.bias"])
    T_pad = int(math.ceil(T / LATENT_FRAME) * LATENT_FRAME)                      # _apply_motion_prior
    dp = F.pad(d, (0, 0, 0, T_pad - T))
    z = F.linear(dp.reshape(B, T_pad // LATENT_FRAME, -1),                       # StackLinearSquash
                 w["sequence_decoder.squasher_2.linear.weight"], w["sequence_decoder.squasher_2.linear.bias"])
    seq = flint_decoder(w, z)[:, :T]
    out = {"predicted_exp": seq[..., :N_EXP], "predicted_jaw": seq[..., N_EXP:N_EXP + N_JAW]}
    if return_intermediates:
        out.update(seq_encoder_output=h, bert=d, latent=z)
    return out
