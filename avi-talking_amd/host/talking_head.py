"""Host-side mirror of the EMOTE talking head used by the reference entry point:
``TalkingHeadWrapper`` (inferno_apps/TalkingHead/evaluation/TalkingHeadWrapper.py:83-138) ->
``TalkingHeadBase.forward`` (inferno/models/talkinghead/TalkingHeadBase.py:503-553) with the
wav2vec2 audio model, ``LinearSequenceEncoder``, ``BertPriorDecoder`` and the FLINT ``L2lDecoder``.

Sample-dict protocol: input keys ``raw_audio`` (B,T,640) int16/fp32 or
``processed_audio`` (B,N) fp32, ``samplerate``, ``gt_shape`` (B,300),
``gt_expression_{label,intensity,identity}_condition`` one-hots; output keys ``audio_feature``,
``seq_encoder_output``, ``predicted_exp`` (B,T,50), ``predicted_jaw`` (B,T,3).
``predicted_vertices`` needs the licensed FLAME model and stays with the reference's FLAME module
(out of scope: "3DMM visualizer untouched").

All arithmetic runs in the HIP kernels behind the C ABI; inference only (BatchNorm in eval mode).
"""
import math

import torch

from .. import ops
from .wav2vec import Wav2Vec2Model

LATENT_FRAME = 8
N_EXP, N_JAW = 50, 3


def alibi_slopes(n):
    """inferno/models/temporal/TransformerMasking.py:46-56."""
    def pow2(n):
        start = 2 ** (-2 ** -(math.log2(n) - 3))
        return [start * start ** i for i in range(n)]
    if math.log2(n).is_integer():
        return pow2(n)
    c = 2 ** math.floor(math.log2(n))
    return pow2(c) + alibi_slopes(2 * c)[0::2][:n - c]


class _EncoderLayer:
    """torch.nn.TransformerEncoderLayer (post-LN, batch_first) over the C-ABI ops."""

    def __init__(self, w, p, d, nhead, act, prec):
        self.d, self.nhead, self.act, self.prec = d, nhead, act, prec
        self.qkv = ops.PackedWeight(w[p + ".self_attn.in_proj_weight"], w[p + ".self_attn.in_proj_bias"])
        self.out = ops.PackedWeight(w[p + ".self_attn.out_proj.weight"], w[p + ".self_attn.out_proj.bias"])
        self.l1 = ops.PackedWeight(w[p + ".linear1.weight"], w[p + ".linear1.bias"])
        self.l2 = ops.PackedWeight(w[p + ".linear2.weight"], w[p + ".linear2.bias"])
        self.n1 = (w[p + ".norm1.weight"], w[p + ".norm1.bias"])
        self.n2 = (w[p + ".norm2.weight"], w[p + ".norm2.bias"])

    def __call__(self, x, bias_mode=0, slopes=None, period=1):
        B, T, d = x.shape
        dh = d // self.nhead
        qkv = ops.linear(x, self.qkv, prec=self.prec)
        att = ops.attention(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], self.nhead, dh, 3 * d, 3 * d, T, T, B,
                            dh ** -0.5, bias_mode=bias_mode, slopes=slopes, period=period)
        x = ops.linear(att, self.out, residual=x, prec=self.prec)
        x = ops.layernorm(x, *self.n1, out=x)
        f = ops.linear(x, self.l1, act=self.act, prec=self.prec)
        x = ops.linear(f, self.l2, residual=x, prec=self.prec)
        return ops.layernorm(x, *self.n2, out=x)


def _pad_k(w2d, mult=64):
    """Zero-pad the K (input) dimension of a Linear weight to a multiple of 64."""
    N, K = w2d.shape
    Kp = (K + mult - 1) // mult * mult
    if Kp == K:
        return w2d
    out = torch.zeros((N, Kp), dtype=w2d.dtype, device=w2d.device)
    out[:, :K] = w2d
    return out


class EmoteHead:
    """LinearSequenceEncoder + BertPriorDecoder + FLINT decoder (row D of SURVEY.md 8a)."""

    def __init__(self, state_dict, device="cuda", prec=ops.PREC_BF16X3):
        self.device = torch.device(device)
        self.prec = prec
        w = {k: v.detach().to(self.device, torch.float32).contiguous() for k, v in state_dict.items()
             if v.is_floating_point()}
        self.seq_enc = ops.PackedWeight(w["sequence_encoder.linear.weight"], w["sequence_encoder.linear.bias"])
        cw = w["sequence_decoder.obj_vector.map.weight"]
        self.cond_dim = cw.shape[1]
        self.cond = ops.PackedWeight(_pad_k(cw), w["sequence_decoder.obj_vector.map.bias"])
        self.bert = _EncoderLayer(w, "sequence_decoder.bert_decoder.layers.0", 128, 8, ops.ACT_GELU, prec)
        self.dec = ops.PackedWeight(w["sequence_decoder.decoder.weight"], w["sequence_decoder.decoder.bias"])
        self.squash = ops.PackedWeight(w["sequence_decoder.squasher_2.linear.weight"],
                                       w["sequence_decoder.squasher_2.linear.bias"])
        m = "sequence_decoder.motion_prior.motion_decoder."
        wt = w[m + "expander.0.0.weight"]                                   # ConvTranspose1d: (in, out, k)
        # stride-2 transposed conv = two interleaved stride-1 convs over the zero-padded latent:
        #   even t=2u : taps j=4,2,0 on x[u-1],x[u],x[u+1];  odd t=2u+1 : taps j=3,1 on x[u],x[u+1]
        even = torch.stack([wt[:, :, 4], wt[:, :, 2], wt[:, :, 0]], 0)      # (slot, in, out)
        odd = torch.stack([wt[:, :, 3], wt[:, :, 1]], 0)
        bt = w[m + "expander.0.0.bias"]
        self.ct_even = ops.PackedWeight(even.permute(2, 0, 1).reshape(256, -1), bt)
        self.ct_odd = ops.PackedWeight(odd.permute(2, 0, 1).reshape(256, -1), bt)
        self.bn = [self._bn_affine(w, m + f"expander.{i}.2") for i in range(3)]
        self.convs = [ops.PackedWeight(w[m + f"expander.{i}.0.weight"].permute(0, 2, 1).reshape(256, -1),
                                       w[m + f"expander.{i}.0.bias"]) for i in (1, 2)]
        self.lin = ops.PackedWeight(w[m + "decoder_linear_embedding.weight"], w[m + "decoder_linear_embedding.bias"])
        self.tel = _EncoderLayer(w, m + "decoder_transformer.layers.0", 256, 8, ops.ACT_GELU, prec)
        self.smooth = ops.PackedWeight(w[m + "cross_smooth_layer.weight"].permute(0, 2, 1).reshape(53, -1),
                                       w[m + "cross_smooth_layer.bias"])
        self.slopes = torch.tensor(alibi_slopes(8), dtype=torch.float32, device=self.device)

    @staticmethod
    def _bn_affine(w, p, eps=1e-5):
        sc = w[p + ".weight"] / torch.sqrt(w[p + ".running_var"] + eps)
        return sc.contiguous(), (w[p + ".bias"] - w[p + ".running_mean"] * sc).contiguous()

    def style_condition(self, expr_onehot, intensity_onehot, identity_onehot, shape):
        """LinearEmotionCondition (FaceFormerDecoder.py:186-268): -> (B,T,128)."""
        B, T = expr_onehot.shape[:2]
        cond = torch.zeros((B, T, self.cond.K), dtype=torch.float32, device=self.device)
        parts = [expr_onehot, intensity_onehot, identity_onehot, shape[:, None, :].expand(B, T, shape.shape[-1])]
        o = 0
        for p_ in parts:
            cond[..., o:o + p_.shape[-1]] = p_.to(self.device, torch.float32)
            o += p_.shape[-1]
        if o != self.cond_dim:
            raise ValueError(f"condition has {o} features, the style map expects {self.cond_dim}")
        return ops.linear(cond, self.cond, prec=self.prec)

    def flint_decoder(self, z, out_dtype=torch.float32):
        """L2lDecoder.forward (L2lMotionPrior.py:460-495): z (B,Tl,256) -> (B,8*Tl,53); ``out_dtype`` float16: the last
        layer's epilogue stores the coefficients as IEEE half."""
        B, Tl, _ = z.shape
        P = self.prec
        zp = ops.pad_repeat(z, 1, 1, 1, 0)
        x = torch.empty((B, 2 * Tl, 256), dtype=torch.float32, device=self.device)
        sc, sh = self.bn[0]
        ops.conv1d_cl(zp, self.ct_even, 3, 1, out=x, act=ops.ACT_LRELU02, prec=P, scale=sc, shift=sh,
                      out_rows=Tl, out_row_stride=512, out_offset=0)
        ops.conv1d_cl(zp, self.ct_odd, 2, 1, out=x, act=ops.ACT_LRELU02, prec=P, scale=sc, shift=sh,
                      out_rows=Tl, out_row_stride=512, out_offset=256, in_row_offset=1)
        rep = 1
        for i, pw in enumerate(self.convs):
            xp = ops.pad_repeat(x, rep, 2, 2, 1)                 # replicate padding (+ previous repeat_interleave)
            sc, sh = self.bn[i + 1]
            x = ops.conv1d_cl(xp, pw, 5, 1, act=ops.ACT_LRELU02, prec=P, scale=sc, shift=sh)
            rep = 2
        x = ops.pad_repeat(x, 2, 0, 0, 0)
        x = ops.linear(x, self.lin, prec=P)
        x = self.tel(x, bias_mode=1, slopes=self.slopes)
        xp = ops.pad_repeat(x, 1, 2, 2, 0)
        return ops.conv1d_cl(xp, self.smooth, 5, 1, prec=P, out_dtype=out_dtype)

    def forward(self, audio_feature, style_emb, out_dtype=torch.float32):
        """audio_feature (B,T,768), style_emb (B,1,128)/(B,128)/(B,T,128) -> dict with predicted_exp/jaw (``out_dtype``
        float32, or float16 for long-form batches: BASELINE.json configs[4] "fp16 coeffs")."""
        B, T, _ = audio_feature.shape
        P = self.prec
        h = ops.linear(audio_feature, self.seq_enc, prec=P)
        style = style_emb.to(self.device, torch.float32)
        if style.dim() == 3 and style.shape[1] == T and T != 1:
            styled = _add_full(h, style.contiguous())
        else:
            styled = ops.add_rowbcast(h, style.reshape(B, -1).contiguous())
        d = self.bert(styled)
        d = ops.linear(d, self.dec, prec=P)
        T_pad = int(math.ceil(T / LATENT_FRAME) * LATENT_FRAME)
        dp = ops.pad_repeat(d, 1, 0, T_pad - T, 0)
        # (B*T/8) x 2048 -> 256: sixteen 128-row tiles walking 32 K tiles each are pure latency (72 us); K slices as
        # one batched launch + the partial-sum epilogue take a third of that
        z = ops.linear_ln_skinny(dp.view(B, T_pad // LATENT_FRAME, LATENT_FRAME * 256), self.squash, do_ln=False,
                                 prec=P)
        seq = self.flint_decoder(z, out_dtype)[:, :T]
        return {"seq_encoder_output": h, "latent": z,
                "predicted_exp": seq[..., :N_EXP], "predicted_jaw": seq[..., N_EXP:N_EXP + N_JAW]}

    __call__ = forward


def _add_full(h, style):
    """per-frame style (B,T,128): one add_rowbcast over B*T pseudo-batches of one row each."""
    B, T, C = h.shape
    return ops.add_rowbcast(h.view(B * T, 1, C), style.reshape(B * T, C).contiguous()).view(B, T, C)


class TalkingHeadWrapper:
    """Mirror of ``TalkingHeadWrapper.forward(sample, style_emb, only_style_emb, is_external_style_emb)``
    (TalkingHeadWrapper.py:123-138; patched FeedForwardDecoder.forward, FaceFormerDecoder.py:598-611)."""

    def __init__(self, audio_state_dict, head_state_dict, device="cuda", prec=None, joint_norm=True):
        self.device = torch.device(device)
        prec = ops.prec_plan(prec)            # None = ops.DEFAULT_PREC
        # EMOTE's resampled wav2vec2 rounds the output length up (AudioEncoders.py:19-20)
        self.audio_model = Wav2Vec2Model(audio_state_dict, device=device, prec=prec, length_mode="ceil")
        self.head = EmoteHead(head_state_dict, device=device, prec=ops.fp32_operand_prec(prec))
        self.joint_norm = joint_norm        # AudioEncoders.py:170-178: HF processor sees ONE (B*L) array

    # the three counts the reference's sample builders ask the wrapper for (TalkingHeadWrapper.py:113-121); the identity count
    # follows from the style map's input width (expressions + intensities + identities + 300 shape coefficients)
    def get_num_emotions(self):
        return 8

    def get_num_intensities(self):
        return 3

    def get_num_identities(self):
        return self.head.cond_dim - 8 - 3 - 300

    def forward_audio(self, sample, cus=0, front_only=False):
        """Wav2Vec2Encoder._forward (AudioEncoders.py:165-200).  ``cus``: compute units free for the big GEMMs (0 = all).
        ``front_only``: stop in front of the 12 transformer layers and return their input (B, T, 768) instead of the sample
        (the pipelined replay runs the layers of two groups of clips as separate graphs: audio_model.encoder_layers)."""
        if "raw_audio" in sample:
            raw = sample["raw_audio"].to(self.device)
            B, T = raw.shape[0], raw.shape[1]
            pcm = raw.reshape(B, -1).contiguous()
            if pcm.dtype not in (torch.int16, torch.float32):
                pcm = pcm.to(torch.float32)
            x = ops.audio_normalize(pcm, joint=self.joint_norm)
            sample["processed_audio"] = x
        else:
            x = sample["processed_audio"].to(self.device, torch.float32).contiguous()
            T = sample.get("frame_num")
        if front_only:
            return self.audio_model.front(x, frame_num=T, cus=cus)
        out = self.audio_model(x, frame_num=T, cus=cus)
        sample["audio_feature"] = out.last_hidden_state
        return sample

    def forward(self, sample, style_emb=None, only_style_emb=False, is_external_style_emb=False):
        sample = self.forward_audio(sample)
        B, T = sample["audio_feature"].shape[:2]
        if only_style_emb or not (style_emb is not None and is_external_style_emb):
            def cond(key):
                c = sample[key].to(self.device, torch.float32)
                if c.dim() == 2:
                    c = c[:, None, :]
                return c.expand(B, T, c.shape[-1]) if c.shape[1] == 1 else c
            own = self.head.style_condition(cond("gt_expression_label_condition"),
                                            cond("gt_expression_intensity_condition"),
                                            cond("gt_expression_identity_condition"),
                                            sample["gt_shape"].to(self.device, torch.float32))
            if only_style_emb:
                return own                                        # FaceFormerDecoder.py:599-601
            style_emb = own
        out = self.head(sample["audio_feature"], style_emb)
        sample.update(out)
        return sample

    __call__ = forward
