"""Host-side mirror of the reference audio encoder: ``models/lib/wav2vec.py`` ``Wav2Vec2Model``
(and its EMOTE twin ``inferno/models/temporal/AudioEncoders.py`` ``Wav2Vec2ModelResampled``).

Same call signature and result object as the reference wrapper (models/lib/wav2vec.py:80-156);
the arithmetic runs in the HIP kernels behind the C ABI.  Weights come from a HF-named
``state_dict`` (``Wav2Vec2Model.from_state_dict``), so a real wav2vec2-base checkpoint loads as is.

Inference only: dropout/SpecAugment/LayerDrop are training-time branches of the reference
(models/lib/wav2vec.py:123-141) and are not taken in ``eval()``.
"""
import math
import os
from types import SimpleNamespace

import torch

from .. import ops

CONV_KERNEL = (10, 3, 3, 3, 3, 2, 2)
CONV_STRIDE = (5, 2, 2, 2, 2, 2, 2)
HIDDEN, HEADS, LAYERS, POS_K, POS_G = 768, 12, 12, 128, 16
SPLIT_MIN_ROWS = int(os.environ.get("AVI_W2V_SPLIT_MIN_ROWS", "2000"))   # rows (clips x frames) per encoder chain


def conv_out_lengths(n):
    out = []
    for k, s in zip(CONV_KERNEL, CONV_STRIDE):
        n = (n - k) // s + 1
        out.append(n)
    return out


class Wav2Vec2Model:
    """wav2vec2-base with 50->25 Hz linear resampling between the CNN and the transformer."""

    def __init__(self, state_dict, device="cuda", prec=None, length_mode="int"):
        self.device = torch.device(device)
        # prec: None (= ops.DEFAULT_PREC), an AVI_PREC_* value, a plan name, or an ops.PrecPlan: one precision per group of plane-operand
        # GEMMs (conv layers 1-6 | q/k/v/out | ffn: 98 % of the FLOPs).  A 2-term fp16 group (PREC_F16X2) reads fp16 hi/lo
        # planes and one fp16 weight plane; the small fp32-operand GEMMs (feature projection, pos-conv) stay on the 3-term bf16 split
        self.plan = plan = ops.prec_plan(prec)
        self.prec = prec = plan.small
        if self.device.type == "cuda" and ops.plan_uses_fp16_planes(plan):
            from . import status
            status.words()      # fp16 planes have a finite range: their producers report overflow / underflow here
        self.length_mode = length_mode          # "int": wav2vec.py:69-71; "ceil": AudioEncoders.py:19-20
        # Activation format.  Conv stack: bf16 hi/lo planes feeding the 256x256 ping-pong GEMM (gemm_pp.hip), each
        # layer's epilogue emitting the next layer's planes (AVI_W2V_PLANES=0: fp32 activations + gemm.hip).
        # Transformer: planes as well - LayerNorm, attention and ffn1 emit them, the four projections run on the
        # 128x192 ping-pong GEMM (gemm_pp192.hip: 63 x N/192 tiles fill 256 CUs to 98 % at M = 8000)
        # (AVI_W2V_TF_PLANES=0: fp32 activations + gemm.hip's 128x128 tiles).
        self.use_planes = os.environ.get("AVI_W2V_PLANES", "1") == "1"
        self.use_planes_tf = os.environ.get("AVI_W2V_TF_PLANES", "1") == "1"
        # positional conv: its own kernel (AVI_W2V_POSCONV=gemm: the overlapping-row GEMM per (clip, group) on gemm.hip)
        self.posconv_kernel = os.environ.get("AVI_W2V_POSCONV", "kernel") != "gemm"
        # stream-K for the 128-row projections: measured slower than the data-parallel launch at these sizes (csrc/
        # gemm_pp192.hip), so the 400 MB workspace is only allocated on request (AVI_W2V_STREAMK=1 + AVI_GEMM_STREAMK=1/2)
        self.stream_k = os.environ.get("AVI_W2V_STREAMK", "0") == "1"
        self._sk_ws = None
        # the 12 transformer layers as independent chains of clips on separate streams (_encoder_layers_split):
        # AVI_W2V_SPLIT = number of chains; default 2 when the process has the hardware queues for it and the batch is
        # big enough that half of it still fills the chip's tiles (>= SPLIT_MIN_ROWS rows per chain)
        from .. import HW_QUEUES
        env = os.environ.get("AVI_W2V_SPLIT")
        self.split_streams = int(env) if env is not None else (2 if HW_QUEUES >= 8 else 1)
        self._split_streams = []
        w = {k: v.detach().to(self.device, torch.float32).contiguous() for k, v in state_dict.items()
             if v.is_floating_point()}
        fe = "feature_extractor.conv_layers."
        self.w0 = w[fe + "0.conv.weight"].reshape(512, 10).contiguous()
        self.gn_g = w[fe + "0.layer_norm.weight"]
        self.gn_b = w[fe + "0.layer_norm.bias"]
        # Conv1d weight (Cout, Cin, k) -> tap-major [Cout][k*Cin] for the overlapping-row GEMM
        self.convs = [ops.PackedWeight(w[fe + f"{i}.conv.weight"].permute(0, 2, 1).reshape(512, -1))
                      for i in range(1, 7)]
        self.fp_g = w["feature_projection.layer_norm.weight"]
        self.fp_b = w["feature_projection.layer_norm.bias"]
        self.proj = ops.PackedWeight(w["feature_projection.projection.weight"], w["feature_projection.projection.bias"])
        # pos-conv: fold weight-norm (dim=2), regroup to [G][64 (48 used)][k*48]
        g_ = w["encoder.pos_conv_embed.conv.parametrizations.weight.original0"]
        v_ = w["encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
        wpc = g_ * v_ / v_.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()            # (768, 48, 128)
        cg = HIDDEN // POS_G
        wg = wpc.view(POS_G, cg, cg, POS_K).permute(0, 1, 3, 2).reshape(POS_G, cg, POS_K * cg)
        wpad = torch.zeros((POS_G, 64, POS_K * cg), device=self.device)
        wpad[:, :cg] = wg
        self.pos = ops.PackedWeight(wpad.reshape(POS_G * 64, POS_K * cg))
        self.pos_bias = w["encoder.pos_conv_embed.conv.bias"]
        self.enc_g = w["encoder.layer_norm.weight"]
        self.enc_b = w["encoder.layer_norm.bias"]
        self.layers = []
        for i in range(LAYERS):
            p = f"encoder.layers.{i}."
            a = p + "attention."
            qkv_w = torch.cat([w[a + "q_proj.weight"], w[a + "k_proj.weight"], w[a + "v_proj.weight"]], 0)
            qkv_b = torch.cat([w[a + "q_proj.bias"], w[a + "k_proj.bias"], w[a + "v_proj.bias"]], 0)
            self.layers.append(SimpleNamespace(
                qkv=ops.PackedWeight(qkv_w, qkv_b),
                out=ops.PackedWeight(w[a + "out_proj.weight"], w[a + "out_proj.bias"]),
                ln1=(w[p + "layer_norm.weight"], w[p + "layer_norm.bias"]),
                ff1=ops.PackedWeight(w[p + "feed_forward.intermediate_dense.weight"],
                                     w[p + "feed_forward.intermediate_dense.bias"]),
                ff2=ops.PackedWeight(w[p + "feed_forward.output_dense.weight"],
                                     w[p + "feed_forward.output_dense.bias"]),
                ln2=(w[p + "final_layer_norm.weight"], w[p + "final_layer_norm.bias"])))

        # the single fp16 weight plane of a 2-term group is derived from hi / lo on first use (ops.PackedWeight.f16_plane):
        # build it HERE, on the constructing stream - the first use may otherwise come from one encoder chain's stream
        # while another chain reads the half-built plane from its own
        two_term = lambda p: (p & 0xff) == ops.PREC_F16X2
        for pw in (self.convs if two_term(plan.conv) else []):
            pw.f16_plane()
        for ly in self.layers:
            for pw in ((ly.qkv, ly.out) if two_term(plan.attn) else ()) + ((ly.ff1, ly.ff2) if two_term(plan.ffn) else ()):
                pw.f16_plane()
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()

    @classmethod
    def from_state_dict(cls, state_dict, **kw):
        return cls(state_dict, **kw)

    def eval(self):
        return self

    # ------------------------------------------------------------------ stages
    def feature_extractor(self, input_values, cus=0):
        """(B, N) -> channels-last (B, L, 512) (the reference returns (B, 512, L)).
        Activations between the conv layers travel as split bf16 planes (x = hi + lo): each producer splits once in
        its epilogue and the LDS-DMA GEMM consumes the planes without any conversion in its loop."""
        if not self.use_planes:
            h = ops.conv0_gn_gelu(input_values, self.w0, self.gn_g, self.gn_b)
            for pw, k, s in zip(self.convs, CONV_KERNEL[1:], CONV_STRIDE[1:]):
                h = ops.conv1d_cl(h, pw, k, s, act=ops.ACT_GELU, prec=self.prec)
            return h
        pc = self.plan.conv
        h = ops.conv0_gn_gelu_planes(input_values, self.w0, self.gn_g, self.gn_b, fmt=ops.plane_fmt(pc))
        n = len(self.convs)
        for i, (pw, k, s) in enumerate(zip(self.convs, CONV_KERNEL[1:], CONV_STRIDE[1:])):
            h = ops.conv1d_cl_planes(h, pw, k, s, act=ops.ACT_GELU, prec=pc, out_planes=i + 1 < n, cus=cus)
        return h

    def output_length(self, L50, frame_num=None):
        if frame_num is not None:
            return int(frame_num)
        seq_len = L50 / 50.0
        return int(seq_len * 25) if self.length_mode == "int" else int(math.ceil(seq_len * 25))

    def encoder(self, hp, cus=0, _inner=False):
        return self._encoder_layers(self.pos_embed(hp), cus, _inner)

    def pos_embed(self, hp):
        """hp + GELU(grouped positional conv(hp)): the input of the 12 layers."""
        B, T, _ = hp.shape
        cg = HIDDEN // POS_G
        if self.posconv_kernel:      # one launch, the input window resident in LDS (csrc/posconv.hip)
            return ops.posconv_gelu_residual(hp, self.pos, self.pos_bias, POS_G, POS_K, 64)
        xg = ops.group_pad_pack(hp, POS_G, POS_K // 2)
        h = torch.empty_like(hp)
        Tp = T + POS_K
        ops.gemm_raw(A=xg.data_ptr(), lda=cg, Whi=self.pos.hi.data_ptr(), Wlo=self.pos.lo.data_ptr(),
                     C_=h.data_ptr(), ldc=HIDDEN, M=T, N=cg, K=POS_K * cg, bias=self.pos_bias.data_ptr(),
                     R=hp.data_ptr(), ldr=HIDDEN, act=ops.ACT_GELU, prec=self.prec, batch=B * POS_G, z_inner=POS_G,
                     sA=(POS_G * Tp * cg, Tp * cg), sW=(0, 64 * POS_K * cg), sC=(T * HIDDEN, cg), sB=(0, cg),
                     sR=(T * HIDDEN, cg))
        return h

    # the pass in two stages, for callers that run the layers of different clips as separate launches / graphs
    # (host/pipeline.py capture_pipelined): front(x) -> h, then encoder_layers(h[a:b]) per group of clips
    def front(self, input_values, frame_num=None, cus=0):
        x = input_values.to(self.device, torch.float32).contiguous()
        feats = self.feature_extractor(x, cus)
        T = self.output_length(feats.shape[1], frame_num)
        h25 = ops.interp_layernorm(feats, T, self.fp_g, self.fp_b)
        return self.pos_embed(ops.linear(h25, self.proj, prec=self.prec))

    def encoder_layers(self, h, cus=0):
        """The 12 layers on the clips of ``h`` (B', T, 768) - rows of different clips never meet in them - as ONE chain on the
        current stream (no fan-out); overwrites ``h``."""
        return self._encoder_layers(h, cus, _inner=True)

    def _encoder_layers(self, h, cus=0, _inner=False):
        d = HIDDEN // HEADS
        if not self.use_planes_tf:
            h = ops.layernorm(h, self.enc_g, self.enc_b, out=h)
            for ly in self.layers:
                qkv = ops.linear(h, ly.qkv, prec=self.prec)
                att = ops.attention_d64(qkv, HEADS, d ** -0.5)
                h = ops.linear(att, ly.out, residual=h, prec=self.prec)
                h = ops.layernorm(h, *ly.ln1, out=h)
                f = ops.linear(h, ly.ff1, act=ops.ACT_GELU, prec=self.prec)
                h = ops.linear(f, ly.ff2, residual=h, prec=self.prec)
                h = ops.layernorm(h, *ly.ln2, out=h)
            return h
        # LayerNorm outputs feed a big GEMM (qkv / ffn1) AND the residual: emitted as split planes + fp32
        PA, PF = self.plan.attn, self.plan.ffn               # every producer writes the format its consumer reads
        FA, FF = ops.plane_fmt(PA), ops.plane_fmt(PF)
        # stream-K workspace of the 128-row GEMMs (csrc/gemm_pp192.hip): only when the caller says how many CUs are free -
        # M = 8000 on the 224 CUs beside the sampler leaves 16 % of the last round of qkv / out / ffn2 idle otherwise
        # ONE workspace per model and its contract is one launch at a time: the chains of clips (_inner) run concurrently on
        # their own streams and inside their own graph captures, so only the single-chain path may use it
        ws = None
        if cus and self.stream_k and not _inner and self.split_streams <= 1:
            M = h.shape[0] * h.shape[1]
            if self._sk_ws is None or self._sk_ws[0] != M:
                self._sk_ws = (M, ops.stream_k_workspace(M, 3 * HIDDEN, self.device))
            ws = self._sk_ws[1]
        if (self.split_streams > 1 and not _inner and h.shape[0] % self.split_streams == 0
                and h.shape[0] * h.shape[1] >= self.split_streams * SPLIT_MIN_ROWS):
            return self._encoder_layers_split(h, cus)
        h, hp_ = ops.layernorm_planes(h, self.enc_g, self.enc_b, out=h, fmt=FA)
        d = HIDDEN // HEADS
        for ly in self.layers:   # every projection on the ping-pong GEMMs, every activation split once
            qkv = ops.linear_planes(hp_, ly.qkv, prec=PA, cus=cus, sk_ws=ws)          # (B,T,2304) fp32
            att = ops.attention_d64_planes(qkv, HEADS, d ** -0.5, fmt=FA)            # planes
            h = ops.linear_planes(att, ly.out, residual=h, prec=PA, cus=cus, sk_ws=ws)
            h, hp_ = ops.layernorm_planes(h, *ly.ln1, out=h, fmt=FF)
            f = ops.linear_planes(hp_, ly.ff1, act=ops.ACT_GELU, prec=PF, out_planes=True, cus=cus)
            h = ops.linear_planes(f, ly.ff2, residual=h, prec=PF, cus=cus, sk_ws=ws)
            h, hp_ = ops.layernorm_planes(h, *ly.ln2, out=h, fmt=FA)
        return h

    def _encoder_layers_split(self, h, cus):
        """The 12 layers as ``split_streams`` independent chains of clips, one stream each: the fixed part of every launch
        (pipeline fill, the epilogue's burst of stores when all tiles of a one-round GEMM end together, the launch gap)
        of one chain overlaps the matrix loop of another.  Same kernels, same per-row arithmetic: bit-identical.
        Memory: the result is allocated on the LAUNCH stream before the fork (its consumers run there; a block owned by a
        side stream would go back to that stream's pool while they still read it) and the side streams' use of it and of
        the input is recorded; everything a chain allocates lives and dies on its own stream."""
        n = self.split_streams
        cur = torch.cuda.current_stream(self.device)
        while len(self._split_streams) < n - 1:
            self._split_streams.append(torch.cuda.Stream(self.device))
        per = h.shape[0] // n
        out = torch.empty_like(h)
        for i in list(range(1, n)) + [0]:                # chain 0 last, on the launch stream
            st = cur if i == 0 else self._split_streams[i - 1]
            if i:
                st.wait_stream(cur)
                h.record_stream(st)
                out.record_stream(st)
            with torch.cuda.stream(st):
                out[i * per:(i + 1) * per].copy_(self._encoder_layers(h[i * per:(i + 1) * per], cus, _inner=True))
        for i in range(1, n):
            cur.wait_stream(self._split_streams[i - 1])
        return out

    def forward(self, input_values, dataset="vocaset", attention_mask=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, frame_num=None, cus=0):
        """models/lib/wav2vec.py:80-156.  ``attention_mask`` (padding) is not supported on the
        hot path (the reference callers never pass one: models/faceformer.py:330,673).  ``cus`` (not in the reference
        signature): compute units the big GEMMs may count on, 0 = the whole chip (ops.gemm_raw)."""
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is not used on the reference hot path")
        if input_values.dim() != 2:
            raise ValueError("input_values must be (B, N)")
        x = input_values.to(self.device, torch.float32).contiguous()
        feats = self.feature_extractor(x, cus)
        T = self.output_length(feats.shape[1], frame_num)
        h25 = ops.interp_layernorm(feats, T, self.fp_g, self.fp_b)
        hp = ops.linear(h25, self.proj, prec=self.prec)
        h = self.encoder(hp, cus)
        out = SimpleNamespace(last_hidden_state=h, hidden_states=None, attentions=None,
                              extract_features=feats)
        return out if (return_dict is None or return_dict) else (h,)

    __call__ = forward
