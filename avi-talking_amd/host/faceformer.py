"""Host-side mirror of the reference ``models/faceformer.py`` ``Faceformer`` (decoder side).

``predict(audio, head_img, eye_img, emotion_img, text=None)`` keeps the reference signature
(models/faceformer.py:670) and returns un-normalised coefficients ``(B, T, 53)``.  The FAN image
encoder that turns the three image arguments into per-frame embeddings
(third_party/pd_fgc_inference, models/faceformer.py:677-697) is out of scope; callers
that have those embeddings pass them as ``cond_embeds=(eye (B,T,6), emo (B,T,30), head (B,T,6))`` and the
``v_merge2hidden`` Linear (:185,707-708) is applied; without them the memory is
``audio_feature_map(wav2vec2(audio))`` as in upstream FaceFormer.

Unlike the reference (batch forced to 1 by its 3-D mask, prefix re-decoded every step) the HIP path
decodes B utterances at once with a KV cache.  Two device paths behind ``decode``:
  * narrow decoders (D < 256): all T steps in ONE launch, one workgroup per utterance (csrc/faceformer.hip);
  * wide decoders (D >= 256, e.g. feature_dim 1024 of config/vocaset/demo.yaml): a chain of 6-7 small launches per
    frame, each spread over the whole chip (csrc/faceformer_steps.hip), captured once per (B, T, chunk) in a hipGraph
    and replayed;
  * wide decoders, ONE utterance (D in 256 / 512 / 1024): ONE persistent launch of 256 workgroups that keep their
    rows of every matrix in LDS and exchange the frame's vectors as tagged granules (csrc/faceformer_persist.hip); the
    launch chain is its fallback (``decode_checked``) and the path for batches (a second row costs the persistent kernel
    more than the chain charges for it: its stages run the rows one after the other).
Long-form (T > 600, which the reference's tables do not reach: models/faceformer.py:88,147): ``decode(..., chunk=C)``
selects the chunked-causal window defined in include/avi_talking.h (avi_faceformer_decode_chunked).
"""
import ctypes as C
import math
import os

import torch

from .. import lib as L
from .. import ops
from .wav2vec import Wav2Vec2Model

NHEAD = 4


def alibi_slopes(n):
    """models/faceformer.py:52-62."""
    def pow2(n):
        start = 2 ** (-2 ** -(math.log2(n) - 3))
        return [start * start ** i for i in range(n)]
    if math.log2(n).is_integer():
        return pow2(n)
    c = 2 ** math.floor(math.log2(n))
    return pow2(c) + alibi_slopes(2 * c)[0::2][:n - c]


def ppe_period(d_model, period):
    """One period of PeriodicPositionalEncoding (models/faceformer.py:87-99)."""
    pe = torch.zeros(period, d_model)
    position = torch.arange(0, period, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def _pad_k(w2d, mult=64):
    N, K = w2d.shape
    Kp = (K + mult - 1) // mult * mult
    if Kp == K:
        return w2d.contiguous()
    out = torch.zeros((N, Kp), dtype=w2d.dtype, device=w2d.device)
    out[:, :K] = w2d
    return out


class Faceformer:
    def __init__(self, state_dict, audio_state_dict=None, period=30, device="cuda", prec=None,
                 coeff_mean=None, coeff_std=None, max_seq_len=600):
        self.device = torch.device(device)
        prec = ops.prec_plan(prec)                      # None = ops.DEFAULT_PREC (the audio encoder's plane-operand GEMMs)
        self.prec = ops.prec_plan(prec).small           # the decoder's own GEMMs are fp32-operand launches
        self.period = period
        self.max_seq_len = max_seq_len          # reference mask / PPE tables stop at 600 frames (:88,147)
        w = {k: v.detach().to(torch.float32) for k, v in state_dict.items()}
        self.D = w["audio_feature_map.weight"].shape[0]
        self.V = w["vertice_map_r.weight"].shape[0]
        self.audio_encoder = (Wav2Vec2Model(audio_state_dict, device=device, prec=prec)
                              if audio_state_dict is not None else None)
        dv = lambda t: t.to(self.device).contiguous()
        self.audio_feature_map = ops.PackedWeight(dv(w["audio_feature_map.weight"]), dv(w["audio_feature_map.bias"]))
        self.v_merge2hidden = None
        if "v_merge2hidden.weight" in w:
            self.v_merge2hidden = ops.PackedWeight(_pad_k(dv(w["v_merge2hidden.weight"])),
                                                   dv(w["v_merge2hidden.bias"]))
            self.merge_in = w["v_merge2hidden.weight"].shape[1]
        p = "transformer_decoder.layers.0."
        D = self.D
        Wc, bc = w[p + "multihead_attn.in_proj_weight"], w[p + "multihead_attn.in_proj_bias"]
        self.cross_v = ops.PackedWeight(dv(Wc[2 * D:]), dv(bc[2 * D:]))
        self.cross_o = ops.PackedWeight(dv(w[p + "multihead_attn.out_proj.weight"]),
                                        dv(w[p + "multihead_attn.out_proj.bias"]))
        self._w, self._p, self._tf = w, p, None          # the teacher-forced pass packs its matrices on first use
        self._keep = []

        def dev(t):
            t = dv(t)
            self._keep.append(t)
            return t.data_ptr()

        cw = L.AviFaceformerWeights()
        cw.D, cw.V, cw.period = D, self.V, period
        cw.wqkv, cw.bqkv = dev(w[p + "self_attn.in_proj_weight"].t()), dev(w[p + "self_attn.in_proj_bias"])
        cw.wo, cw.bo = dev(w[p + "self_attn.out_proj.weight"].t()), dev(w[p + "self_attn.out_proj.bias"])
        for i in (1, 2, 3):
            setattr(cw, f"n{i}g", dev(w[p + f"norm{i}.weight"]))
            setattr(cw, f"n{i}b", dev(w[p + f"norm{i}.bias"]))
        cw.w1, cw.b1 = dev(w[p + "linear1.weight"].t()), dev(w[p + "linear1.bias"])
        cw.w2, cw.b2 = dev(w[p + "linear2.weight"].t()), dev(w[p + "linear2.bias"])
        cw.wr, cw.br = dev(w["vertice_map_r.weight"].t()), dev(w["vertice_map_r.bias"])
        cw.wm, cw.bm = dev(w["vertice_map.weight"].t()), dev(w["vertice_map.bias"])
        cw.pe = dev(ppe_period(D, period))
        cw.slopes = dev(torch.tensor(alibi_slopes(NHEAD), dtype=torch.float32))
        cw.obj_embedding = dev(w["obj_embedding"].reshape(-1))
        if coeff_mean is not None:
            cw.coeff_mean = dev(torch.as_tensor(coeff_mean, dtype=torch.float32).reshape(-1)[: self.V])
            cw.coeff_std = dev(torch.as_tensor(coeff_std, dtype=torch.float32).reshape(-1)[: self.V])
        self.cw = cw
        mode = os.environ.get("AVI_FF_STEPS", "auto")           # "0" / "1" force a path (tests), auto: by width
        dh = D // NHEAD
        can = D % 64 == 0 and D <= 1024 and dh in (16, 32, 64, 128, 256)
        self.use_steps = can and (mode == "1" or (mode == "auto" and D >= 256))
        self.planes = self._build_planes(w, p) if self.use_steps else None
        self._graphs = {}
        # persistent single-launch decode for small batches (csrc/faceformer_persist.hip): AVI_FF_PERSIST=0 switches it off
        self.use_persist = (self.use_steps and D in (256, 512, 1024) and os.environ.get("AVI_FF_PERSIST", "1") != "0"
                            and self.device.type == "cuda"
                            and torch.cuda.get_device_properties(self.device).multi_processor_count >= 256)
        self._persist = None
        self.last_fallback = self._last_path = None

    def _build_planes(self, w, p):
        """Derived constants of the launch-chain path (include/avi_talking.h AviFaceformerPlanes): fragment-major bf16
        hi/lo planes of the four streamed matrices and the fused input map of a frame's q/k/v:
        qkv_i = in_proj(vertice_map(o_{i-1}) + pe_i) = (W_in W_map) o_{i-1} + [W_in (b_map + pe_i) + b_in]
        (the products are formed in fp64 and rounded once to fp32)."""
        D, V, dvc = self.D, self.V, self.device

        def frag(mat):                                       # [N][K] fp32 -> hi, lo [N/16][K/32][4][16][8]
            pw = ops.PackedWeight(mat.to(dvc).contiguous())
            N, K = pw.hi.shape
            f = lambda t: t.view(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()
            hi, lo = f(pw.hi), f(pw.lo)
            self._keep += [hi, lo]
            return hi.data_ptr(), lo.data_ptr()

        def keep(t):
            t = t.to(torch.float32).to(dvc).contiguous()
            self._keep.append(t)
            return t.data_ptr()

        pl = L.AviFaceformerPlanes()
        wm_pad = torch.zeros(D, 64)
        wm_pad[:, :V] = w["vertice_map.weight"]
        # [out_proj | vertice_map]: s1 = self-attention + input embedding comes out of ONE accumulation over [att | o]
        pl.wo_hi, pl.wo_lo = frag(torch.cat([w[p + "self_attn.out_proj.weight"], wm_pad], 1))
        pl.w1_hi, pl.w1_lo = frag(w[p + "linear1.weight"])
        pl.w2_hi, pl.w2_lo = frag(w[p + "linear2.weight"])
        pl.wr_hi, pl.wr_lo = frag(w["vertice_map_r.weight"])                    # N = V -> padded to 64 rows of zeros
        Win, b_in = w[p + "self_attn.in_proj_weight"].double(), w[p + "self_attn.in_proj_bias"].double()
        Wm, bm = w["vertice_map.weight"].double(), w["vertice_map.bias"].double()   # (D, V), (D)
        pe = ppe_period(D, self.period).double()
        wf_t = torch.zeros(64, 3 * D, dtype=torch.float64)
        wf_t[:V] = (Win @ Wm).t()
        pl.wf_t = keep(wf_t)
        pl.bf = keep((bm[None] + pe) @ Win.t() + b_in[None])                     # (period, 3D)
        x0 = w["obj_embedding"].reshape(-1).double() + pe[0]
        pl.qkv0 = keep(Win @ x0 + b_in)
        pl.x0 = keep(x0)
        return pl

    def _chunk(self, T, chunk):
        if chunk is None:
            if T > self.max_seq_len:
                raise ValueError(f"T={T} exceeds the reference's mask/PPE length {self.max_seq_len}: pass chunk= to "
                                 "select the chunked-causal window (include/avi_talking.h)")
            return T
        chunk = int(chunk)
        if chunk <= 0 or (chunk < T and chunk % self.period):
            raise ValueError("chunk must be a positive multiple of the PPE period")
        return min(chunk, T)


    def decode_checked(self, hidden_states, chunk=None, out_dtype=torch.float32):
        """``decode`` with the persistent kernel's safety net: synchronises, and if a workgroup of the persistent launch
        timed out on an exchange (it did not get all 256 CUs: output NaN, ``status.EXCHANGE_TIMEOUT``) decodes again on the
        launch chain.  ``last_fallback`` says whether that happened."""
        from . import status
        self.last_fallback = self._last_path = None
        out = self.decode(hidden_states, chunk=chunk, out_dtype=out_dtype)
        if self._last_path != "persist" or torch.cuda.is_current_stream_capturing():
            return out                   # the other paths cannot time out (and a capture cannot be synchronised)
        torch.cuda.synchronize(self.device)
        status.words()
        if status._view[status.EXCHANGE_TIMEOUT]:
            status.clear_word(status.EXCHANGE_TIMEOUT)
            self.last_fallback = "launch chain"
            out = self.decode(hidden_states, chunk=chunk, out_dtype=out_dtype, no_persist=True)
        return out

    def decode(self, hidden_states, chunk=None, out_dtype=torch.float32, no_persist=False):
        """The autoregressive loop of ``predict`` (:710-729) for memory ``hidden_states`` (B,T,D).  ``chunk``: attention
        window for long-form decoding (None = the reference's full causal window, T <= 600).  ``out_dtype`` float16: the
        frame-writing kernel stores IEEE half (BASELINE.json configs[4] "fp16 coeffs"; the fed-back frame stays fp32)."""
        if out_dtype not in (torch.float32, torch.float16):
            raise ValueError("out_dtype must be float32 or float16")
        hs = hidden_states.to(self.device, torch.float32).contiguous()
        B, T, D = hs.shape
        if D != self.D:
            raise ValueError(f"hidden_states has D={D}, decoder has D={self.D}")
        chunk = self._chunk(T, chunk)
        cross = ops.linear(ops.linear(hs, self.cross_v, prec=self.prec), self.cross_o, prec=self.prec)
        if self.use_steps:
            if self.use_persist and not no_persist and B == 1 and 6 * T + 6 < 65535 and chunk <= 1024:
                return self._decode_persistent(cross, B, T, chunk, out_dtype=out_dtype)
            self._last_path = "steps"
            return self._decode_steps(cross, B, T, chunk, out_dtype=out_dtype)
        self._last_path = "single"
        kv = torch.empty((B, T, 2 * D), dtype=torch.float32, device=self.device)
        out = torch.empty((B, T, self.V), dtype=out_dtype, device=self.device)
        fn = L.load().avi_faceformer_decode_chunked_f16 if out_dtype == torch.float16 else L.load().avi_faceformer_decode_chunked
        L.check(fn(C.byref(self.cw), cross.data_ptr(), B, T, chunk, kv.data_ptr(), out.data_ptr(), L.stream_ptr()),
                "avi_faceformer_decode_chunked")
        return out

    def _decode_persistent(self, cross, B, T, chunk, out_dtype=torch.float32):
        """One utterance of a wide decoder: one persistent launch (+ the epoch bump), captured per (B, T, chunk)."""
        from . import status
        self._last_path = "persist"
        so = L.load()
        status.words()                                   # the kernel reports a timed-out exchange there
        if self._persist is None:
            nimg, nx = C.c_longlong(), C.c_longlong()
            L.check(so.avi_faceformer_persist_sizes(self.D, C.byref(nimg), C.byref(nx)), "avi_faceformer_persist_sizes")
            image = torch.empty(nimg.value, dtype=torch.float32, device=self.device)
            L.check(so.avi_faceformer_persist_pack(C.byref(self.cw), C.byref(self.planes), image.data_ptr(), L.stream_ptr()),
                    "avi_faceformer_persist_pack")
            self._persist = dict(image=image, xch=torch.zeros(nx.value // 8, dtype=torch.int64, device=self.device))
        P = self._persist
        key = ("persist", B, T, chunk, out_dtype)
        g = self._graphs.pop(key, None)
        if g is not None:
            self._graphs[key] = g
        if g is None:
            st = dict(cross=torch.empty((B, T, self.D), dtype=torch.float32, device=self.device),
                      kv=torch.empty((B, T, 2 * self.D), dtype=torch.float32, device=self.device),
                      out=torch.empty((B, T, self.V), dtype=out_dtype, device=self.device))
            half = out_dtype == torch.float16

            def launch():
                L.check(so.avi_faceformer_decode_persistent(C.byref(self.cw), C.byref(self.planes), P["image"].data_ptr(),
                                                            st["cross"].data_ptr(), B, T, chunk, st["kv"].data_ptr(),
                                                            P["xch"].data_ptr(), None if half else st["out"].data_ptr(),
                                                            st["out"].data_ptr() if half else None, L.stream_ptr()),
                        "avi_faceformer_decode_persistent")
            if torch.cuda.is_current_stream_capturing():
                st["cross"].copy_(cross)
                launch()
                return st["out"].clone()
            st["cross"].copy_(cross)
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                launch()
            if len(self._graphs) >= 4:
                self._graphs.pop(next(iter(self._graphs)))
            g = self._graphs[key] = (graph, st)
        graph, st = g
        st["cross"].copy_(cross)
        graph.replay()
        return st["out"].clone()

    def _decode_steps(self, cross, B, T, chunk, rows_per_call=32, out_dtype=torch.float32):
        """Wide decoders: the per-frame launch chain, one hipGraph per (rows, T, chunk) over static buffers, replayed
        for every block of up to 32 utterances."""
        out = torch.empty((B, T, self.V), dtype=out_dtype, device=self.device)
        so = L.load()
        entry = so.avi_faceformer_decode_steps_f16 if out_dtype == torch.float16 else so.avi_faceformer_decode_steps
        for b0 in range(0, B, rows_per_call):
            nb = min(rows_per_call, B - b0)
            key = (nb, T, chunk, out_dtype)
            g = self._graphs.pop(key, None)
            if g is not None:
                self._graphs[key] = g                # most recently used last
            if g is None:
                n = C.c_longlong()
                L.check(so.avi_faceformer_steps_work_floats(self.D, nb, C.byref(n)), "work size")
                st = dict(cross=torch.empty((nb, T, self.D), dtype=torch.float32, device=self.device),
                          kv=torch.empty((nb, T, 2 * self.D), dtype=torch.float32, device=self.device),
                          work=torch.zeros(n.value, dtype=torch.float32, device=self.device),
                          out=torch.empty((nb, T, self.V), dtype=out_dtype, device=self.device))

                def chain():
                    L.check(entry(C.byref(self.cw), C.byref(self.planes), st["cross"].data_ptr(),
                                                           nb, T, chunk, st["kv"].data_ptr(), st["work"].data_ptr(),
                                                           st["out"].data_ptr(), L.stream_ptr()),
                            "avi_faceformer_decode_steps")
                if torch.cuda.is_current_stream_capturing():   # already inside a caller's graph: just enqueue
                    st["cross"].copy_(cross[b0:b0 + nb])
                    chain()
                    out[b0:b0 + nb] = st["out"]
                    continue
                st["cross"].copy_(cross[b0:b0 + nb])
                torch.cuda.synchronize(self.device)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    chain()
                if len(self._graphs) >= 4:           # static buffers per shape: keep the four most recent shapes
                    self._graphs.pop(next(iter(self._graphs)))
                g = self._graphs[key] = (graph, st)
            graph, st = g
            st["cross"].copy_(cross[b0:b0 + nb])
            graph.replay()
            out[b0:b0 + nb] = st["out"]
        return out

    def _audio_memory(self, audio, cond_embeds=None, frame_num=None):
        """:673-674 (+ :707-708): audio -> wav2vec2 -> audio_feature_map (-> v_merge2hidden) = the decoder's memory."""
        if self.audio_encoder is None:
            raise RuntimeError("Faceformer was built without audio encoder weights")
        feats = self.audio_encoder(audio.to(self.device), "vocaset", frame_num=frame_num).last_hidden_state
        hs = ops.linear(feats, self.audio_feature_map, prec=self.prec)
        if cond_embeds is None:
            return hs
        if self.v_merge2hidden is None:
            raise RuntimeError("cond_embeds given but the state_dict has no v_merge2hidden")
        eye, emo, head = [t.to(self.device, torch.float32) for t in cond_embeds]
        B, T, _ = hs.shape
        cat = torch.zeros((B, T, self.v_merge2hidden.K), dtype=torch.float32, device=self.device)
        o = 0
        for part in (eye, emo, hs, head):                                              # :707
            cat[..., o:o + part.shape[-1]] = part
            o += part.shape[-1]
        if o != self.merge_in:
            raise ValueError("cond_embeds do not match v_merge2hidden's input width")
        return ops.linear(cat, self.v_merge2hidden, prec=self.prec)                    # :708

    @torch.no_grad()
    def forward_teacher_forced(self, hidden_states, coeff):
        """The teacher-forced decoder pass (models/faceformer.py:378-391) for memory ``hidden_states`` (B,T,D) and
        NORMALISED ground-truth coefficients ``coeff`` (B,T,>=V): shift-right -> ``vertice_map`` -> PPE -> one
        ``TransformerDecoderLayer`` (ALiBi-causal self-attention, diagonal cross-attention, ReLU FFN, post-LN) ->
        ``vertice_map_r``; returns the normalised prediction (B,T,V) for all frames at once.

        The reference runs it one utterance at a time (its 3-D float mask only fits batch 1, :376); here the batch is
        one pass: M = B*T rows through avi_gemm, avi_attention with bias mode 2 (the (4,T,T) mask evaluated
        analytically), and the diagonal memory mask turned into cross_i = out_proj(v_proj(memory_i))."""
        hs = hidden_states.to(self.device, torch.float32).contiguous()
        B, T, D = hs.shape
        if D != self.D:
            raise ValueError(f"hidden_states has D={D}, decoder has D={self.D}")
        if T > self.max_seq_len:
            raise ValueError(f"T={T} exceeds the reference's mask/PPE length {self.max_seq_len}")
        c = coeff.to(self.device, torch.float32)
        if c.dim() != 3 or c.shape[0] != B or c.shape[1] != T or c.shape[2] < self.V:
            raise ValueError(f"coeff must be (B={B}, T={T}, >={self.V}), got {tuple(c.shape)}")
        c = c[..., :self.V].contiguous()                                 # the models slice [:53] (:414)
        if self._tf is None:
            w, p = self._w, self._p
            dv = lambda t: t.to(self.device).contiguous()
            pw = lambda wk, bk: ops.PackedWeight(dv(w[wk]), dv(w[bk]))
            self._tf = dict(
                qkv=pw(p + "self_attn.in_proj_weight", p + "self_attn.in_proj_bias"),
                out=pw(p + "self_attn.out_proj.weight", p + "self_attn.out_proj.bias"),
                l1=pw(p + "linear1.weight", p + "linear1.bias"), l2=pw(p + "linear2.weight", p + "linear2.bias"),
                vr=pw("vertice_map_r.weight", "vertice_map_r.bias"),
                norms=[(dv(w[p + f"norm{i}.weight"]), dv(w[p + f"norm{i}.bias"])) for i in (1, 2, 3)],
                slopes=torch.tensor(alibi_slopes(NHEAD), dtype=torch.float32, device=self.device))
        tf, P = self._tf, self.prec
        x = torch.empty((B, T, D), dtype=torch.float32, device=self.device)
        L.check(L.load().avi_faceformer_tf_embed(C.byref(self.cw), c.data_ptr(), B, T, x.data_ptr(), L.stream_ptr()),
                "avi_faceformer_tf_embed")                                                       # :382-384
        dh = D // NHEAD
        qkv = ops.linear(x, tf["qkv"], prec=P)
        att = ops.attention(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], NHEAD, dh, 3 * D, 3 * D, T, T, B,
                            dh ** -0.5, bias_mode=2, slopes=tf["slopes"], period=self.period)   # tgt_mask (:385)
        x = ops.layernorm(ops.linear(att, tf["out"], residual=x, prec=P), *tf["norms"][0])
        cv = ops.linear(hs, self.cross_v, prec=P)                                                # memory_mask (:387)
        x = ops.layernorm(ops.linear(cv, self.cross_o, residual=x, prec=P), *tf["norms"][1])
        f = ops.linear(x, tf["l1"], act=ops.ACT_RELU, prec=P)
        x = ops.layernorm(ops.linear(f, tf["l2"], residual=x, prec=P), *tf["norms"][2])
        return ops.linear(x, tf["vr"], prec=P)                                                   # :391

    @torch.no_grad()
    def forward(self, audio, coeff, criterion=None, teacher_forcing=True, cond_embeds=None, lip_coeff_weight=1.0):
        """The coefficient term of ``Faceformer.forward`` (models/faceformer.py:316-415): memory from the audio
        (``frame_num`` = the coefficient length, :330), the teacher-forced pass (or the AR loop, :392-409), then
        ``mean(criterion(pred[..., :53], coeff[..., :53]) * lip_coeff_weight)`` (:413-415; ``criterion`` defaults to the
        element-wise squared error the reference passes, ``nn.MSELoss(reduction='none')``).  The render / landmark /
        emotion losses that follow in the reference need external renderers and are out of scope.
        Returns (loss, prediction)."""
        T = coeff.shape[1]
        hs = self._audio_memory(audio, cond_embeds, frame_num=T)
        c = coeff.to(self.device, torch.float32)[..., :self.V].contiguous()
        if teacher_forcing:
            pred = self.forward_teacher_forced(hs, c)
        else:
            if self.cw.coeff_mean:
                raise RuntimeError("the AR branch of forward() compares NORMALISED coefficients: build the decoder "
                                   "without coeff_mean/std")
            pred = self.decode_checked(hs)
        if criterion is not None:
            return torch.mean(criterion(pred, c) * lip_coeff_weight), pred
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        dpred = torch.empty_like(pred)                 # weight * dloss/dpred: what a trainer would start backward from
        L.check(L.load().avi_mse_loss(pred.data_ptr(), c.data_ptr(), pred.numel(), float(lip_coeff_weight),
                                      loss.data_ptr(), dpred.data_ptr(), L.stream_ptr()), "avi_mse_loss")
        return loss[0] * lip_coeff_weight, pred

    @torch.no_grad()
    def predict(self, audio, head_img=None, eye_img=None, emotion_img=None, text=None, cond_embeds=None, chunk=None):
        return self.decode_checked(self._audio_memory(audio, cond_embeds), chunk=chunk)
