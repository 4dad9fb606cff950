"""Host mirror of the reference's frozen CLIP text embedder (the text branch of the prior).

Mirrors ``FrozenCLIPEmbedder`` (reference models/diffusion_prior.py:29-55): ``forward`` returns
``CLIPTextModel(...).last_hidden_state`` for tokens padded to ``max_length`` 77.  The tokenizer (``:49-51``) is host
string work and stays with the caller's ``CLIPTokenizer``: ``forward`` takes what that call returns
(``batch_encoding["input_ids"]``), or a callable ``tokenizer`` may be attached to accept strings as the reference does.

Device work per layer (pre-LN CLIPEncoderLayer): LayerNorm emitting split bf16 planes -> fused q|k|v projection on
the ping-pong GEMM -> causal head-dim-64 attention on the matrix cores -> out_proj (+residual) -> LayerNorm planes ->
fc1 + quick_gelu (planes out) -> fc2 (+residual).  Nothing here falls back to the CPU.
"""
import os
from types import SimpleNamespace

import torch

from .. import ops

MAX_LENGTH = 77


class FrozenCLIPEmbedder:
    def __init__(self, state_dict, heads=12, eps=1e-5, device="cuda", max_length=MAX_LENGTH, tokenizer=None):
        self.device = torch.device(device)
        self.heads = heads
        self.eps = eps
        self.max_length = max_length
        self.tokenizer = tokenizer
        # rows (prompts x tokens) up to which the fp32-operand GEMM with 64 x 64 tiles serves every projection; above it the
        # plane-operand path, whose narrow projections (out_proj, fc2: N = 768) are cut along K while their 128-row tiles
        # would not fill the chip (split_rows)
        # MEASURED (round 4, one hipGraph per forward): 32 prompts = 2464 rows: fp32-operand path 2.69 ms (155 TFLOP/s), plane
        # path 3.32 ms, plane path with K slices 2.73 ms; 64 prompts: plane path 4.30 ms, with K slices 4.19 ms (200 TFLOP/s)
        self.small_rows = int(os.environ.get("AVI_CLIP_SMALL_ROWS", "3072"))
        self.split_rows = int(os.environ.get("AVI_CLIP_SPLIT_ROWS", "6144"))
        # checkpoints of CLIPTextModel carry a "text_model." prefix; newer transformers state_dicts drop it
        w = {(k[len("text_model."):] if k.startswith("text_model.") else k):
             v.detach().to(self.device, torch.float32).contiguous()
             for k, v in state_dict.items() if v.is_floating_point()}
        self.tok = w["embeddings.token_embedding.weight"]
        self.pos = w["embeddings.position_embedding.weight"]
        self.hidden = self.tok.shape[1]
        if self.hidden % heads or self.hidden // heads != 64:
            raise ValueError("FrozenCLIPEmbedder: the attention kernel is built for head dim 64")
        self.final = (w["final_layer_norm.weight"], w["final_layer_norm.bias"])
        self.zero_slopes = torch.zeros(heads, dtype=torch.float32, device=self.device)   # causal mask, no ALiBi term
        self.layers = []
        i = 0
        while f"encoder.layers.{i}.layer_norm1.weight" in w:
            p = f"encoder.layers.{i}."
            a = p + "self_attn."
            qkv_w = torch.cat([w[a + "q_proj.weight"], w[a + "k_proj.weight"], w[a + "v_proj.weight"]], 0)
            qkv_b = torch.cat([w[a + "q_proj.bias"], w[a + "k_proj.bias"], w[a + "v_proj.bias"]], 0)
            self.layers.append(SimpleNamespace(
                ln1=(w[p + "layer_norm1.weight"], w[p + "layer_norm1.bias"]),
                qkv=ops.PackedWeight(qkv_w, qkv_b),
                out=ops.PackedWeight(w[a + "out_proj.weight"], w[a + "out_proj.bias"]),
                ln2=(w[p + "layer_norm2.weight"], w[p + "layer_norm2.bias"]),
                fc1=ops.PackedWeight(w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"]),
                fc2=ops.PackedWeight(w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])))
            i += 1
        if not self.layers:
            raise ValueError("FrozenCLIPEmbedder: no encoder layers in the state_dict")

    @classmethod
    def from_state_dict(cls, state_dict, **kw):
        return cls(state_dict, **kw)

    def eval(self):
        return self

    def freeze(self):          # models/diffusion_prior.py:43-46: nothing here holds gradients
        return self

    def encode_ids(self, input_ids):
        """(B, T<=77) int64 token ids -> last_hidden_state (B, T, hidden)."""
        if input_ids.dim() != 2 or input_ids.dtype != torch.int64:
            raise ValueError("input_ids must be a (B, T) int64 tensor")
        if input_ids.shape[1] > self.pos.shape[0]:
            raise ValueError(f"sequence length {input_ids.shape[1]} exceeds the {self.pos.shape[0]} positions")
        if not input_ids.is_cuda:    # tokenizer output lives on the host: range-check it there, as nn.Embedding would
            if input_ids.numel() and (int(input_ids.min()) < 0 or int(input_ids.max()) >= self.tok.shape[0]):
                raise IndexError("index out of range in self")
        ids = input_ids.to(self.device).contiguous()
        B, T = ids.shape
        C, H = self.hidden, self.heads
        h = ops.embed_tokens(ids, self.tok, self.pos)
        if B * T <= self.small_rows:
            # few rows: the 128-row plane-operand tiles would cover a quarter of the chip (60-80 tiles for the 768-wide
            # projections at 32 prompts); the fp32-operand GEMM has 64 x 64 tiles for such grids
            for ly in self.layers:
                x = ops.layernorm(h, *ly.ln1, eps=self.eps)
                qkv = ops.linear(x, ly.qkv)
                att = ops.attention(qkv, qkv[..., C:], qkv[..., 2 * C:], H, 64, 3 * C, 3 * C, T, T, B, 64 ** -0.5,
                                    bias_mode=2, slopes=self.zero_slopes, period=1)
                h = ops.linear(att, ly.out, residual=h)
                x = ops.layernorm(h, *ly.ln2, eps=self.eps)
                f = ops.linear(x, ly.fc1, act=ops.ACT_QUICK_GELU)
                h = ops.linear(f, ly.fc2, residual=h)
            return ops.layernorm(h, *self.final, eps=self.eps)
        # A batch of 32 prompts is 2464 rows: 20 row tiles of 128.  q|k|v (N = 2304) and fc1 (N = 3072) make 240 tiles of
        # 128 x 192 / 128 x 256 - one round of the chip - but out_proj and fc2 (N = 768) only 60-80; cut along K (2 and 4
        # slices) they launch 160 and 240 workgroups, the partial sums folded with bias and residual by one small launch
        split = B * T <= self.split_rows and C % (2 * 96) == 0
        for ly in self.layers:
            _, xp = ops.layernorm_planes(h, *ly.ln1, eps=self.eps, want_f32=False)
            qkv = ops.linear_planes(xp, ly.qkv)                                          # (B, T, 3C) fp32
            att = ops.attention_d64_planes(qkv, H, 64 ** -0.5, bias_mode=2, slopes=self.zero_slopes, period=1)
            h = ops.linear_planes_splitk(att, ly.out, 2, residual=h) if split else ops.linear_planes(att, ly.out, residual=h)
            _, xp = ops.layernorm_planes(h, *ly.ln2, eps=self.eps, want_f32=False)
            f = ops.linear_planes(xp, ly.fc1, act=ops.ACT_QUICK_GELU, out_planes=True)
            h = (ops.linear_planes_splitk(f, ly.fc2, 4, residual=h) if split and ly.fc2.K % (4 * 96) == 0
                 else ops.linear_planes(f, ly.fc2, residual=h))
        return ops.layernorm(h, *self.final, eps=self.eps)

    def forward(self, text):
        """models/diffusion_prior.py:48-53.  ``text``: token ids, or strings when a tokenizer was attached."""
        if not torch.is_tensor(text):
            if self.tokenizer is None:
                raise TypeError("FrozenCLIPEmbedder.forward takes input_ids (attach tokenizer= to pass strings)")
            enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                                 return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
            text = enc["input_ids"]
        return self.encode_ids(text)

    __call__ = forward

    def encode(self, text):    # models/diffusion_prior.py:54-55
        return self(text)

    def voxel(self, text):
        """``clip_extractor(text).mean(dim=1)`` (train_diffusion_prior.py:438-439,710-711): the (B, hidden) text
        feature the aligner (``voxel2clip``) consumes."""
        return ops.mean_tokens(self(text))

    # ---- hipGraph of one forward at a fixed (B, T): ~90 launches replayed as one submission
    def capture(self, input_ids, warmup=2):
        if input_ids.dtype != torch.int64 or input_ids.dim() != 2:
            raise ValueError("input_ids must be a (B, T) int64 tensor")
        self._ids = input_ids.to(self.device).contiguous().clone()
        for _ in range(warmup):
            self.encode_ids(self._ids)
        torch.cuda.synchronize(self.device)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            h = self.encode_ids(self._ids)
            self._out = (h, ops.mean_tokens(h))
        return self

    def replay(self, input_ids=None):
        """-> (last_hidden_state, voxel) of the captured shape; ids on the host are range-checked as in forward."""
        if getattr(self, "_graph", None) is None:
            raise RuntimeError("capture() first")
        if input_ids is not None:
            if tuple(input_ids.shape) != tuple(self._ids.shape) or input_ids.dtype != torch.int64:
                raise ValueError("replay: input_ids must match the captured shape and dtype")
            if not input_ids.is_cuda and input_ids.numel() and (
                    int(input_ids.min()) < 0 or int(input_ids.max()) >= self.tok.shape[0]):
                raise IndexError("index out of range in self")
            self._ids.copy_(input_ids, non_blocking=True)
        self._graph.replay()
        return self._out
