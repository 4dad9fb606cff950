"""Oracle: FaceFormer-style autoregressive coefficient decoder (fp32, CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates reference ``models/faceformer.py``:
  * :51-72   init_biased_mask (causal ALiBi with period)        -> oracle.emote.faceformer_biased_mask
  * :75-83   enc_dec_mask ("vocaset": only the diagonal is visible)
  * :87-102  PeriodicPositionalEncoding (period 25 default, args.period in the ctor :145)
  * :138-158 ctor layers: audio_feature_map, vertice_map, nn.TransformerDecoderLayer(d, 4 heads,
             ff 2d, ReLU, post-LN, batch_first), vertice_map_r, obj_embedding
  * :378-391 teacher-forced pass;  :710-726 autoregressive ``predict`` loop;  :729 un-normalise.
``predict_as_written`` re-decodes the whole prefix every step exactly like the reference (O(T^2));
``predict_cached`` is the mathematically identical KV-cached form the HIP path implements (the
diagonal memory mask makes cross-attention at step i read memory row i only).
Pinned by tests/golden/faceformer_*.npz (reference module imported with stubs).
"""
import math

import torch
import torch.nn.functional as F

from .emote import faceformer_biased_mask, mha

NHEAD = 4


def enc_dec_mask(T, S):
    """models/faceformer.py:75-83, dataset == "vocaset": True = blocked, diagonal open."""
    mask = torch.ones(T, S)
    for i in range(min(T, S)):
        mask[i, i] = 0
    return mask == 1


def ppe_table(d_model, period, max_seq_len=600):
    """models/faceformer.py:87-99."""
    pe = torch.zeros(period, d_model)
    position = torch.arange(0, period, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0).repeat(1, (max_seq_len // period) + 1, 1)


def _ln(w, p, x):
    return F.layer_norm(x, (x.shape[-1],), w[p + ".weight"], w[p + ".bias"], 1e-5)


def decoder_layer(w, tgt, memory, tgt_mask, memory_mask):
    """torch.nn.TransformerDecoderLayer, norm_first=False, activation relu, eval mode."""
    p = "transformer_decoder.layers.0"
    x = _ln(w, p + ".norm1", tgt + mha(w, p + ".self_attn", tgt, tgt, NHEAD, tgt_mask))
    x = _ln(w, p + ".norm2", x + mha(w, p + ".multihead_attn", x, memory, NHEAD, memory_mask))
    f = F.linear(F.relu(F.linear(x, w[p + ".linear1.weight"], w[p + ".linear1.bias"])),
                 w[p + ".linear2.weight"], w[p + ".linear2.bias"])
    return _ln(w, p + ".norm3", x + f)


def predict_as_written(w, hidden_states, period, coeff_mean=None, coeff_std=None):
    """models/faceformer.py:710-729 given ``hidden_states`` (1,T,D) (= memory after audio_feature_map /
    v_merge2hidden): returns (1,T,53), un-normalised when mean/std are given."""
    T, D = hidden_states.shape[1], hidden_states.shape[2]
    pe = ppe_table(D, period)
    mask_full = faceformer_biased_mask(NHEAD, 600, period)
    vertice_emb = w["obj_embedding"].unsqueeze(1)                      # (1,1,D)
    for i in range(T):
        vertice_input = vertice_emb + pe[:, :vertice_emb.shape[1]]
        L = vertice_input.shape[1]
        tgt_mask = mask_full[:, :L, :L]
        memory_mask = enc_dec_mask(L, T)
        out = decoder_layer(w, vertice_input, hidden_states, tgt_mask, memory_mask)
        out = F.linear(out, w["vertice_map_r.weight"], w["vertice_map_r.bias"])
        new = F.linear(out[:, -1, :], w["vertice_map.weight"], w["vertice_map.bias"]).unsqueeze(1)
        vertice_emb = torch.cat((vertice_emb, new), 1)
    if coeff_mean is not None:
        out = out * coeff_std + coeff_mean
    return out


def predict_cached(w, hidden_states, period, coeff_mean=None, coeff_std=None, chunk=None):
    """Same function, O(T) decoder work: self-attention K/V appended per step; cross-attention reduces
    to out_proj(v_proj(memory[i])) because row i of the memory mask opens column i only.

    ``chunk`` (a multiple of ``period``; None = the reference's full causal window) is the long-form extension the
    reference does NOT have (its mask and PPE tables stop at 600 frames, models/faceformer.py:88,147, and predict()
    fails beyond them): frame i attends to frames floor(i/chunk)*chunk .. i only; ALiBi distance (i-j)//period,
    PPE phase i % period and the fed-back embedding vertice_map(previous output) are unchanged.  PARITY UNPINNED for
    chunk < T (there is no reference behaviour to pin it to); chunk >= T is pinned by tests/golden/faceformer_*.npz."""
    B, T, D = hidden_states.shape
    dh = D // NHEAD
    pe = ppe_table(D, period)[0]
    p = "transformer_decoder.layers.0"
    Wi, bi = w[p + ".self_attn.in_proj_weight"], w[p + ".self_attn.in_proj_bias"]
    Wc, bc = w[p + ".multihead_attn.in_proj_weight"], w[p + ".multihead_attn.in_proj_bias"]
    cross = F.linear(F.linear(hidden_states, Wc[2 * D:], bc[2 * D:]),
                     w[p + ".multihead_attn.out_proj.weight"], w[p + ".multihead_attn.out_proj.bias"])
    from .emote import get_slopes
    slopes = torch.tensor(get_slopes(NHEAD))
    Kc = torch.zeros(B, T, D)
    Vc = torch.zeros(B, T, D)
    emb = w["obj_embedding"].expand(B, D)
    outs = []
    for i in range(T):
        x = emb + pe[i % period]             # the table is periodic (rows beyond 600 + period do not exist in it)
        qkv = F.linear(x, Wi, bi)
        q, Kc[:, i], Vc[:, i] = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        qh = q.view(B, NHEAD, 1, dh)
        k0 = 0 if not chunk else (i // chunk) * chunk
        n = i + 1 - k0
        kh = Kc[:, k0:i + 1].view(B, n, NHEAD, dh).transpose(1, 2)
        vh = Vc[:, k0:i + 1].view(B, n, NHEAD, dh).transpose(1, 2)
        s = torch.matmul(qh, kh.transpose(2, 3)) * dh ** -0.5
        dist = (i - torch.arange(k0, i + 1)) // period
        s = s - slopes[None, :, None, None] * dist[None, None, None, :].float()
        a = torch.matmul(torch.softmax(s, -1), vh).reshape(B, D)
        x = _ln(w, p + ".norm1", x + F.linear(a, w[p + ".self_attn.out_proj.weight"], w[p + ".self_attn.out_proj.bias"]))
        x = _ln(w, p + ".norm2", x + cross[:, i])
        f = F.linear(F.relu(F.linear(x, w[p + ".linear1.weight"], w[p + ".linear1.bias"])),
                     w[p + ".linear2.weight"], w[p + ".linear2.bias"])
        x = _ln(w, p + ".norm3", x + f)
        o = F.linear(x, w["vertice_map_r.weight"], w["vertice_map_r.bias"])
        outs.append(o)
        emb = F.linear(o, w["vertice_map.weight"], w["vertice_map.bias"])
    out = torch.stack(outs, 1)
    if coeff_mean is not None:
        out = out * coeff_std + coeff_mean
    return out


def teacher_forced(w, hidden_states, coeff, period):
    """models/faceformer.py:378-391: shifted ground-truth coefficients in, all frames out at once."""
    D = hidden_states.shape[2]
    vin = torch.cat([torch.zeros_like(coeff[:, -1:]), coeff[:, :-1]], 1)
    vin = F.linear(vin, w["vertice_map.weight"], w["vertice_map.bias"])
    vin = vin + ppe_table(D, period)[:, :vin.shape[1]]
    L = vin.shape[1]
    tgt_mask = faceformer_biased_mask(NHEAD, 600, period)[:, :L, :L]
    out = decoder_layer(w, vin, hidden_states, tgt_mask, enc_dec_mask(L, hidden_states.shape[1]))
    return F.linear(out, w["vertice_map_r.weight"], w["vertice_map_r.bias"])
