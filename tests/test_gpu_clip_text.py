"""CLIP text encoder (FrozenCLIPEmbedder, models/diffusion_prior.py:29-55) on the GPU against the
transformers.CLIPTextModel golden and the CPU oracle.  Tolerance: 1e-3 max-abs on last_hidden_state (unit-variance
LayerNorm outputs), the same bar as the coefficients of the main path; observed errors are printed."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-3


@pytest.fixture(scope="module")
def clip_w():
    from avi_talking_amd.weights import make_clip_text_weights
    return make_clip_text_weights(5)


@pytest.fixture(scope="module")
def embedder(gpu, clip_w):
    from avi_talking_amd.host.clip_text import FrozenCLIPEmbedder
    return FrozenCLIPEmbedder.from_state_dict(clip_w, device=gpu)


def test_clip_text_matches_transformers_golden(gpu, embedder):
    g = np.load(os.path.join(G, "clip_text.npz"))
    ids = torch.from_numpy(g["ids"])
    out = embedder(ids)
    assert list(out.shape) == list(g["out_shape"])
    err = np.abs(out[:, :, ::8].cpu().numpy() - g["out_slice"]).max()
    print(f"CLIP text vs transformers golden: {err:.2e}")
    assert err < TOL
    short = embedder.encode(ids[:1, :20].contiguous())
    assert np.abs(short[0, :, ::8].cpu().numpy() - g["short_slice"]).max() < TOL


@pytest.mark.parametrize("B,T", [(1, 77), (5, 77), (2, 1), (3, 50), (32, 77)])
def test_clip_text_matches_oracle(gpu, embedder, clip_w, B, T):
    from oracle import clip_text as OC
    ids = torch.randint(0, 49408, (B, T), generator=torch.Generator().manual_seed(7 * B + T))
    ref = OC.clip_text_forward(clip_w, ids)
    out = embedder(ids.to(gpu))                 # device-resident ids take the same path
    err = (out.cpu() - ref).abs().max().item()
    print(f"CLIP text B={B} T={T}: {err:.2e}")
    assert err < TOL


def test_clip_text_plane_operand_path(gpu, clip_w, monkeypatch):
    """Above ~3000 rows the layers run on the plane-operand ping-pong GEMMs (forced here at 5 x 77 rows): same parity
    bar, and agreement with the few-rows path."""
    from avi_talking_amd.host.clip_text import FrozenCLIPEmbedder
    from oracle import clip_text as OC
    monkeypatch.setenv("AVI_CLIP_SMALL_ROWS", "0")
    big = FrozenCLIPEmbedder.from_state_dict(clip_w, device=gpu)
    monkeypatch.setenv("AVI_CLIP_SMALL_ROWS", "100000")
    small = FrozenCLIPEmbedder.from_state_dict(clip_w, device=gpu)
    ids = torch.randint(0, 49408, (5, 77), generator=torch.Generator().manual_seed(21))
    ref = OC.clip_text_forward(clip_w, ids)
    a, b = big(ids), small(ids)
    assert (a.cpu() - ref).abs().max().item() < TOL and (b.cpu() - ref).abs().max().item() < TOL
    assert (a - b).abs().max().item() < 2e-4


def test_clip_text_is_causal_and_batch_independent(gpu, embedder):
    """Position t depends on tokens <= t only, and a row's result does not depend on its batch mates (bit-exact)."""
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 49408, (4, 77), generator=g)
    base = embedder(ids)
    changed = ids.clone()
    changed[:, 40:] = torch.randint(0, 49408, (4, 37), generator=g)
    out2 = embedder(changed)
    assert torch.equal(base[:, :40], out2[:, :40])
    assert not torch.equal(base[:, 40:], out2[:, 40:])
    single = embedder(ids[2:3].contiguous())
    assert (single[0] - base[2]).abs().max().item() < 1e-5


def test_clip_text_voxel_and_graph_replay(gpu, embedder, clip_w):
    """The pooled (B,768) text feature, and a captured forward replayed on new ids (bit-equal to the eager pass)."""
    from oracle import clip_text as OC
    g = torch.Generator().manual_seed(17)
    ids = torch.randint(0, 49408, (4, 77), generator=g)
    ref = OC.clip_text_forward(clip_w, ids).mean(dim=1)             # train_diffusion_prior.py:438-439
    vox = embedder.voxel(ids)
    assert vox.shape == (4, 768)
    assert (vox.cpu() - ref).abs().max().item() < TOL
    embedder.capture(torch.zeros((4, 77), dtype=torch.int64))
    h, v = embedder.replay(ids)
    assert torch.equal(h, embedder(ids)) and torch.equal(v, vox)
    ids2 = torch.randint(0, 49408, (4, 77), generator=g)
    h2, _ = embedder.replay(ids2)
    assert torch.equal(h2, embedder(ids2))
    with pytest.raises(ValueError):
        embedder.replay(ids[:2])


def test_clip_text_rejects_bad_input(gpu, embedder):
    with pytest.raises(IndexError):
        embedder(torch.full((1, 77), 49408, dtype=torch.int64))
    with pytest.raises(ValueError):
        embedder(torch.zeros((1, 78), dtype=torch.int64))
    with pytest.raises(ValueError):
        embedder(torch.zeros((1, 77), dtype=torch.int32))
    with pytest.raises(TypeError):
        embedder(["a prompt"])


def test_quick_gelu_epilogue(gpu):
    """AVI_ACT_QUICK_GELU in the GEMM epilogue against x*sigmoid(1.702x)."""
    from avi_talking_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(200, 128, generator=g)
    w = torch.randn(192, 128, generator=g) / 8
    b = torch.randn(192, generator=g)
    ref = x @ w.t() + b
    ref = ref * torch.sigmoid(1.702 * ref)
    out = ops.linear(x.to(gpu), ops.PackedWeight(w.to(gpu), b.to(gpu)), act=ops.ACT_QUICK_GELU)
    assert (out.cpu() - ref).abs().max().item() < 1e-4   # 3-term bf16 products at |x| ~ 5
