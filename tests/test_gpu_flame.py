"""FLAME vertices on the GPU against the reference's own lbs() (golden) and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def basis():
    from avi_talking_amd.weights import make_flame_basis
    return make_flame_basis(4)


def test_flame_forward_matches_reference_lbs_golden(gpu, basis):
    from avi_talking_amd.host.flame import FLAME
    g = np.load(os.path.join(G, "flame.npz"))
    t = lambda k: torch.from_numpy(g[k]).to(gpu)
    fl = FLAME(basis, device=gpu)
    v, _, _ = fl(t("shape"), t("exp"), t("pose"), eye_pose_params=t("eye"), neck_pose=t("neck"))
    assert v.shape == (6, 5023, 3)
    err = (v[:, torch.from_numpy(g["vidx"]).to(gpu)].cpu() - torch.from_numpy(g["verts"])).abs().max().item()
    print(f"FLAME vs reference lbs(): {err:.2e}")
    assert err < 2e-6
    assert np.abs(v.double().sum((1, 2)).cpu().numpy() - g["vsum"]).max() < 1e-3


@pytest.mark.parametrize("B,T", [(1, 1), (2, 9), (3, 250), (5, 17)])
def test_flame_clip_path_matches_oracle(gpu, basis, B, T):
    """One shape per clip, per-frame expression + jaw (+ small global / neck / eye rotations), ragged frame groups."""
    from avi_talking_amd.host.flame import FLAME
    from oracle import flame as OF
    g = torch.Generator().manual_seed(100 + B * 1000 + T)
    shape = torch.randn(B, 300, generator=g)
    exp = torch.randn(B, T, 50, generator=g) * 0.8
    pose = torch.randn(B, T, 15, generator=g) * 0.12
    pose[0, 0] = 0.0
    fl = FLAME(basis, device=gpu)
    out = fl.vertices(shape.to(gpu), exp.to(gpu), pose.to(gpu)).cpu()
    sel = [(0, 0), (B - 1, T - 1), (B // 2, T // 2), (0, T - 1)]
    for b, t in sel:
        betas = torch.cat([shape[b], exp[b, t]])[None]
        ref = OF.lbs(betas, pose[b, t][None], basis)[0][0]
        err = (out[b, t] - ref).abs().max().item()
        assert err < 2e-6, (b, t, err)


def test_flame_from_coefficients_and_full_size_properties(gpu, basis):
    """config[1] size (32 clips x 250 frames): rest pose + zero expression reproduces the shaped template in every
    frame, vertices are affine in the expression at rest pose, and frames are independent of their neighbours."""
    from avi_talking_amd.host.flame import FLAME
    fl = FLAME(basis, device=gpu)
    B, T = 32, 250
    g = torch.Generator().manual_seed(5)
    shape = torch.randn(B, 300, generator=g).to(gpu)
    exp = (torch.randn(B, T, 50, generator=g) * 0.8).to(gpu)
    jaw = (torch.randn(B, T, 3, generator=g) * torch.tensor([0.2, 0.03, 0.03])).to(gpu)
    zero = torch.zeros_like(jaw)
    v0 = fl.from_coefficients(shape, torch.zeros_like(exp), zero)                 # (B, T, V*3)
    sd = basis["shapedirs"].reshape(-1, 350)[:, :300].to(gpu)
    vs = basis["v_template"].reshape(-1).to(gpu)[None] + shape @ sd.t()
    assert (v0 - vs[:, None]).abs().max().item() < 2e-6
    va = fl.from_coefficients(shape, exp, zero)
    vb = fl.from_coefficients(shape, 2 * exp, zero)
    assert ((vb - v0) - 2 * (va - v0)).abs().max().item() < 5e-6                  # affine in exp at rest pose
    full = fl.from_coefficients(shape, exp, jaw)
    assert torch.isfinite(full).all()
    sub = fl.from_coefficients(shape[7:9], exp[7:9, 100:103], jaw[7:9, 100:103])
    assert torch.equal(sub, full[7:9, 100:103])                                   # frames independent, bit-exact


@pytest.mark.parametrize("n_exp,V,matrix_cores", [(50, 5023, False), (100, 333, True), (70, 333, False),
                                                  (124, 130, True), (20, 47, True)])
def test_flame_kernel_variants_match_oracle(gpu, n_exp, V, matrix_cores):
    """The fp32 vector-pipe kernel, the 160-wide matrix-core instantiation (n_exp = 100 -> 136 basis vectors, and its
    limit 124), the vector-pipe limit (70) and tiny ragged vertex counts."""
    from avi_talking_amd.host.flame import FLAME
    from avi_talking_amd.weights import make_flame_basis
    from oracle import flame as OF
    basis = make_flame_basis(9, n_vertices=V, n_exp=n_exp)
    B, T = 3, 21
    g = torch.Generator().manual_seed(n_exp + V)
    shape = torch.randn(B, 300, generator=g)
    exp = torch.randn(B, T, n_exp, generator=g) * 0.8
    pose = torch.randn(B, T, 15, generator=g) * 0.12
    fl = FLAME(basis, n_exp=n_exp, device=gpu, matrix_cores=matrix_cores)
    assert (fl.fb.basis_hi is not None) == matrix_cores
    out = fl.vertices(shape.to(gpu), exp.to(gpu), pose.to(gpu)).cpu()
    assert out.shape == (B, T, V, 3)
    for b, t in [(0, 0), (B - 1, T - 1), (1, 16), (2, 15)]:
        betas = torch.cat([shape[b], exp[b, t]])[None]
        ref = OF.lbs(betas, pose[b, t][None], basis)[0][0]
        assert (out[b, t] - ref).abs().max().item() < 2e-6, (b, t)


def test_flame_rejects_bases_wider_than_the_kernels(gpu):
    from avi_talking_amd.host.flame import FLAME
    from avi_talking_amd.weights import make_flame_basis
    with pytest.raises(ValueError):
        FLAME(make_flame_basis(9, n_vertices=64, n_exp=125), n_exp=125, device=gpu)
    with pytest.raises(ValueError):
        FLAME(make_flame_basis(9, n_vertices=64, n_exp=71), n_exp=71, device=gpu, matrix_cores=False)
