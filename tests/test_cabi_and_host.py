"""CPU: the C-ABI library loads and exports every symbol include/avi_talking.h declares; host-side logic
(weight naming, length arithmetic, schedule/tables, sharding) agrees with the oracle.  No GPU compute."""
import importlib.util
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import avi_talking_amd.lib as L
    if not os.path.exists(L.LIB_PATH):
        spec = importlib.util.spec_from_file_location("avi_build", os.path.join(ROOT, "avi-talking_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build(verbose=False)
    return L


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "avi_talking.h")).read()
    declared = set(re.findall(r"\b(avi_[a-z0-9_]+)\s*\(", hdr))
    assert "avi_gemm" in declared and "avi_prior_sample" in declared and "avi_faceformer_decode" in declared
    so = lib.load()
    for name in declared:
        assert hasattr(so, name), f"{name} declared in the header but not exported"
    assert declared - {"avi_version"} == set(lib.SIGNATURES), "ctypes table and header disagree"
    assert lib.version().startswith("avi_talking_hip")


def test_struct_layouts_match_header(lib):
    import ctypes as C
    # AviGemm: 21 pointer/long long fields of 8 bytes + 7 ints, padded to 8
    assert C.sizeof(lib.AviGemm) == 21 * 8 + 8 * 4 + 4 * 8 + 8 + 8 + 16  # + ldw, cus (two ints) + C16 + sk_ws, sk_ws_floats
    assert C.sizeof(lib.AviPriorLayer) == 8 * 8
    assert C.sizeof(lib.AviPriorWeights) == 8 + 13 * 8 + 8 * 64 + 5 * 8
    assert C.sizeof(lib.AviFaceformerWeights) == 16 + 23 * 8
    assert C.sizeof(lib.AviFaceformerPlanes) == 12 * 8
    assert C.sizeof(lib.AviFlameBasis) == 7 * 8 + 3 * 4 + 4 + 2 * 8     # + basis_hi, basis_lo (after padding)
    assert C.sizeof(lib.AviTransposeJob) == 4 * 8 + 4 * 4 + 8           # in, out, hi, lo | R, C, C_pad, first_block | colsum


def test_no_cpu_fallback(lib):
    from avi_talking_amd import ops
    with pytest.raises(RuntimeError):
        ops.layernorm(torch.zeros(4, 64), torch.ones(64), torch.zeros(64))
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros(4, 64), None)


def test_bad_arguments_are_rejected_before_launch(lib):
    import ctypes as C
    so = lib.load()
    g = lib.AviGemm()
    assert so.avi_gemm(C.byref(g), None) == -1                      # null pointers
    g.A, g.Whi, g.C, g.M, g.N, g.K, g.batch, g.z_inner, g.prec = 16, 16, 16, 4, 4, 63, 1, 1, 1
    assert so.avi_gemm(C.byref(g), None) == -1                      # K not a multiple of 64
    assert so.avi_attention(None, None, None, None, 1, 1, 1, 1, 64, 64, 64, 64, 1.0, 0, None, 1, None) == -1
    assert so.avi_layernorm(None, 1, 64, None, None, 1e-5, None, None) == -1
    w = lib.AviPriorWeights()
    assert so.avi_prior_sample(C.byref(w), None, None, 1, 1.0, None, None, None) == -1


def test_weight_key_names_and_shapes():
    from avi_talking_amd import weights as W
    a = W.make_wav2vec2_weights(0)
    assert a["feature_extractor.conv_layers.0.conv.weight"].shape == (512, 1, 10)
    assert a["encoder.pos_conv_embed.conv.parametrizations.weight.original1"].shape == (768, 48, 128)
    assert sum(v.numel() for v in a.values()) == 94371712           # = HF Wav2Vec2Model(Wav2Vec2Config())
    p = W.make_prior_weights(3)
    assert p["net.causal_transformer.layers.5.0.to_kv.weight"].shape == (128, 128)
    assert p["net.causal_transformer.layers.0.1.1.weight"].shape == (1024, 128)
    assert p["net.causal_transformer.rel_pos_bias.relative_attention_bias.weight"].shape == (32, 8)
    nb = sum(v.numel() for k, v in p.items() if k.startswith("voxel2clip."))
    assert nb == 75571712                                           # BrainNetwork (75.57 M)
    again = W.make_prior_weights(3)
    assert all(torch.equal(p[k], again[k]) for k in p)              # seeded: identical on every machine


def test_length_arithmetic():
    from avi_talking_amd.host.wav2vec import conv_out_lengths
    from oracle import wav2vec2 as OW
    for n in (16000, 64000, 64080, 79872, 160000, 960000):
        assert conv_out_lengths(n) == OW.conv_out_lengths(n)
    assert conv_out_lengths(160000)[-1] == 499 and conv_out_lengths(64000)[-1] == 199
    assert conv_out_lengths(640 * 100 + 80)[-1] == 200              # dataset/data_loader.py:340
    assert OW.resample_length(499, "int") == 249 and OW.resample_length(499, "ceil") == 250


def test_host_tables_match_oracle():
    from avi_talking_amd.host import diffusion_prior as HP, faceformer as HF, talking_head as HT
    from oracle import emote as OE, faceformer as OF, prior as OP
    hs, os_ = HP.cosine_schedule(100), OP.cosine_schedule(100)
    assert all(torch.equal(hs[k], os_[k]) for k in os_)
    emb = torch.randn(32, 8, generator=torch.Generator().manual_seed(1))
    assert torch.equal(HP._rel_pos_bias_table(emb, 3), OP.rel_pos_bias(emb, 3, 4))
    assert HF.alibi_slopes(4) == OE.get_slopes(4) and HT.alibi_slopes(8) == OE.get_slopes(8)
    assert torch.equal(HF.ppe_period(64, 30), OF.ppe_table(64, 30)[0, :30])
    assert torch.equal(OF.ppe_table(64, 30)[0, 30:60], OF.ppe_table(64, 30)[0, :30])   # periodic
    # rotary tables: oracle rotates with cos/sin of pos * freq, each freq twice
    c, s = HP._rotary_tables(3)
    t = torch.randn(3, 64, generator=torch.Generator().manual_seed(2))
    rot = OP.rotary(t)
    x = t[:, :32].reshape(3, 16, 2)
    manual = t[:, :32] * c + torch.stack((-x[..., 1], x[..., 0]), -1).reshape(3, 32) * s
    assert torch.allclose(rot[:, :32], manual, atol=1e-7) and torch.equal(rot[:, 32:], t[:, 32:])
    assert torch.allclose(HP._time_table(100)[17], OP.time_embed.__globals__["torch"].cat(
        ((17.0 * torch.exp(torch.arange(64, dtype=torch.float32) * -(torch.log(torch.tensor(10000.0)) / 63))).sin(),
         (17.0 * torch.exp(torch.arange(64, dtype=torch.float32) * -(torch.log(torch.tensor(10000.0)) / 63))).cos())),
        atol=1e-5)


def test_partition_by_length():
    from avi_talking_amd.host.sharding import pad_to_multiple, partition_by_length
    lengths = [250, 100, 600, 30, 250, 249, 1, 400]
    for world in (1, 2, 4, 8):
        parts = partition_by_length(lengths, world)
        assert sorted(i for p in parts for i in p) == list(range(len(lengths)))
        loads = [sum(lengths[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(lengths)
    assert partition_by_length([], 4) == [[], [], [], []]
    assert pad_to_multiple(250) == 256 and pad_to_multiple(256) == 256 and pad_to_multiple(1) == 8


def test_library_has_no_packed_fp32_instructions(lib):
    """build.py compiles every translation unit without v_pk_{fma,mul,add}_f32 (kernels using them were corrupted by
    matrix-core kernels of a second stream; the mechanism is unknown, so the ban is enforced on the shipped binary
    itself): disassemble every gfx950 code object of the library.  Matrix-core instructions must be there."""
    spec = importlib.util.spec_from_file_location("avi_build", os.path.join(ROOT, "avi-talking_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.OBJDUMP):
        pytest.skip("llvm-objdump not available")
    census = mod.instruction_census(lib.LIB_PATH)
    assert census["code_objects"] >= 10
    assert census["v_mfma"] > 2000
    assert census[r"v_pk_(fma|mul|add)_f32"] == 0, census
    assert "packed-fp32" not in lib.version()


def test_loader_refuses_a_packed_fp32_diagnostic_build(lib, monkeypatch):
    """lib.load() raises on a library whose version string marks an AVI_PACKED_FP32=1 build."""
    import ctypes as C

    class Fake:
        def __init__(self, real):
            self._real = real
            self.avi_version = lambda: b"avi_talking_hip 0.2.0 (gfx950, packed-fp32 DIAGNOSTIC build)"
            self.avi_version.restype = None
            self.avi_version.argtypes = None

        def __getattr__(self, k):
            return getattr(self._real, k)

    real = lib.load()
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(C, "CDLL", lambda path: Fake(real))
    monkeypatch.delenv("AVI_ALLOW_PACKED_FP32", raising=False)
    with pytest.raises(RuntimeError, match="packed-FP32"):
        lib.load()
    monkeypatch.setenv("AVI_ALLOW_PACKED_FP32", "1")
    monkeypatch.setattr(lib, "_lib", None)
    assert lib.load() is not None
    monkeypatch.setattr(lib, "_lib", real)
