"""Generate the golden vectors under tests/golden/ by IMPORTING the reference's own Python.

Run in the build container only (needs /root/reference; the GPU box has no reference):
    python tests/golden/make_golden.py

What is pinned:
  * wav2vec2_*.npz   reference ``models/lib/wav2vec.py`` ``Wav2Vec2Model`` (HF transformers base class),
                     seeded weights from avi_talking_amd.weights loaded with strict=True
                     (this also pins the state_dict key names), inputs: the reference's fixture WAV
                     (experiments/wav_dir/0001, channel 0, per-clip normalised) and a seeded randn clip.
  * masks.npz        ``models/faceformer.py`` init_biased_mask / enc_dec_mask / PeriodicPositionalEncoding and
                     inferno ``TransformerMasking.py`` init_alibi_biased_mask_future.
  * brain.npz        ``models/diffusion_prior.py`` BrainNetwork (absent third-party imports stubbed).
  * faceformer_*.npz ``models/faceformer.py`` ``Faceformer.predict`` run UNMODIFIED on an object built with
                     ``__new__`` + hand-attached submodules (``__init__`` needs FLAME assets, the network and
                     files that do not exist in the repo); the FAN image encoder (out of scope) is a stub
                     returning zero embeddings.
  * flame.npz        inferno ``utils/lbs.py`` ``lbs`` (imported as is: pure torch) on the synthetic FLAME basis of
                     avi_talking_amd.weights.make_flame_basis, with the pose assembly of ``FLAME.forward``
                     (DecaFLAME.py:236-244); inputs + a slice of the vertices.
Absent packages (cv2, easydict, omegaconf, torchvision, clip, dalle2_pytorch, gdl, pirender ...) are
replaced by MagicMock modules so the reference files import; none of the mocked symbols is on the
executed path.  The dalle2-based prior classes cannot be executed (dalle2_pytorch absent): unpinned.
Fixtures hold inputs/outputs only (small slices), never reference source.
"""
import importlib.util
import os
import sys
import types
import wave
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from avi_talking_amd import weights as W  # noqa: E402


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def read_wav_ch0(path):
    with wave.open(path, "rb") as f:
        assert f.getsampwidth() == 2 and f.getframerate() == 16000
        nch, n = f.getnchannels(), f.getnframes()
        pcm = np.frombuffer(f.readframes(n), dtype="<i2").reshape(n, nch)
    return pcm[:, 0].copy()


def gen_wav2vec2():
    from transformers import Wav2Vec2Config
    ref = load_by_path("ref_wav2vec", os.path.join(REF, "models/lib/wav2vec.py"))
    model = ref.Wav2Vec2Model(Wav2Vec2Config(attn_implementation="eager")).eval()
    w = W.make_wav2vec2_weights(0)
    print(model.load_state_dict(w, strict=True))
    pcm = read_wav_ch0(os.path.join(REF, "experiments/wav_dir/0001/M012_front_neutral_level1_017.wav"))
    xf = torch.from_numpy(pcm.astype(np.float32))
    clip = ((xf - xf.mean()) / torch.sqrt(xf.var(unbiased=False) + 1e-7))[None]       # data_loader.py:289-290
    rnd = torch.randn(1, 32000, generator=torch.Generator().manual_seed(5))
    for tag, x, frame_num in (("fixture", clip, None), ("randn", rnd, None), ("randn_fn40", rnd, 40)):
        with torch.no_grad():
            feats = model.feature_extractor(x)
            out = model(x, "vocaset", frame_num=frame_num).last_hidden_state
        np.savez_compressed(os.path.join(HERE, f"wav2vec2_{tag}.npz"),
                            n_samples=np.int64(x.shape[1]),
                            frame_num=np.int64(-1 if frame_num is None else frame_num),
                            pcm_head=pcm[:64] if tag == "fixture" else np.zeros(0, np.int16),
                            conv_shape=np.array(feats.shape),
                            conv_slice=feats[0, ::16, ::9].numpy(),
                            out_shape=np.array(out.shape),
                            out_slice=out[0, ::3, ::8].numpy())
        print(tag, tuple(feats.shape), tuple(out.shape))
    # the fixture clip itself (int16 channel 0, 160 KB) travels as data so GPU tests can use it
    np.savez_compressed(os.path.join(HERE, "fixture_wav_ch0.npz"), pcm=pcm)


def stub_modules():
    for name in ["cv2", "easydict", "omegaconf", "torchvision", "torchvision.transforms", "clip", "PIL",
                 "dalle2_pytorch", "dalle2_pytorch.dalle2_pytorch", "dalle2_pytorch.train_configs",
                 "gdl", "gdl.models", "gdl.models.DecaFLAME", "gdl.layers", "gdl.layers.losses",
                 "gdl.layers.losses.DecaLosses", "gdl.utils", "gdl.utils.DecaUtils", "gdl.models.DECA",
                 "third_party", "third_party.pirender", "third_party.pirender.generators",
                 "third_party.pirender.generators.face_model", "third_party.pirender.config",
                 "third_party.pirender.loss", "third_party.pirender.loss.perceptual",
                 "third_party.pirender.util", "third_party.pirender.util.meters"]:
        if name not in sys.modules:
            sys.modules[name] = MagicMock()
    # dalle2's DiffusionPrior is used as a BASE CLASS at import time: give it a real class
    sys.modules["dalle2_pytorch"].DiffusionPrior = type("DiffusionPrior", (torch.nn.Module,), {})


def import_reference_models():
    import transformers  # noqa: F401  (before stubbing anything it may probe)
    stub_modules()
    sys.path.insert(0, REF)
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["models"] = pkg
    lib = types.ModuleType("models.lib")
    lib.__path__ = [os.path.join(REF, "models/lib")]
    sys.modules["models.lib"] = lib
    load_by_path("models.lib.wav2vec", os.path.join(REF, "models/lib/wav2vec.py"))
    load_by_path("models.network_utils", os.path.join(REF, "models/network_utils.py"))
    ff = load_by_path("models.faceformer", os.path.join(REF, "models/faceformer.py"))
    dp = load_by_path("models.diffusion_prior", os.path.join(REF, "models/diffusion_prior.py"))
    return ff, dp


def gen_masks(ff):
    tm = load_by_path("ref_masking", os.path.join(
        REF, "third_party/inferno/inferno/models/temporal/TransformerMasking.py"))
    biased = ff.init_biased_mask(n_head=4, max_seq_len=600, period=30)
    biased25 = ff.init_biased_mask(n_head=4, max_seq_len=600, period=25)
    alibi = tm.init_alibi_biased_mask_future(8, 96)
    edm = ff.enc_dec_mask("cpu", "vocaset", 7, 9)
    ppe = ff.PeriodicPositionalEncoding(64, period=30).pe
    np.savez_compressed(os.path.join(HERE, "masks.npz"),
                        biased_p30_block=biased[:, :96, :96].numpy(),
                        biased_p30_strided=biased[:, ::13, ::7].numpy(),
                        biased_p25_block=biased25[:, :64, :64].numpy(),
                        alibi_future_8_96=alibi.numpy(),
                        enc_dec_7_9=edm.numpy(),
                        ppe_64_p30=ppe[0, :100].numpy())
    print("masks", tuple(biased.shape), tuple(alibi.shape))


def gen_brain(dp):
    w = W.make_prior_weights(3)
    net = dp.BrainNetwork(out_dim=128, in_dim=768, clip_size=128, h=4096, n_blocks=4).eval()
    sd = {k[len("voxel2clip."):]: v for k, v in w.items() if k.startswith("voxel2clip.")}
    print(net.load_state_dict(sd, strict=True))
    x = torch.randn(4, 768, generator=torch.Generator().manual_seed(11))
    with torch.no_grad():
        a, b = net(x)
    np.savez_compressed(os.path.join(HERE, "brain.npz"), x=x.numpy(), out=a.numpy(), proj=b.numpy())
    print("brain", tuple(a.shape), tuple(b.shape))


def gen_faceformer(ff):
    from transformers import Wav2Vec2Config
    ref_w2v = sys.modules["models.lib.wav2vec"]
    coeff_mean = np.load(os.path.join(REF, "misc/coeff_mean.npy"))
    coeff_std = np.load(os.path.join(REF, "misc/coeff_std.npy"))
    for D, n_samples in ((64, 32000), (1024, 16000)):
        w = W.make_faceformer_weights(2, feature_dim=D)
        m = ff.Faceformer.__new__(ff.Faceformer)
        torch.nn.Module.__init__(m)
        m.args = types.SimpleNamespace(load_mld=0, period=30, feature_dim=D, vertice_dim=53)
        m.dataset = "vocaset"
        m.device = "cpu"
        m.audio_encoder = ref_w2v.Wav2Vec2Model(Wav2Vec2Config(attn_implementation="eager"))
        m.audio_encoder.load_state_dict(W.make_wav2vec2_weights(0), strict=True)
        m.audio_feature_map = torch.nn.Linear(768, D)
        m.vertice_map = torch.nn.Linear(53, D)
        m.PPE = ff.PeriodicPositionalEncoding(D, period=30)
        m.biased_mask = ff.init_biased_mask(n_head=4, max_seq_len=600, period=30)
        layer = torch.nn.TransformerDecoderLayer(d_model=D, nhead=4, dim_feedforward=2 * D, batch_first=True)
        m.transformer_decoder = torch.nn.TransformerDecoder(layer, num_layers=1)
        m.vertice_map_r = torch.nn.Linear(D, 53)
        m.obj_embedding = torch.nn.Parameter(torch.zeros(1, D))
        m.v_merge2hidden = torch.nn.Linear(6 + 6 + 30 + D, D)
        m.coeff_mean = torch.from_numpy(coeff_mean)[None, None]
        m.coeff_std = torch.from_numpy(coeff_std)[None, None]
        # FAN image encoder is out of scope: zero embeddings of the dims v_merge2hidden expects
        m.fan_net = lambda img: (torch.zeros(1, 6), torch.zeros(1, 6), torch.zeros(1, 30), None)
        g = torch.Generator().manual_seed(21)
        vm_w = (torch.rand(D, 42 + D, generator=g) * 2 - 1) / (42 + D) ** 0.5
        vm_b = (torch.rand(D, generator=g) * 2 - 1) / (42 + D) ** 0.5
        sd = {k: v for k, v in w.items()}
        sd["v_merge2hidden.weight"], sd["v_merge2hidden.bias"] = vm_w, vm_b
        missing = m.load_state_dict(sd, strict=False)
        assert not [k for k in missing.missing_keys if not k.startswith(("audio_encoder", "PPE"))], missing
        assert not missing.unexpected_keys, missing
        m.eval()
        audio = torch.randn(1, n_samples, generator=torch.Generator().manual_seed(5))
        img = torch.zeros(3, 3, 8, 8)
        out = m.predict(audio, img, img, img)
        with torch.no_grad():
            hs_a = m.audio_feature_map(m.audio_encoder(audio, "vocaset").last_hidden_state)
            T = hs_a.shape[1]
            hidden = m.v_merge2hidden(torch.cat([torch.zeros(1, T, 6), torch.zeros(1, T, 30), hs_a,
                                                 torch.zeros(1, T, 6)], -1))
        np.savez_compressed(os.path.join(HERE, f"faceformer_D{D}.npz"), n_samples=np.int64(n_samples),
                            hidden_states=hidden.numpy(), predict=out.numpy())
        print("faceformer", D, tuple(out.shape))


def gen_flame():
    lbs_mod = load_by_path("ref_lbs", os.path.join(REF, "third_party/inferno/inferno/utils/lbs.py"))
    basis = W.make_flame_basis(4)
    g = torch.Generator().manual_seed(21)
    N = 6
    shape = torch.randn(N, 300, generator=g)
    exp = torch.randn(N, 50, generator=g) * 0.8
    pose = torch.randn(N, 6, generator=g) * torch.tensor([0.1, 0.1, 0.1, 0.25, 0.05, 0.05])
    pose[0] = 0.0                                                    # rest pose: exercises the 1e-8 epsilon path
    eye = torch.randn(N, 6, generator=g) * 0.1
    neck = torch.randn(N, 3, generator=g) * 0.1
    betas = torch.cat([shape, exp], 1)
    full_pose = torch.cat([pose[:, :3], neck, pose[:, 3:], eye], 1)  # DecaFLAME.py:240-241
    with torch.no_grad():
        v, J = lbs_mod.lbs(betas, full_pose, basis["v_template"][None].expand(N, -1, -1), basis["shapedirs"],
                           basis["posedirs"], basis["J_regressor"], basis["parents"], basis["lbs_weights"])
    idx = torch.arange(0, 5023, 17)
    np.savez_compressed(os.path.join(HERE, "flame.npz"), shape=shape.numpy(), exp=exp.numpy(), pose=pose.numpy(),
                        eye=eye.numpy(), neck=neck.numpy(), vidx=idx.numpy(), verts=v[:, idx].numpy(),
                        joints=J.numpy(), vsum=v.double().sum((1, 2)).numpy())
    print("flame.npz: verts", tuple(v.shape), "slice", tuple(v[:, idx].shape))


def gen_clip_text():
    """The class FrozenCLIPEmbedder wraps (models/diffusion_prior.py:40,52-53) with the text config of
    openai/clip-vit-large-patch14; from_pretrained needs the network, so the weights are the seeded random init."""
    from transformers import CLIPTextConfig, CLIPTextModel
    cfg = CLIPTextConfig(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                         num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu",
                         layer_norm_eps=1e-5)
    model = CLIPTextModel(cfg).eval()
    w = W.make_clip_text_weights(5)
    keys = set(model.state_dict().keys())             # this transformers drops the "text_model." prefix
    print(model.load_state_dict({(k if k in keys else k[len("text_model."):]): v for k, v in w.items()}, strict=True))
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(0, 49406, (3, 77), generator=g)
    ids[:, 0] = 49406                                  # <|startoftext|>
    for b, n in enumerate((9, 30, 76)):                # <|endoftext|> then padding with it (padding="max_length")
        ids[b, n:] = 49407
    with torch.no_grad():
        out = model(input_ids=ids).last_hidden_state
        short = model(input_ids=ids[:1, :20].contiguous()).last_hidden_state
    np.savez_compressed(os.path.join(HERE, "clip_text.npz"), ids=ids.numpy(), out_shape=np.array(out.shape),
                        out_slice=out[:, :, ::8].numpy(), out_sum=out.double().sum((1, 2)).numpy(),
                        short_slice=short[0, :, ::8].numpy())
    print("clip_text.npz:", tuple(out.shape))


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] in ("flame", "clip_text"):      # regenerate only one fixture
        {"flame": gen_flame, "clip_text": gen_clip_text}[sys.argv[1]]()
        sys.exit(0)
    gen_flame()
    gen_clip_text()
    gen_wav2vec2()
    ff, dp = import_reference_models()
    gen_masks(ff)
    gen_brain(dp)
    gen_faceformer(ff)
    np.save(os.path.join(HERE, "coeff_mean.npy"), np.load(os.path.join(REF, "misc/coeff_mean.npy")))
    np.save(os.path.join(HERE, "coeff_std.npy"), np.load(os.path.join(REF, "misc/coeff_std.npy")))
    print("done")
