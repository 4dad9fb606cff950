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
  * faceformer_tf.npz the teacher-forced decoder pass (models/faceformer.py:378-391) on the reference's own submodules,
                     D = 64 and 1024, T = 49 and 250.
  * flame.npz        inferno ``utils/lbs.py`` ``lbs`` (imported as is: pure torch) on the synthetic FLAME basis of
                     avi_talking_amd.weights.make_flame_basis, with the pose assembly of ``FLAME.forward``
                     (DecaFLAME.py:236-244); inputs + a slice of the vertices.
  * fixture_chain.npz the fixture WAV (and two seeded clips) through the reference's wav2vec2 wrapper AND its EMOTE head +
                     FLINT decoder: audio -> coefficients on the reference's own modules (gen_fixture_chain).
  * emote_audio.npz  EMOTE's own audio wrapper (AudioEncoders.py ``Wav2Vec2Encoder._forward`` + ``Wav2Vec2ModelResampled``) on
                     int16 raw_audio: processor call with its joint statistics, desired_output_length, the ceil length rule.
  * sample_dict.npz  the reference's own ``read_audio`` / ``process_audio`` / ``create_base_sample`` / ``create_condition`` /
                     ``create_high_intensity_emotions`` on the fixture WAV (gen_sample_dict; own process, like train_helpers).
  * train_helpers.npz ``train_diffusion_prior.py``'s own ``soft_clip_loss`` / ``cosine_anneal`` / ``batchwise_cosine_similarity``
                     / ``topk`` (the entry point imported as a module, everything it pulls in stubbed).
  * emote.npz        inferno ``LinearSequenceEncoder`` (SequenceEncoders.py:180-197), ``LinearEmotionCondition``
                     (FaceFormerDecoder.py:186-268), ``BertPriorDecoder.forward`` -> ``FeedForwardDecoder.forward/_style``,
                     ``_decode``, ``_post_prediction`` -> ``_apply_motion_prior`` (:598-682,1104-1224) with
                     ``StackLinearSquash`` (:967-985), ``MotionPrior.decoding_step`` / ``decompose_sequential_output``
                     (MotionPrior.py:316-329,371-375) and ``L2lDecoder`` (L2lMotionPrior.py:361-495): the reference's
                     own classes, METHODS RUN UNMODIFIED.  ``L2lDecoder``, ``StackLinearSquash``, ``LinearSequenceEncoder``
                     and ``LinearEmotionCondition`` are built by their own constructors from the yaml values
                     (bertprior_wild.yaml, l2l_decoder.yaml, l2l_sizes.yaml); ``BertPriorDecoder`` / ``L2lVqVae`` are built
                     with ``__new__`` + hand-attached submodules (their ``__init__`` loads a trained FLINT checkpoint and
                     the licensed FLAME model from cluster paths).  The FLAME vertex post-processor (out of scope here,
                     row 8f-1 has its own fixture) is a stub returning zero vertices.  Absent packages get FUNCTIONAL
                     stand-ins for the few symbols the executed path touches (``OmegaConf.to_container``, ``munchify``,
                     ``open_dict``, ``pl.LightningModule = nn.Module``); everything else is a MagicMock.
                     ``torch.backends.mha.set_fastpath_enabled(False)``: torch >= 1.12 routes an eval-mode
                     ``TransformerEncoderLayer`` through a fused fast path that mishandles the (B*H,T,T) FLOAT mask of
                     ``L2lDecoder`` (0.96 max-abs off the mathematical definition on this input); the reference pins
                     torch 1.9, which has no fast path, so the slow path IS the reference behaviour.
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


def _reference_faceformer(ff, D, with_audio=True):
    """``models/faceformer.py`` ``Faceformer`` built with ``__new__`` + hand-attached submodules of the ctor's own
    types (:138-158); returns the module and the state_dict it was loaded from."""
    from transformers import Wav2Vec2Config
    ref_w2v = sys.modules["models.lib.wav2vec"]
    w = W.make_faceformer_weights(2, feature_dim=D)
    m = ff.Faceformer.__new__(ff.Faceformer)
    torch.nn.Module.__init__(m)
    m.args = types.SimpleNamespace(load_mld=0, period=30, feature_dim=D, vertice_dim=53)
    m.dataset = "vocaset"
    m.device = "cpu"
    if with_audio:
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
    return m, w


def gen_faceformer_teacher_forced(ff):
    """The teacher-forced decoder pass of ``Faceformer.forward_switch_frame`` (models/faceformer.py:378-391): the
    statements of those lines executed on the reference's own submodules (``vertice_map``, ``PPE``, ``biased_mask``,
    module-level ``enc_dec_mask``, ``transformer_decoder``, ``vertice_map_r``) - the method itself cannot run (FAN image
    encoder, ``coeff2style``, FLAME criterion).  ``hidden_states`` stands for ``hidden_states_mix[j:j+1]`` (:376).
    Inputs are stored rounded to fp16 (exactly representable), outputs in full."""
    out = {}
    for D in (64, 1024):
        m, w = _reference_faceformer(ff, D, with_audio=False)
        missing = m.load_state_dict(w, strict=False)
        assert not [k for k in missing.missing_keys if not k.startswith("PPE")], missing
        assert not missing.unexpected_keys, missing
        m.eval()
        for T in (49, 250):
            g = torch.Generator().manual_seed(100 + T)
            hidden_states = torch.randn(1, T, D, generator=g).half().float()
            coeff = (torch.randn(1, T, 53, generator=g) * 0.7).half().float()
            with torch.no_grad():
                # --- models/faceformer.py:382-391, statement by statement -------------------------------------
                vertice_input = torch.cat([torch.zeros_like(coeff[:, -1:]), coeff[:, :-1]], 1)
                vertice_input = m.vertice_map(vertice_input)
                vertice_input = m.PPE(vertice_input)
                tgt_mask = m.biased_mask[:, :vertice_input.shape[1], :vertice_input.shape[1]].clone().detach().to(
                    device=m.device)
                memory_mask = ff.enc_dec_mask(m.device, m.dataset, vertice_input.shape[1], hidden_states.shape[1])
                vertice_out = m.transformer_decoder(vertice_input, hidden_states, tgt_mask=tgt_mask,
                                                    memory_mask=memory_mask)
                vertice_out = m.vertice_map_r(vertice_out)
            out[f"D{D}_T{T}_hidden"] = hidden_states[0].numpy().astype(np.float16)
            out[f"D{D}_T{T}_coeff"] = coeff[0].numpy().astype(np.float16)
            out[f"D{D}_T{T}_out"] = vertice_out[0].numpy()
            print("teacher-forced", D, T, tuple(vertice_out.shape), float(vertice_out.abs().max()))
    np.savez_compressed(os.path.join(HERE, "faceformer_tf.npz"), **out)


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


class _Munch(dict):
    """Functional stand-in for munch.Munch / an OmegaConf DictConfig node: a dict with attribute access."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _munchify(x):
    if isinstance(x, dict):
        return _Munch({k: _munchify(v) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return type(x)(_munchify(v) for v in x)
    return x


def _to_container(x, **kw):
    if isinstance(x, dict):
        return {k: _to_container(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_to_container(v) for v in x]
    return x


def import_inferno():
    """Import the reference's EMOTE / FLINT modules from third_party/inferno as they lie."""
    import contextlib
    inf = os.path.join(REF, "third_party/inferno/inferno")
    oc = types.ModuleType("omegaconf")
    oc.OmegaConf = type("OmegaConf", (), {"to_container": staticmethod(_to_container)})
    oc.DictConfig = _Munch
    oc.open_dict = lambda cfg: contextlib.nullcontext(cfg)
    sys.modules["omegaconf"] = oc
    mu = types.ModuleType("munch")
    mu.Munch, mu.munchify = _Munch, _munchify
    sys.modules["munch"] = mu
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = torch.nn.Module
    sys.modules["pytorch_lightning"] = pl
    for n in ("pytorch3d", "pytorch3d.transforms"):
        sys.modules[n] = MagicMock()
    for name, sub in (("inferno", ""), ("inferno.models", "models"), ("inferno.models.temporal", "models/temporal"),
                      ("inferno.models.temporal.motion_prior", "models/temporal/motion_prior"),
                      ("inferno.models.talkinghead", "models/talkinghead"), ("inferno.utils", "utils"),
                      ("inferno.layers", "layers"), ("inferno.layers.losses", "layers/losses")):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(inf, sub)]
        sys.modules[name] = m
    # modules that drag in renderers / datasets / pytorch3d and are not on the executed path
    for n in ("inferno.layers.losses.RotationLosses", "inferno.models.MLP", "inferno.models.IO", "inferno.utils.other",
              "inferno.models.temporal.BlockFactory", "inferno.utils.ValueScheduler"):
        sys.modules[n] = MagicMock()
    import importlib
    return (importlib.import_module("inferno.models.temporal.motion_prior.L2lMotionPrior"),
            importlib.import_module("inferno.models.talkinghead.FaceFormerDecoder"),
            importlib.import_module("inferno.models.temporal.SequenceEncoders"))


def build_reference_emote(n_identities=32, n_shape=300):
    """The reference's EMOTE head + FLINT decoder as a module tree with the TalkingHeadBase attribute names
    (``sequence_encoder``, ``sequence_decoder``) so that make_emote_weights() loads with strict=True."""
    nn = torch.nn
    L2l, FFD, SE = import_inferno()
    torch.backends.mha.set_fastpath_enabled(False)      # see the module docstring
    # motion_prior_conf/model/sequence_decoder/l2l_decoder.yaml + sizes/l2l_sizes.yaml
    dcfg = _munchify(dict(type="L2lDecoder", num_layers=1, feature_dim=256, intermediate_size=384, nhead=8, dropout=0.0,
                          activation="gelu", positional_encoding=dict(type="none"),
                          temporal_bias=dict(type="alibi_future", max_len=600), last_layer_init=False))
    sizes = _munchify(dict(quant_factor=3, sequence_length=32, quant_sequence_length=4))
    prior = L2l.L2lVqVae.__new__(L2l.L2lVqVae)
    nn.Module.__init__(prior)
    prior.cfg = _munchify(dict(model=dict(sequence_components=dict(exp=50, jaw="rot"), rotation_representation="aa",
                                          sizes=sizes)))
    prior.motion_decoder = L2l.L2lDecoder(dcfg, sizes, 53)
    prior.motion_quantizer = None
    prior.motion_encoder = None                           # BertPriorDecoder.__init__ discards it (:1025)

    def flame_post(rec_batch, input_key=None, output_prefix="", with_grad=True):   # FLAME vertices: out of scope
        B, T = rec_batch["gt_exp"].shape[:2]
        rec_batch[output_prefix + "vertices"] = torch.zeros(B, T, 6)
        return rec_batch
    prior.postprocessor = flame_post
    # talkinghead_conf/model/sequence_decoder/bertprior_wild.yaml
    style_cfg = _munchify(dict(type="emotion_linear", use_shape=True, shape_dim=n_shape, use_video_expression=False,
                               gt_expression_label=True, gt_expression_intensity=True, n_intensities=3,
                               gt_expression_identity=True, n_identities=n_identities, disentangle_identity=False,
                               use_expression=False, n_expression=8, use_valence=False, use_arousal=False,
                               use_emotion_feature=False, use_bias=True))
    cfg = _munchify(dict(type="BertPriorDecoder", num_layers=1, feature_dim=128, nhead=8, dropout=0.25, activation="gelu",
                         squash_before=False, squash_after=True, squash_type="stack_linear", post_bug_fix=True,
                         positional_encoding=dict(type="none"), style_embedding=style_cfg,
                         motion_prior=dict(trainable=False), flame=dict(n_shape=n_shape)))
    dec = FFD.BertPriorDecoder.__new__(FFD.BertPriorDecoder)
    nn.Module.__init__(dec)
    dec.cfg = cfg
    dec.style_type = cfg.style_embedding
    dec.obj_vector = FFD.LinearEmotionCondition(style_cfg, output_dim=128)          # style_from_cfg (:87-97)
    dec.style_op = "add"
    dec.PE = None                                                                 # positional_encoding.type none
    layer = nn.TransformerEncoderLayer(d_model=128, nhead=8, dim_feedforward=128, activation="gelu", dropout=0.25,
                                       batch_first=True)                          # :995-1001
    dec.bert_decoder = nn.TransformerEncoder(layer, num_layers=1)
    dec.post_bug_fix = True
    dec.temporal_bias_type, dec.biased_mask = "none", None
    dec.motion_prior = prior
    dec.latent_frame_size = 8                                                     # 2 ** quant_factor
    dec.squasher = None
    dec.decoder = nn.Linear(128, 256)                                             # :1034
    dec.squasher_2 = FFD.StackLinearSquash(256, 8, 256)                           # _create_squasher (:1049-1055)

    class Flame(nn.Module):
        def forward(self, shape, exp):
            return torch.zeros(shape.shape[0], 2, 3), None, None
    dec.flame = Flame()
    root = nn.Module()
    root.sequence_encoder = SE.LinearSequenceEncoder(_munchify(dict(feature_dim=128, input_feature_dim=768)))
    root.sequence_decoder = dec
    print(root.load_state_dict(W.make_emote_weights(1, n_identities, n_shape), strict=True))
    return root.eval()


def run_reference_emote(root, feat, style_emb=None, sample_extra=None):
    """TalkingHeadBase.forward after forward_audio (:531-553): sequence encoder, then the decoder's own forward."""
    B, T = feat.shape[:2]
    sample = {"fused_feature": feat, "processed_audio": feat, "raw_audio": torch.zeros(B, T, 640),
              "template": torch.zeros(B, 6), "gt_shape": torch.zeros(B, 300)}
    sample.update(sample_extra or {})
    with torch.no_grad():
        sample = root.sequence_encoder(sample, input_key="fused_feature")
        sample = root.sequence_decoder(sample, style_emb=style_emb, is_external_style_emb=style_emb is not None)
    return sample


def gen_emote():
    root = build_reference_emote()
    out = {}
    for tag, B, T in (("a", 2, 250), ("b", 1, 61), ("c", 3, 8)):      # the shapes of tests/test_gpu_emote.py
        g = torch.Generator().manual_seed(31)
        feat, style = torch.randn(B, T, 768, generator=g), torch.randn(B, 1, 128, generator=g) * 0.5
        s = run_reference_emote(root, feat, style)
        assert s["predicted_exp"].shape == (B, T, 50) and s["predicted_jaw"].shape == (B, T, 3)
        out[f"{tag}_shape"] = np.array([B, T])
        out[f"{tag}_seq_encoder_output"] = s["seq_encoder_output"][:, ::7, ::5].numpy()
        out[f"{tag}_latent"] = s["prior_input_sequence"].numpy()
        out[f"{tag}_exp"], out[f"{tag}_jaw"] = s["predicted_exp"].numpy(), s["predicted_jaw"].numpy()
    # per-frame (B,T,128) external style, and the style from the sample's own one-hot conditions
    B, T = 2, 25
    g = torch.Generator().manual_seed(41)
    feat = torch.randn(B, T, 768, generator=g)
    style_t = torch.randn(B, T, 128, generator=g) * 0.5
    one_hot = torch.nn.functional.one_hot
    cond = {"gt_expression_label_condition": one_hot(torch.tensor([3, 5]), 8)[:, None].expand(B, T, 8),
            "gt_expression_intensity_condition": one_hot(torch.tensor([2, 0]), 3)[:, None].expand(B, T, 3),
            "gt_expression_identity_condition": one_hot(torch.tensor([7, 30]), 32),      # (B, N): expanded by the module
            "gt_shape": torch.randn(B, 300, generator=g), "gt_vertices": torch.zeros(B, T, 6)}
    s = run_reference_emote(root, feat, style_t)
    out["t_exp"], out["t_jaw"] = s["predicted_exp"].numpy(), s["predicted_jaw"].numpy()
    with torch.no_grad():
        sample = dict(cond, seq_encoder_output=feat[..., :128])
        out["own_style"] = root.sequence_decoder(sample, only_style_emb=True).numpy()
    s = run_reference_emote(root, feat, None, cond)
    out["own_exp"], out["own_jaw"] = s["predicted_exp"].numpy(), s["predicted_jaw"].numpy()
    # L2lDecoder alone on a latent sequence (FLINT decoder, L2lMotionPrior.py:460-495)
    z = torch.randn(2, 5, 256, generator=torch.Generator().manual_seed(43))
    with torch.no_grad():
        out["flint_z_out"] = root.sequence_decoder.motion_prior.motion_decoder(
            {"encoded_features": z}, input_key="encoded_features")["decoded_sequence"].numpy()
    np.savez_compressed(os.path.join(HERE, "emote.npz"), **out)
    print("emote.npz:", {k: v.shape for k, v in out.items()})


def gen_fixture_chain():
    """The reference's one real input end to end, through the reference's OWN modules: fixture WAV (channel 0, whole
    640-sample frames, per-clip normalised) -> ``models/lib/wav2vec.py`` Wav2Vec2Model (frame_num = T, the call of
    models/faceformer.py:330,673) -> last_hidden_state -> LinearSequenceEncoder -> BertPriorDecoder (external style, the
    call of evaluation_functions.py:381) -> FLINT L2lDecoder -> predicted_exp / predicted_jaw.  A second case: two seeded
    2 s clips.  Pins the audio -> coefficient chain of every precision plan on the reference itself (no oracle)."""
    from transformers import Wav2Vec2Config
    ref = load_by_path("ref_wav2vec", os.path.join(REF, "models/lib/wav2vec.py"))
    model = ref.Wav2Vec2Model(Wav2Vec2Config(attn_implementation="eager")).eval()
    print(model.load_state_dict(W.make_wav2vec2_weights(0), strict=True))
    root = build_reference_emote()
    out = {}
    pcm = read_wav_ch0(os.path.join(REF, "experiments/wav_dir/0001/M012_front_neutral_level1_017.wav"))
    T = len(pcm) // 640
    fix = torch.from_numpy(pcm[:T * 640].astype(np.float32))[None]
    g = torch.Generator().manual_seed(77)
    rnd = (torch.randn(2, 50 * 640, generator=g) * 3000).round().clamp(-32768, 32767)       # int16-valued, like raw_audio
    # the fixture as the reference's ENTRY POINT frames it: create_base_sample's padding line adds one zero frame and one
    # zero sample column (raw_audio (125, 641), sample_dict.npz); the audio model sees the rows back to back, 125 frames
    ref_frames = np.load(os.path.join(HERE, "sample_dict.npz"))["base_raw_audio"]
    fix641 = torch.from_numpy(ref_frames.astype(np.float32).reshape(1, -1))
    for tag, x, T in (("fixture", fix, None), ("randn2", rnd, None), ("fixture641", fix641, ref_frames.shape[0])):
        B, T = x.shape[0], (x.shape[1] // 640 if T is None else T)
        xn = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-7)
        style = torch.randn(B, 1, 128, generator=g) * 0.5
        with torch.no_grad():
            feat = model(xn, "vocaset", frame_num=T).last_hidden_state
        s = run_reference_emote(root, feat, style)
        assert s["predicted_exp"].shape == (B, T, 50)
        out[f"{tag}_pcm"] = x.numpy().astype(np.int16)
        out[f"{tag}_style"] = style.numpy()
        out[f"{tag}_hidden_slice"] = feat[:, ::5, ::16].numpy()
        out[f"{tag}_exp"], out[f"{tag}_jaw"] = s["predicted_exp"].numpy(), s["predicted_jaw"].numpy()
        print(tag, tuple(feat.shape), float(s["predicted_exp"].abs().max()), float(s["predicted_jaw"].abs().max()))
    np.savez_compressed(os.path.join(HERE, "fixture_chain.npz"), **out)


def gen_train_helpers():
    """The loss / schedule helpers of the reference ENTRY POINT itself: ``train_diffusion_prior.py`` imported as a module
    (its module-level imports of datasets, the EMOTE wrapper, pirender's meters, talkclip and dalle2 are MagicMock stubs -
    none is on the executed path) and its own ``soft_clip_loss`` (:125-133), ``cosine_anneal`` (:122-123),
    ``batchwise_cosine_similarity`` (:146-153) and ``topk`` (:139-145) run on seeded inputs of the training step's shapes
    (B = 64 L2-normalised 128-d rows, the reference's temperatures 0.004 ... 0.0075)."""
    import transformers  # noqa: F401
    stub_modules()
    for name in ["inferno_apps", "inferno_apps.TalkingHead", "inferno_apps.TalkingHead.evaluation",
                 "inferno_apps.TalkingHead.evaluation.TalkingHeadWrapper",
                 "inferno_apps.TalkingHead.evaluation.evaluation_functions", "inferno", "inferno.datasets",
                 "inferno.datasets.FaceVideoDataModule", "dataset", "dataset.data_loader", "talkclip_text_generation",
                 "talkclip_text_generation.text_gen", "emoca_utils", "models", "models.diffusion_prior"]:
        m = MagicMock()
        m.__all__ = []                      # `from ... import *` of a stub imports nothing
        sys.modules[name] = m
    ref = load_by_path("ref_train_entry", os.path.join(REF, "train_diffusion_prior.py"))
    g = torch.Generator().manual_seed(2024)
    out = {}
    B = 64
    preds = torch.nn.functional.normalize(torch.randn(B, 128, generator=g), dim=-1)
    targs = torch.nn.functional.normalize(torch.randn(B, 128, generator=g) + 0.5 * preds, dim=-1)
    out["preds"], out["targs"] = preds.numpy(), targs.numpy()
    temps = ref.cosine_anneal(0.004, 0.0075, 40)
    out["cosine_anneal_0.004_0.0075_40"] = temps.numpy()
    out["cosine_anneal_1_0_7"] = ref.cosine_anneal(1.0, 0.0, 7).numpy()
    for t in (0.004, 0.005, 0.0075, 0.125):
        out[f"soft_clip_loss_T{t}"] = np.float64(ref.soft_clip_loss(preds, targs, temp=t))
    sim = ref.batchwise_cosine_similarity(preds, targs)
    out["batchwise_cosine_similarity"] = sim.numpy()
    labels = torch.arange(B)
    out["topk_1"] = np.float64(ref.topk(sim, labels, k=1))
    out["topk_5"] = np.float64(ref.topk(sim, labels, k=5))
    np.savez_compressed(os.path.join(HERE, "train_helpers.npz"), **out)
    print("train_helpers.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


def gen_sample_dict():
    """Row F / A0: the reference's OWN sample builders - ``read_audio`` / ``process_audio`` / ``create_base_sample`` /
    ``create_condition`` / ``create_high_intensity_emotions`` of inferno_apps/TalkingHead/evaluation/evaluation_functions.py,
    imported as they lie (renderers, mesh I/O, datasets stubbed; ``librosa.load`` - the decoder, absent here - is a functional
    stand-in that returns the fixture WAV's channel 0 as floats at 16 kHz, everything after the decode is the reference's
    code) and ``FpParser.recursive_collate`` semantics are not needed for the fixture.  The talking-head object the builders
    query for three counts and two names is a stub with the released EMOTE config's values."""
    import enum
    for name in ["inferno_apps", "inferno_apps.TalkingHead", "inferno_apps.TalkingHead.evaluation",
                 "inferno_apps.TalkingHead.evaluation.TalkingHeadWrapper", "inferno_apps.TalkingHead.utils",
                 "inferno_apps.TalkingHead.utils.video", "inferno", "inferno.datasets", "inferno.datasets.FaceVideoDataModule",
                 "inferno.utils", "inferno.utils.collate", "inferno.utils.PyRenderMeshSequenceRenderer",
                 "inferno.datasets.AffectNetAutoDataModule", "trimesh", "soundfile", "psbody", "psbody.mesh",
                 "inferno.utils.other", "librosa"]:
        sys.modules[name] = MagicMock()

    class AffectNetExpressions(enum.Enum):       # functional stand-in: only .Neutral.value and (value).name are used
        Neutral, Happy, Sad, Surprise, Fear, Disgust, Anger, Contempt = range(8)
    sys.modules["inferno.datasets.AffectNetAutoDataModule"].AffectNetExpressions = AffectNetExpressions
    wav = os.path.join(REF, "experiments/wav_dir/0001/M012_front_neutral_level1_017.wav")
    pcm0 = read_wav_ch0(wav)
    sys.modules["librosa"].load = lambda path, sr=None: (pcm0.astype(np.float32) / 32768.0, 16000)
    ev = load_by_path("ref_eval_functions", os.path.join(
        REF, "third_party/inferno/inferno_apps/TalkingHead/evaluation/evaluation_functions.py"))
    style = _munchify(dict(gt_expression_label=True, gt_expression_intensity=True, gt_expression_identity=True))
    th = MagicMock()
    th.cfg = _munchify(dict(data=dict(reconstruction_type=["EMICA-MEAD_flame2020"]),
                            model=dict(sequence_decoder=dict(style_embedding=style))))
    th.get_num_emotions.return_value, th.get_num_intensities.return_value, th.get_num_identities.return_value = 8, 3, 32
    subjects = [f"M{i:03d}" for i in range(32)]
    th.get_subject_labels.return_value = subjects
    out = {}
    wavdata, sr = ev.read_audio(wav)
    out["read_audio"], out["read_audio_sr"] = wavdata, np.int64(sr)
    fr = ev.process_audio(wavdata, sr, 25)
    out["process_audio_raw"] = fr["raw_audio"]
    base = ev.create_base_sample(th, wav)
    out["base_raw_audio"] = base["raw_audio"]
    rec = base["reconstruction"]["EMICA-MEAD_flame2020"]
    out["base_rec_shapes"] = np.array([rec["gt_exp"].shape[0], rec["gt_exp"].shape[1], rec["gt_shape"].shape[0],
                                       rec["gt_jaw"].shape[1], rec["gt_tex"].shape[0]])
    for k in ("gt_expression_label_condition", "gt_expression_intensity_condition", "gt_expression_identity_condition"):
        out["base_" + k] = base[k]
    b2 = ev.create_base_sample(th, wav, smallest_unit=8, silent_frames_start=3, silent_frames_end=2)
    out["base8_raw_audio_shape"] = np.array(b2["raw_audio"].shape)
    out["base8_raw_audio_sum"] = np.float64(b2["raw_audio"].astype(np.float64).sum())
    hs = ev.create_high_intensity_emotions(th, base, identity_list=[5, 30], emotion_index_list=[3, 6], intensity_list=[2, 0])
    for i, h in enumerate(hs):
        for k in ("gt_expression_label_condition", "gt_expression_intensity_condition", "gt_expression_identity_condition"):
            out[f"hi{i}_{k}"] = h[k]
        out[f"hi{i}_name"] = np.array(h["output_name"])
    hq = ev.create_high_intensity_emotions(th, base, identity_list=[1], emotion_index_list=[4], intensity_list=[1],
                                           silent_frames_start=4, silent_emotion_start=0)
    out["hiq_gt_expression_label_condition"] = hq[0]["gt_expression_label_condition"]
    np.savez_compressed(os.path.join(HERE, "sample_dict.npz"), **out)
    print("sample_dict.npz:", {k: (v.shape if hasattr(v, "shape") and v.shape else v) for k, v in out.items()})


def gen_emote_audio():
    """Rows A1-A4 through EMOTE's OWN audio wrapper: ``inferno/models/temporal/AudioEncoders.py`` imported as it lies
    (``inferno.models.temporal.Bases.TemporalAudioEncoder`` = nn.Module, ``inferno.utils.other`` stubbed: neither is on the
    executed path) - ``Wav2Vec2ModelResampled`` built by its constructor from the library-default config (= base-960h's
    architecture) with our seeded weights, ``Wav2Vec2Encoder`` via ``__new__`` (its ``__init__`` fetches the pretrained model
    and processor by name) with the installed ``Wav2Vec2FeatureExtractor`` as ``input_processor`` (the object the reference's
    ``Wav2Vec2Processor`` forwards audio to) - and ``Wav2Vec2Encoder._forward(sample)`` run UNMODIFIED on int16 ``raw_audio``
    (B, T, 640): the processor's joint-over-the-batch statistics (:170-178), ``desired_output_length = T``, the
    ``ceil`` length rule of ``temporal_interpolation`` when no length is given."""
    from transformers import Wav2Vec2Config, Wav2Vec2FeatureExtractor
    bases = types.ModuleType("inferno.models.temporal.Bases")
    bases.TemporalAudioEncoder = torch.nn.Module
    for name, mod in (("inferno", MagicMock()), ("inferno.models", MagicMock()), ("inferno.models.temporal", MagicMock()),
                      ("inferno.models.temporal.Bases", bases), ("inferno.utils", MagicMock()),
                      ("inferno.utils.other", MagicMock())):
        sys.modules[name] = mod
    ae = load_by_path("ref_audio_encoders", os.path.join(REF, "third_party/inferno/inferno/models/temporal/AudioEncoders.py"))
    model = ae.Wav2Vec2ModelResampled(Wav2Vec2Config(attn_implementation="eager")).eval()
    print(model.load_state_dict(W.make_wav2vec2_weights(0), strict=True))
    enc = ae.Wav2Vec2Encoder.__new__(ae.Wav2Vec2Encoder)
    torch.nn.Module.__init__(enc)
    enc.model, enc.resampling, enc.dropout, enc.trainable = model, True, None, False
    enc.input_processor = Wav2Vec2FeatureExtractor(feature_size=1, sampling_rate=16000, padding_value=0.0, do_normalize=True,
                                                   return_attention_mask=False)
    g = torch.Generator().manual_seed(2025)
    raw = (torch.randn(2, 30, 640, generator=g) * torch.tensor([800.0, 5000.0])[:, None, None]).to(torch.int16)
    with torch.no_grad():
        s = enc._forward({"raw_audio": raw.clone(), "samplerate": [16000, 16000]})
        free = model(s["processed_audio"][:, :19000].contiguous()).last_hidden_state          # no length given: ceil rule
    out = {"raw_audio": raw.numpy(), "processed_audio_slice": s["processed_audio"][:, ::97].numpy(),
           "audio_feature_shape": np.array(s["audio_feature"].shape), "audio_feature_slice": s["audio_feature"][:, :, ::8].numpy(),
           "free_len_input": np.int64(19000), "free_len_shape": np.array(free.shape), "free_len_slice": free[:, :, ::8].numpy()}
    np.savez_compressed(os.path.join(HERE, "emote_audio.npz"), **out)
    print("emote_audio.npz:", {k: (v.shape if hasattr(v, "shape") and v.shape else v) for k, v in out.items()})


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
    if len(sys.argv) > 1 and sys.argv[1] in ("flame", "clip_text", "emote", "fixture_chain", "train_helpers", "sample_dict", "emote_audio"):
        {"flame": gen_flame, "clip_text": gen_clip_text, "emote": gen_emote, "fixture_chain": gen_fixture_chain,
         "train_helpers": gen_train_helpers, "sample_dict": gen_sample_dict, "emote_audio": gen_emote_audio}[sys.argv[1]]()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "faceformer_tf":
        gen_faceformer_teacher_forced(import_reference_models()[0])
        sys.exit(0)
    gen_flame()
    gen_clip_text()
    gen_wav2vec2()
    gen_emote()
    gen_fixture_chain()
    ff, dp = import_reference_models()
    # (gen_train_helpers replaces `models*` in sys.modules by stubs: run it in its own process, `make_golden.py train_helpers`)
    gen_masks(ff)
    gen_brain(dp)
    gen_faceformer(ff)
    gen_faceformer_teacher_forced(ff)
    np.save(os.path.join(HERE, "coeff_mean.npy"), np.load(os.path.join(REF, "misc/coeff_mean.npy")))
    np.save(os.path.join(HERE, "coeff_std.npy"), np.load(os.path.join(REF, "misc/coeff_std.npy")))
    print("done")
