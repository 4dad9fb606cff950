"""ctypes binding of ``libavi_talking_hip.so`` (the C ABI in ``include/avi_talking.h``).

torch is used here only for device memory and the current HIP stream: every wrapper passes
``tensor.data_ptr()`` and ``torch.cuda.current_stream().cuda_stream`` across the C ABI.
There is no CPU fallback: a missing library or a failed launch raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libavi_talking_hip.so")

ACT_NONE, ACT_GELU, ACT_LRELU02, ACT_RELU, ACT_SILU, ACT_QUICK_GELU = 0, 1, 2, 3, 4, 5
PREC_BF16, PREC_F16X2, PREC_BF16X3 = 1, 2, 3
AVI_OK, AVI_EINVAL, AVI_ENOSPC = 0, -1, -2          # status codes of include/avi_talking.h
PLANES_BF16, PLANES_F16 = 0, 1

_vp, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong


class AviGemm(C.Structure):
    _fields_ = [
        ("A", _vp), ("lda", _ll), ("sAo", _ll), ("sAi", _ll),
        ("Whi", _vp), ("Wlo", _vp), ("sWo", _ll), ("sWi", _ll),
        ("C", _vp), ("ldc", _ll), ("sCo", _ll), ("sCi", _ll),
        ("bias", _vp), ("sBo", _ll), ("sBi", _ll),
        ("R", _vp), ("ldr", _ll), ("sRo", _ll), ("sRi", _ll),
        ("scale", _vp), ("shift", _vp),
        ("M", _i), ("N", _i), ("K", _i),
        ("batch", _i), ("z_inner", _i),
        ("act", _i), ("prec", _i),
        ("Ahi", _vp), ("Alo", _vp), ("Chi", _vp), ("Clo", _vp),
        ("ldw", _i), ("cus", _i),
        ("C16", _vp),
        ("sk_ws", _vp), ("sk_ws_floats", _ll),
    ]


class AviFlameBasis(C.Structure):
    _fields_ = [(n, _vp) for n in ("v_template", "shape_basis", "frame_basis", "j_template", "j_shape", "j_exp",
                                   "lbs_weights")] + [("V", _i), ("n_shape", _i), ("n_exp", _i),
                                                     ("basis_hi", _vp), ("basis_lo", _vp)]


class AviTransposeJob(C.Structure):
    _fields_ = [("in_", _vp), ("out", _vp), ("hi", _vp), ("lo", _vp), ("R", _i), ("C", _i), ("C_pad", _i),
                ("first_block", _i), ("colsum", _vp)]

    def blocks(self):
        """Workgroups of this job in a DEVICE TABLE (avi_transpose_table): plane-only jobs with R % 64 == 0 and
        C_pad % 64 == 0 run on 64 x 64 tiles, everything else on 32 x 32 (avi_talking.h)."""
        if self.colsum:
            return (self.C + 15) // 16
        if self.hi and not self.out and self.R % 64 == 0 and self.C_pad % 64 == 0:
            return (self.C_pad // 64) * (self.R // 64)
        cp = self.C_pad if self.hi else self.C
        return ((cp + 31) // 32) * ((self.R + 31) // 32)


PRIOR_MAX_DEPTH = 8


class AviPriorLayer(C.Structure):
    _fields_ = [(n, _vp) for n in ("norm_g", "wqkv", "null_kv", "wout", "out_g", "ff_g", "w1", "w2")]


class AviPriorWeights(C.Structure):
    _fields_ = ([("depth", _i), ("timesteps", _i)] +
                [(n, _vp) for n in ("time_table", "t_w0", "t_b0", "t_w1", "t_b1", "t_w2", "t_b2", "learned_query",
                                    "null_brain", "null_image", "rel_bias", "rot_cos", "rot_sin")] +
                [("layer", AviPriorLayer * PRIOR_MAX_DEPTH)] +
                [(n, _vp) for n in ("final_g", "wproj", "coef1", "coef2", "logvar")])


class AviPriorLayerPlanes(C.Structure):
    _fields_ = [(n, _vp) for n in ("qkv_hi", "qkv_lo", "out_hi", "out_lo", "w1_hi", "w1_lo", "w2_hi", "w2_lo")]


class AviPriorPlanes(C.Structure):
    _fields_ = [("layer", AviPriorLayerPlanes * PRIOR_MAX_DEPTH), ("proj_hi", _vp), ("proj_lo", _vp)]


class AviFaceformerWeights(C.Structure):
    _fields_ = ([("D", _i), ("V", _i), ("period", _i)] +
                [(n, _vp) for n in ("wqkv", "bqkv", "wo", "bo", "n1g", "n1b", "n2g", "n2b", "n3g", "n3b", "w1", "b1",
                                    "w2", "b2", "wr", "br", "wm", "bm", "pe", "slopes", "obj_embedding",
                                    "coeff_mean", "coeff_std")])


class AviPriorTrainDump(C.Structure):
    _fields_ = [(n, _vp) for n in ("tok0", "tok_in", "n1", "qkv", "ao", "o1", "tokm", "n2", "hff", "sw", "tok_out", "fin",
                                   "po")]


class AviPriorTrainBwd(C.Structure):
    _fields_ = ([(n, _vp) for n in ("dtok_top", "tok_in", "qkv", "o1", "tokm", "hff", "dy_w2", "dy_w1", "dy_out", "dy_qkv",
                                    "dtok0", "dgamma_part")] + [("dnull_kv", _vp * PRIOR_MAX_DEPTH), ("drel", _vp), ("attn_part", _vp)])


class AviPriorGainGrads(C.Structure):
    _fields_ = [("g", (_vp * 3) * PRIOR_MAX_DEPTH)]


class AviPlaneJob(C.Structure):
    _fields_ = [("src_hi", _vp), ("src_lo", _vp), ("dst_hi", _vp), ("dst_lo", _vp), ("N", _i), ("K", _i),
                ("first_block", _i), ("transpose", _i)]

    def blocks(self):
        return (self.N * self.K // 8 + 255) // 256


class AviFaceformerPlanes(C.Structure):
    _fields_ = [(n, _vp) for n in ("wo_hi", "wo_lo", "w1_hi", "w1_lo", "w2_hi", "w2_lo", "wr_hi", "wr_lo", "wf_t", "bf",
                                   "qkv0", "x0")]


# name -> argtypes; every function returns int status.  Kept in one table so the CPU-side test can
# check that the library exports every symbol the header declares.
SIGNATURES = {
    "avi_gemm": [C.POINTER(AviGemm), _vp],
    "avi_pack_weight_split": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "avi_audio_normalize": [_vp, _i, _i, _i, _i, _f, _vp, _vp, _vp],
    "avi_conv0_gn_gelu": [_vp, _i, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp],
    "avi_interp_layernorm": [_vp, _i, _i, _i, _i, _vp, _vp, _f, _vp, _vp],
    "avi_interp_layernorm_planes": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _f, _vp, _vp],
    "avi_conv0_gn_gelu_planes": [_vp, _i, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _vp],
    "avi_flame_vertices": [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "avi_flame_pack_basis": [_vp, _vp, _vp, _vp],
    "avi_transpose_pack_split": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "avi_splitk_epilogue": [_vp, _i, _ll, _i, _i, _vp, _vp, _vp, _f, _i, _i, _vp, _vp, _vp],
    "avi_layernorm_planes": [_vp, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _i, _vp],
    "avi_layernorm": [_vp, _i, _i, _vp, _vp, _f, _vp, _vp],
    "avi_layernorm_act": [_vp, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp],
    "avi_group_pad_pack": [_vp, _i, _i, _i, _i, _i, _vp, _vp],
    "avi_posconv_gelu_residual": [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp],
    "avi_pad_repeat": [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "avi_add_rowbcast": [_vp, _vp, _i, _i, _i, _vp, _vp],
    "avi_embed_tokens": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "avi_mean_tokens": [_vp, _i, _i, _i, _vp, _vp],
    "avi_attention": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _i, _vp],
    "avi_attention_d64": [_vp, _i, _i, _i, _i, _f, _vp, _i, _vp, _vp],
    "avi_attention_d64_planes": [_vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _i, _i, _vp],
    "avi_attention_d64_planes_biased": [_vp, _i, _i, _i, _i, _f, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _vp],
    "avi_prior_forward": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "avi_prior_sample": [_vp, _vp, _vp, _i, _f, _vp, _vp, _vp],
    "avi_prior_sample_batched": [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp],
    "avi_prior_sample_batched_tab": [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp],
    "avi_prior_time_table": [_vp, _vp, _vp],
    "avi_prior_sample_paired": [_vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp],
    "avi_prior_pair_workspace_bytes": [_i],                  # returns long long (RESTYPES below)
    "avi_faceformer_decode": [_vp, _vp, _i, _i, _vp, _vp, _vp],
    "avi_faceformer_decode_chunked": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "avi_faceformer_decode_chunked_f16": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "avi_faceformer_decode_steps_f16": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "avi_rng_fill": [_vp, C.c_uint, _i, _f, _ll, _vp, _vp],
    "avi_rng_advance": [_vp, C.c_ulonglong, _vp],
    "avi_faceformer_tf_embed": [_vp, _vp, _i, _i, _vp, _vp],
    "avi_faceformer_steps_work_floats": [_i, _i, C.POINTER(_ll)],
    "avi_faceformer_persist_sizes": [_i, C.POINTER(_ll), C.POINTER(_ll)],
    "avi_faceformer_persist_pack": [_vp, _vp, _vp, _vp],
    "avi_faceformer_decode_persistent": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "avi_faceformer_decode_steps": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "avi_layernorm_ex": [_vp, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _i, _vp, _vp],
    "avi_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "avi_prior_train_forward": [_vp, _vp, _vp, _i, _i, _vp],
    "avi_prior_train_backward": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "avi_pack_fragment_planes": [_vp, _i, _i, _vp],
    "avi_zero": [_vp, _ll, _vp],
    "avi_copy_rows": [_vp, _ll, _vp, _vp, _ll, _i, _i, _vp],
    "avi_prior_rel_bias": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "avi_transpose": [_vp, _i, _i, _vp, _vp],
    "avi_transpose_jobs": [_vp, _i, _vp],
    "avi_transpose_table": [_vp, _i, _i, _vp],
    "avi_colsum": [_vp, _i, _i, _vp, _i, _vp],
    "avi_act_fwd": [_vp, _ll, _i, _vp, _vp],
    "avi_act_bwd": [_vp, _vp, _ll, _i, _vp, _vp],
    "avi_swiglu_fwd": [_vp, _i, _i, _vp, _vp],
    "avi_swiglu_bwd": [_vp, _vp, _i, _i, _vp, _vp],
    "avi_prior_tokens_fwd": [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp],
    "avi_prior_tokens_bwd": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "avi_prior_attn_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "avi_prior_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp],
    "avi_mse_loss": [_vp, _vp, _i, _f, _vp, _vp, _vp],
    "avi_soft_clip_loss": [_vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp, _vp],
    "avi_adamw": [_vp, _vp, _vp, _vp, _ll, _f, _f, _f, _f, _f, _i, _f, _vp, _vp, _vp, _vp],
    "avi_set_status_words": [_vp],
    "avi_status_words": [],                                  # returns void* (RESTYPES below)
    "avi_debug_fault_inject": [_i],
    "avi_debug_raise_status": [_i, _vp],
    "avi_debug_where": [_vp, _i, _i, _i, _i, _vp],
}

RESTYPES = {"avi_prior_pair_workspace_bytes": _ll, "avi_status_words": _vp}          # everything else returns an int status
_lib = None


def load():
    """Load the shared library (raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python avi-talking_amd/build.py` "
                "(or __graft_entry__.build()). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        lib.avi_version.restype = C.c_char_p
        lib.avi_version.argtypes = []
        if b"packed-fp32" in lib.avi_version() and os.environ.get("AVI_ALLOW_PACKED_FP32", "0") != "1":
            raise RuntimeError(
                f"{LIB_PATH} is a diagnostic build WITH packed-FP32 instructions (AVI_PACKED_FP32=1): kernels using "
                "them were corrupted by matrix-core kernels of a second stream (build.py).  Rebuild without it; "
                "scripts/diag_concurrency.py sets AVI_ALLOW_PACKED_FP32=1 to load such a build on purpose.")
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = RESTYPES.get(name, _i)
        _lib = lib
    return _lib


def version():
    return load().avi_version().decode()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()


def check(status, what):
    if status != 0:
        raise RuntimeError(f"{what} failed with status {status}")


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("avi_talking_amd: tensors must live on the GPU (no CPU fallback)")
