"""Diagnostic: what the fp16 hi/lo activation planes and the 2-term fp16 GEMM deliver when the activations are small.
conv layer 0's GroupNorm gain / bias are scaled by s; printed per s: the planes' reconstruction error and conv layer 1's
output error (2-term fp16 vs 3-term bf16), both relative to the tensor's rms, and the status words."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import avi_talking_amd  # noqa: E402
from avi_talking_amd import ops, weights as W  # noqa: E402
from avi_talking_amd.host import status  # noqa: E402

dev = torch.device("cuda:0")
wa = W.make_wav2vec2_weights(0)
fe = "feature_extractor.conv_layers."
x = torch.randn(2, 32000, generator=torch.Generator().manual_seed(5)).to(dev)
w0 = wa[fe + "0.conv.weight"].reshape(512, 10).contiguous().to(dev)
pw = ops.PackedWeight(wa[fe + "1.conv.weight"].permute(0, 2, 1).reshape(512, -1).to(dev))
pw.f16_plane()
status.words()
for s in (1.0, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6):
    g, b = (wa[fe + "0.layer_norm.weight"] * s).to(dev), (wa[fe + "0.layer_norm.bias"] * s).to(dev)
    ref = ops.conv0_gn_gelu(x, w0, g, b)
    status.clear()
    p16 = ops.conv0_gn_gelu_planes(x, w0, g, b, fmt=ops.PLANES_F16)
    pb = ops.conv0_gn_gelu_planes(x, w0, g, b, fmt=ops.PLANES_BF16)
    rms = ref.pow(2).mean().sqrt().item()
    e16 = (p16.float() - ref).abs().max().item() / rms
    eb = (pb.float() - ref).abs().max().item() / rms
    y16 = ops.conv1d_cl_planes(p16, pw, 3, 2, act=ops.ACT_GELU, prec=ops.PREC_F16X2, out_planes=False)
    y3 = ops.conv1d_cl_planes(pb, pw, 3, 2, act=ops.ACT_GELU, prec=ops.PREC_BF16X3, out_planes=False)
    torch.cuda.synchronize()
    yr = y3.pow(2).mean().sqrt().item()
    print(f"s={s:g}: rms {rms:.2e}; planes max err / rms: fp16 {e16:.2e} bf16 {eb:.2e}; conv1 out rms {yr:.2e}, "
          f"2-term fp16 vs 3-term bf16 max / rms {(y16 - y3).abs().max().item() / yr:.2e}; status {status.read()}", flush=True)
