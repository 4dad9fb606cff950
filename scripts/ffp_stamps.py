"""Where a frame of the persistent FaceFormer decode goes (diagnostic build: AVI_DEFINES=-DAVI_FFP_STAMPS python -c
'import __graft_entry__ as g; g.build()').  Prints the per-stage time per frame seen by thread 0 of workgroup 0 (an
attention + coefficient workgroup) and of workgroup 200 (a plain one): python scripts/ffp_stamps.py [D] [B] [T]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W = importlib.import_module("avi_talking_amd.weights")
from avi_talking_amd.host.faceformer import Faceformer  # noqa: E402

D, B, T = (int(a) for a in (sys.argv[1:4] + ["1024", "1", "250"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
m = Faceformer(W.make_faceformer_weights(2, feature_dim=D), period=30, device=dev)
hs = torch.randn(B, T, D, device=dev)
m.decode(hs)
out = m.decode(hs)
torch.cuda.synchronize()
v = out.reshape(-1)[:64].cpu().tolist()
names = ["wait s3 (prev frame)", "LN3 + q/k/v rows", "attention", "wait+merge partials, o", "out-proj rows", "wait s1",
         "LayerNorm 1+2", "linear1 rows", "wait h", "linear2 rows", "  attention: wait q/k/v", "  attention: scores + softmax",
         "  attention: P.V"]
for wg, base in ((0, 0), (13, 32), (70, 48), (200, 16)):
    tot = sum(v[base:base + 13])
    print(f"workgroup {wg}: {tot / 100 / T:.2f} us per frame")
    for k, n in enumerate(names):
        print(f"   {n:22s} {v[base + k] / 100 / T:6.2f} us")
