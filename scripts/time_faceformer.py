"""Dev tool: FaceFormer decode timing (D, B, T from argv) for rocprofv3 --kernel-trace --stats."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avi_talking_amd import weights as W
from avi_talking_amd.host.faceformer import Faceformer
D, B, T = (int(a) for a in sys.argv[1:4])
dev = torch.device("cuda:0")
m = Faceformer(W.make_faceformer_weights(2, feature_dim=D), period=30, device=dev)
hs = torch.randn(B, T, D, device=dev)
m.decode(hs)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    m.decode(hs)
    torch.cuda.synchronize()
    print(f"D={D} B={B} T={T}: {(time.perf_counter() - t0) * 1e3:.3f} ms  ({(time.perf_counter() - t0) * 1e6 / T:.1f} us/frame)")
