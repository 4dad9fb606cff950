"""Quick timing of the HIP wav2vec2 encoder (dev tool; bench.py is the contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avi_talking_amd.weights import make_wav2vec2_weights
from avi_talking_amd.host.wav2vec import Wav2Vec2Model
from avi_talking_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = 160000
dev = torch.device("cuda:0")
w = make_wav2vec2_weights(0)
x = torch.randn(B, N, generator=torch.Generator().manual_seed(5)).to(dev)
for prec, name in ((ops.PREC_BF16X3, "bf16x3"), (ops.PREC_BF16, "bf16")):
    m = Wav2Vec2Model(w, device=dev, prec=prec)
    for _ in range(2):
        o = m(x, "vocaset", frame_num=250)
    torch.cuda.synchronize()
    t = time.time()
    K = 5
    for _ in range(K):
        o = m(x, "vocaset", frame_num=250)
    torch.cuda.synchronize()
    dt = (time.time() - t) / K
    print(f"{name}: {dt*1e3:.2f} ms/forward, {B*250/dt:.0f} frames/s, ~{B*250*0.378e9/dt/1e12:.1f} TFLOP/s algorithmic")
