import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avi_talking_amd.weights import make_prior_weights
from avi_talking_amd.host.diffusion_prior import InstructDiffusionPrior
dev = torch.device("cuda:0")
prior = InstructDiffusionPrior.from_state_dict(make_prior_weights(3), device=dev)
B = 32
te = torch.randn(B, 1, 128, device=dev); noise = torch.randn(101, B, 1, 128, device=dev)
def run(tag, spg):
    f = lambda: prior.p_sample_loop((B, 1, 128), text_cond={"text_embed": te}, noise=noise, samples_per_group=spg)
    for _ in range(2): f()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(3): f()
    torch.cuda.synchronize(); dt = (time.time() - t) / 3
    print(f"{tag}: {dt*1e3:.2f} ms ({dt*1e4:.1f} us/step)", flush=True)
for spg in (1, 2, 4, 5):
    run(f"batched spg={spg}", spg)
buf = torch.zeros(1024 * 1024, dtype=torch.int16, device=dev)
pl = prior.net.planes
for l in range(6):
    lp = pl.layer[l]
    for n in ("qkv_hi", "qkv_lo", "out_hi", "out_lo", "w1_hi", "w1_lo", "w2_hi", "w2_lo"):
        setattr(lp, n, buf.data_ptr())
run("batched spg=4, weights aliased to one 2 MB buffer", 4)
run("batched spg=1, aliased", 1)
