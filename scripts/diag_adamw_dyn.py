"""Diagnostic: avi_adamw with its step scalars as kernel arguments against the same launch reading them from the `dyn`
device buffer (what a replayed graph does), three steps, gradients spanning 12 decades."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import avi_talking_amd.lib as L  # noqa: E402

dev = torch.device("cuda:0")
n = 1 << 22
g = torch.Generator().manual_seed(0)
mag = 10.0 ** (torch.rand(n, generator=g) * 12 - 12)
grads = [(torch.randn(n, generator=g) * mag).to(dev) for _ in range(3)]
p0 = torch.randn(n, generator=g).to(dev) * 0.02
so = L.load()
res = {}
for mode in ("args", "dyn", "dyn_blocking"):
    p, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    dyn = torch.zeros(4, device=dev)
    for step, (gr, lr) in enumerate(zip(grads, (1e-3, 1e-3, 2e-3)), 1):
        if mode != "args":
            host = torch.tensor([lr, 1 - 0.9 ** step, 1 / math.sqrt(1 - 0.999 ** step), 0.9])
            dyn.copy_(host, non_blocking=(mode == "dyn"))
            del host
        L.check(so.avi_adamw(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr, 0.9, 0.999, 1e-8, 1e-2, step, 0.5,
                             dyn.data_ptr() if mode != "args" else None, None, None, L.stream_ptr()), "adamw")
    torch.cuda.synchronize()
    res[mode] = (p.clone(), m.clone(), v.clone())
for a, b in (("args", "dyn"), ("args", "dyn_blocking"), ("dyn", "dyn_blocking")):
    d = [(x - y).abs().max().item() for x, y in zip(res[a], res[b])]
    i = int((res[a][0] - res[b][0]).abs().argmax())
    print(f"{a} vs {b}: p {d[0]:.3e} m {d[1]:.3e} v {d[2]:.3e}; worst element: |g| scale {mag[i].item():.2e}, "
          f"grads {[float(x[i]) for x in grads]}, p {float(res[a][0][i]):.8f} vs {float(res[b][0][i]):.8f}")
