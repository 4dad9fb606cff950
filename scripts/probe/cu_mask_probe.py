"""Probe: which XCDs / CUs the workgroups of a stream created with hipExtStreamCreateWithCUMask land on, and whether a
hipGraph replayed on such a stream keeps the mask.  Prints, per mask, the histogram of XCC ids of 512 workgroups."""
import ctypes as C
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import avi_talking_amd.lib as L  # noqa: E402

dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
hip = C.CDLL("libamdhip64.so")
so = L.load()


def masked_stream(words):
    s = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def where(stream, blocks=512, lds=140 * 1024, spin=200, graph=False):
    out = torch.zeros(blocks, dtype=torch.int32, device=dev)
    if graph:
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(dev)
        with torch.cuda.stream(side):
            L.check(so.avi_debug_where(out.data_ptr(), blocks, 512, lds, spin, L.stream_ptr()), "where")      # warm
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            L.check(so.avi_debug_where(out.data_ptr(), blocks, 512, lds, spin, L.stream_ptr()), "where")
        out.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            g.replay()
    else:
        with torch.cuda.stream(stream):
            L.check(so.avi_debug_where(out.data_ptr(), blocks, 512, lds, spin, L.stream_ptr()), "where")
    torch.cuda.synchronize()
    v = out.cpu().tolist()
    xcc = Counter((x >> 16) & 0xf for x in v)
    cus = len({(x >> 16, (x >> 8) & 0xf, (x >> 12) & 0xf) for x in v})
    return dict(sorted(xcc.items())), cus


full = [0xffffffff] * 8
print("unmasked stream:", where(torch.cuda.Stream(dev)))
print("first 4 workgroups' XCC ids on an unmasked stream, 3 launches:",
      [[(x >> 16) & 0xf for x in (lambda o: o)(torch.zeros(1))] for _ in range(0)])
for name, words in (("bits 0-127", [0xffffffff] * 4 + [0] * 4), ("bits 128-255", [0] * 4 + [0xffffffff] * 4),
                    ("even bits", [0x55555555] * 8), ("bits with (i % 8) < 4", [0x0f0f0f0f] * 8),
                    ("bits with (i % 8) >= 4", [0xf0f0f0f0] * 8), ("bits 0-31", [0xffffffff] + [0] * 7)):
    st = masked_stream(words)
    print(f"mask {name}: eager {where(st)}  graph replay {where(st, graph=True)}", flush=True)
