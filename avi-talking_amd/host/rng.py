"""Device-resident random draws for the captured passes (csrc/rng.hip, include/avi_talking.h ``avi_rng_*``).

The reference draws inside its steps with torch generators: ``torch.randn(..., generator=...)`` for x_T and every DDPM step
(models/diffusion_prior.py:337,349-351), and timesteps / q_sample noise / cond-drop masks / dropout masks of a training
step (train_diffusion_prior.py:449 -> models/diffusion_prior.py:445,453,255-259,62-75).  A hipGraph replay cannot call
a torch generator, so those draws come from a Philox-4x32-10 stream whose seed and step offset live in device memory:
``fill`` launches are part of the captured pass and ``advance`` (its last node) moves the offset, so every replay draws
fresh, reproducible numbers.  Parity runs keep injecting recorded tensors (the numbers here are not torch's)."""
import torch

from .. import lib as L

RAW, NORMAL, KEEP_SCALED, BERNOULLI_U8, RANDINT_I32, UNIFORM = range(6)
_DTYPE = {RAW: torch.int32, NORMAL: torch.float32, KEEP_SCALED: torch.float32, BERNOULLI_U8: torch.uint8,
          RANDINT_I32: torch.int32, UNIFORM: torch.float32}


def _s64(x):
    """uint64 value -> the int64 with the same bits (torch has no uint64 tensors to speak of)."""
    x = int(x) & 0xFFFFFFFFFFFFFFFF
    return x - (1 << 64) if x >= (1 << 63) else x


class DeviceRng:
    def __init__(self, seed=0, device="cuda", offset=0):
        self.device = torch.device(device)
        self.state = torch.tensor([_s64(seed), _s64(offset)], dtype=torch.int64, device=self.device)

    def set_state(self, seed=None, offset=None):
        """Reset the stream (a host->device copy on the current stream): replaying a pass from the same (seed, offset)
        reproduces its draws."""
        cur = self.state.tolist()
        new = [cur[0] if seed is None else _s64(seed), cur[1] if offset is None else _s64(offset)]
        self.state.copy_(torch.tensor(new, dtype=torch.int64))
        torch.cuda.current_stream(self.device).synchronize()    # a control operation: later replays on ANY stream see it

    def get_state(self):
        """(seed, offset) as unsigned 64-bit integers."""
        return tuple(v & 0xFFFFFFFFFFFFFFFF for v in self.state.tolist())

    def fill(self, out, kind, subsequence, param=0.0):
        """Fill ``out`` (contiguous, dtype of the kind) on the current stream; ``subsequence`` (< 65536) separates the
        tensors drawn within one step."""
        L.require_gpu(out)
        if out.dtype != _DTYPE[kind] or not out.is_contiguous():
            raise ValueError(f"rng.fill: kind {kind} writes contiguous {_DTYPE[kind]}, got {out.dtype}")
        L.check(L.load().avi_rng_fill(self.state.data_ptr(), int(subsequence), int(kind), float(param), out.numel(),
                                      out.data_ptr(), L.stream_ptr()), "avi_rng_fill")
        return out

    def advance(self, delta=1):
        L.check(L.load().avi_rng_advance(self.state.data_ptr(), int(delta), L.stream_ptr()), "avi_rng_advance")

    # convenience: fresh tensors
    def normal(self, shape, subsequence=0):
        return self.fill(torch.empty(shape, dtype=torch.float32, device=self.device), NORMAL, subsequence)
