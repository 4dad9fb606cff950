"""MI355X-native implementation of the AVI-Talking generative hot path
(wav2vec2 audio encoder -> diffusion prior -> per-frame expression/jaw decoders).

Layout: ``csrc/`` HIP kernels + the C-ABI (``include/avi_talking.h``), ``lib.py`` the
ctypes binding, ``host/`` Python mirrors of the reference's model classes.
"""
__version__ = "0.1.0"


def _hw_queue_default():
    """The HIP runtime multiplexes every stream of a process onto ``GPU_MAX_HW_QUEUES`` in-order hardware queues (4 unless
    set) and reads that variable when it initialises.  A sampling pass runs four things side by side - the sampler, two
    chains of the audio encoder (host/wav2vec.py) and the previous pass's head - and two of them on one queue serialise
    (measured: the sampler's 10 ms kernel in front of an encoder chain, +1.2 ms per pass).  So unless the caller has set
    the variable, or the GPU is already initialised in this process, ask for 8.  ``HW_QUEUES`` is what this process will
    get as far as the package can know; the hosts only fan out over more than two branches when it is >= 8."""
    import os
    v = os.environ.get("GPU_MAX_HW_QUEUES")
    if v is not None:
        try:
            return int(v)
        except ValueError:
            return 4
    try:
        import torch
        if torch.cuda.is_initialized():
            return 4
    except Exception:       # no torch: the C ABI is used directly, the caller owns the runtime's settings
        return 4
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
    return 8


HW_QUEUES = _hw_queue_default()
