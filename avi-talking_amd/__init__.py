"""MI355X-native implementation of the AVI-Talking generative hot path
(wav2vec2 audio encoder -> diffusion prior -> per-frame expression/jaw decoders).

Layout: ``csrc/`` HIP kernels + the C-ABI (``include/avi_talking.h``), ``lib.py`` the
ctypes binding, ``host/`` Python mirrors of the reference's model classes.
"""
__version__ = "0.1.0"


def _hw_queue_default():
    """What this process will get from the HIP runtime as far as the environment says: ``GPU_MAX_HW_QUEUES`` when it holds
    a number, the runtime's own default (4) otherwise.  Reads only."""
    import os
    v = os.environ.get("GPU_MAX_HW_QUEUES")
    if v is None:
        return 4
    try:
        return int(v)
    except ValueError:
        return 4


HW_QUEUES = _hw_queue_default()


def request_hw_queues(n=8):
    """EXPLICIT opt-in to ``n`` hardware queues (importing the package changes nothing).  The HIP runtime multiplexes every
    stream of a process onto ``GPU_MAX_HW_QUEUES`` in-order hardware queues (4 unless set) and reads that variable when it
    initialises.  A sampling pass runs four things side by side - the sampler, two chains of the audio encoder
    (host/wav2vec.py) and the previous pass's head - and two of them on one queue serialise (measured: the sampler's
    10 ms kernel in front of an encoder chain, +1.2 ms per pass); the hosts only fan out over more than two branches when
    ``HW_QUEUES`` >= 8.  This sets the variable - a process-wide runtime setting, inherited by child processes - only when
    the caller has not set it and the GPU is not initialised yet; programs that own their process call it before their
    first CUDA call (bench.py, tests/conftest.py, __graft_entry__.smoke(), host/cli.py).  Returns the resulting
    ``HW_QUEUES`` and says once on stderr when it changed the environment."""
    global HW_QUEUES
    import os
    import sys
    if os.environ.get("GPU_MAX_HW_QUEUES") is None:
        try:
            import torch
            initialised = torch.cuda.is_initialized()
        except Exception:       # no torch: the C ABI is used directly, the caller owns the runtime's settings
            initialised = True
        if not initialised:
            os.environ["GPU_MAX_HW_QUEUES"] = str(int(n))
            print(f"avi_talking_amd: GPU_MAX_HW_QUEUES={int(n)} requested for this process", file=sys.stderr)
    HW_QUEUES = _hw_queue_default()
    return HW_QUEUES
