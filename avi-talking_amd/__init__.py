"""MI355X-native implementation of the AVI-Talking generative hot path
(wav2vec2 audio encoder -> diffusion prior -> per-frame expression/jaw decoders).

Layout: ``csrc/`` HIP kernels + the C-ABI (``include/avi_talking.h``), ``lib.py`` the
ctypes binding, ``host/`` Python mirrors of the reference's model classes.
"""
__version__ = "0.1.0"
