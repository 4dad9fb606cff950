"""Device-side failure reports of the library (``include/avi_talking.h`` "Status words"), read WITHOUT a synchronisation.

A launch can meet three failures that only the device sees: a value beyond the fp16 range written into an fp16 activation
plane (the 2-term fp16 GEMM groups of the ``mixed`` / ``f16x2`` precision plans), an fp16 plane tile so small that its lo
plane has gone subnormal, and a paired-sampler workgroup whose partner never answered.  The kernels store 1 into a word of
PINNED HOST memory (plain system-scope stores); the host reads the words at the points it is at anyway - the start of
the next pass, ``check()`` after the caller's own synchronisation - and raises.  The reference's counterpart is its
host-side NaN sweep over the sample dict (inferno/utils/batch.py:22-34), five device synchronisations per forward.

One block of words per process (the library drives one GPU per process).
"""
import torch

from .. import lib as L

F16_OVERFLOW, F16_TINY, PAIR_TIMEOUT, EXCHANGE_TIMEOUT, WORDS = 0, 1, 2, 3, 4
FAULT_PAIR_PARTNER_ABSENT = 1
FAULT_EXCHANGE_ABSENT = 2

_words = None
_view = None       # numpy view of the same pinned memory: reading it dispatches no torch operator at all


class RangeError(RuntimeError):
    """An fp16 activation plane left the range in which the 2-term fp16 GEMM groups keep their accuracy."""


class PairTimeout(RuntimeError):
    """A paired-sampler workgroup gave up on its partner; that pass's style (and coefficients) are NaN."""


class ExchangeTimeout(RuntimeError):
    """A workgroup of the persistent FaceFormer decode never saw a granule it waited for; that decode's output is NaN."""


def words():
    """The process's status words (int32[WORDS], pinned host memory the device writes), registered with the library on
    first use."""
    global _words, _view
    if _words is None:
        w = torch.zeros(WORDS, dtype=torch.int32).pin_memory()
        L.check(L.load().avi_set_status_words(w.data_ptr()), "avi_set_status_words")
        _words, _view = w, w.numpy()
    return _words


def read():
    """(overflow, tiny, pair_timeout) as seen by the host NOW: no synchronisation, so a pass still in flight may not
    have reported yet."""
    words()
    return bool(_view[F16_OVERFLOW]), bool(_view[F16_TINY]), bool(_view[PAIR_TIMEOUT])


def clear():
    words()
    _view[:] = 0


def clear_word(k):
    words()
    _view[k] = 0


def raise_if_set(clear_after=True):
    """Raise for whatever has been reported so far (PairTimeout before RangeError: its results are NaN).  The words are
    cleared first, so one failure raises once and the object stays usable (``clear_after=False`` keeps them)."""
    ovf, tiny, pair = read()
    xch = bool(_view[EXCHANGE_TIMEOUT])
    if not (ovf or tiny or pair or xch):
        return
    if clear_after:
        clear()
    if xch:
        raise ExchangeTimeout("persistent FaceFormer decode: a workgroup never saw a granule it waited for within the bounded "
                              "spin (the launch needs all 256 CUs free); that decode's output is NaN "
                              "(csrc/faceformer_persist.hip).  Faceformer.decode_checked() re-runs such a decode on the "
                              "launch chain; AVI_FF_PERSIST=0 never uses the persistent kernel")
    if pair:
        raise PairTimeout("paired DDPM sampler: a workgroup's partner never answered within the bounded spin; the style of "
                          "that pass is NaN (csrc/prior_pair.hip).  SamplingPipeline.run_checked() re-runs such a batch on "
                          "the unpaired kernel; AVI_PRIOR_PAIR=0 never pairs")
    what = []
    if ovf:
        what.append("a value beyond the fp16 range (|x| >= 65520 or non-finite) was written into an fp16 activation plane: "
                    "the planes hold inf")
    if tiny:
        what.append("an fp16 activation plane tile was below 2^-12 everywhere: its lo plane is subnormal and the 2-term fp16 "
                    "GEMM reading it is less accurate than its gate assumes")
    raise RangeError("; ".join(what) + ".  The activations of this checkpoint / input do not fit the 2-term fp16 GEMM groups "
                     "of the precision plan: use prec='bf16x3' (bf16 planes carry fp32's range), or "
                     "SamplingPipeline.run_checked(), which falls back to it by itself")
