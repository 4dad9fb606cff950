"""Host-side audio helpers of the reference's evaluation entry (row A0 of SURVEY.md 8a):
``read_audio`` / ``process_audio`` of inferno_apps/TalkingHead/evaluation/evaluation_functions.py:680-714.

``librosa`` (the reference's decoder / resampler) is not available here, so ``read_audio`` reads 16 kHz PCM WAV files
with the standard library and refuses anything it would have to resample; everything after the decode follows the
reference: mono, x 32768 -> int16, cut to 22 s, reshape into (T, 640) frames at 25 fps.
"""
import wave

import numpy as np

MAX_SECONDS = 22          # evaluation_functions.py:692-694 (the message says 30 s; the code cuts at 22)


def read_audio(audio_path, sampling_rate=16000):
    """-> (int16 mono samples, sampling_rate).  evaluation_functions.py:680-696."""
    with wave.open(str(audio_path), "rb") as f:
        if f.getframerate() != sampling_rate:
            raise ValueError(f"{audio_path}: {f.getframerate()} Hz; resampling to {sampling_rate} Hz needs librosa "
                             "(not available): convert the file first")
        width, nch, n = f.getsampwidth(), f.getnchannels(), f.getnframes()
        raw = f.readframes(n)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float64) / 32768.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float64) - 128.0) / 128.0
    else:
        raise ValueError(f"{audio_path}: unsupported sample width {width}")
    x = x.reshape(-1, nch)
    if nch > 1:
        x = x.mean(axis=1)                       # librosa.to_mono
    else:
        x = x[:, 0]
    wavdata = (x * 32768.0).astype(np.int16)     # :690
    if wavdata.shape[0] > MAX_SECONDS * sampling_rate:
        wavdata = wavdata[:MAX_SECONDS * sampling_rate]
    return wavdata, sampling_rate


def process_audio(wavdata, sampling_rate=16000, video_fps=25):
    """evaluation_functions.py:699-714: whole frames only, -> {"raw_audio": (T, sampling_rate // fps), "samplerate"}."""
    if sampling_rate % video_fps:
        raise AssertionError("sampling_rate must be a multiple of video_fps")
    wav_per_frame = sampling_rate // video_fps
    wavdata = np.asarray(wavdata)
    num_frames = wavdata.shape[0] // wav_per_frame
    out = np.zeros(num_frames * wav_per_frame, dtype=wavdata.dtype)
    if wavdata.size > out.size:
        out[...] = wavdata[:out.size]
    else:
        out[:wavdata.size] = wavdata
    return {"raw_audio": out.reshape(num_frames, wav_per_frame), "samplerate": sampling_rate}
