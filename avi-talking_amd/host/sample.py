"""Builders of the sample dict the EMOTE talking head consumes: host-side mirrors of
``create_base_sample`` (inferno_apps/TalkingHead/evaluation/evaluation_functions.py:141-161), ``create_condition``
(:39-57), ``create_high_intensity_emotions`` (:218-275) and ``create_name`` (:163-171), and of the collate the reference's
entry point applies to a list of such samples (``FpParser.recursive_collate``, train_diffusion_prior.py:108-120).

Pure numpy / torch host code: framing of the audio and one-hot conditions; the arithmetic stays in ``TalkingHeadWrapper``.
The reference functions take the Lightning ``talking_head`` object for three numbers and two names; here they are
arguments (defaults = the released EMOTE config: 8 expressions incl. Contempt, 3 intensities, 32 training identities,
reconstruction type "EMICA-MEAD_flame2020" is whatever the caller's checkpoint says).
"""
import copy

import numpy as np
import torch

from .audio_io import process_audio, read_audio

# inferno/datasets/AffectNetDataModule.py:51-63 (AffectNetExpressions): index -> name, as create_name spells them
EMOTION_NAMES = ("Neutral", "Happy", "Sad", "Surprise", "Fear", "Disgust", "Anger", "Contempt", "None_", "Uncertain",
                 "Occluded", "xxx")
CONDITION_KEYS = ("gt_expression_label_condition", "gt_expression_identity_condition", "gt_expression_intensity_condition")


def _one_hot(indices, num_classes):
    idx = np.asarray(indices, dtype=np.int64)
    if idx.ndim != 1 or (idx.size and (idx.min() < 0 or idx.max() >= num_classes)):
        raise ValueError(f"class indices {indices} outside [0, {num_classes})")
    out = np.zeros((idx.shape[0], num_classes), dtype=np.int64)       # torch.nn.functional.one_hot(...).numpy(): int64
    out[np.arange(idx.shape[0]), idx] = 1
    return out


def create_condition(sample, emotions=None, intensities=None, identities=None, n_emotions=8, n_intensities=3,
                     n_identities=32):
    """evaluation_functions.py:39-57: one-hot (len(list), classes) arrays; defaults Neutral / intensity index 2 / identity 0.
    A count of 0 switches that condition off (the config flag ``style_embedding.gt_expression_*`` of the reference)."""
    if n_emotions:
        sample["gt_expression_label_condition"] = _one_hot([0] if emotions is None else emotions, n_emotions)
    if n_intensities:
        sample["gt_expression_intensity_condition"] = _one_hot([2] if intensities is None else intensities, n_intensities)
    if n_identities:
        sample["gt_expression_identity_condition"] = _one_hot([0] if identities is None else identities, n_identities)
    return sample


def create_base_sample(audio, reconstruction_type="EMICA-MEAD_flame2020", smallest_unit=1, silent_frames_start=0,
                       silent_frames_end=0, silence_all=False, **condition_sizes):
    """evaluation_functions.py:141-161.  ``audio``: a path to a 16 kHz PCM WAV, or int16 mono samples already read.
    -> {"raw_audio" (T, 640) int16, "samplerate", "reconstruction": {type: gt_exp (T,50), gt_shape (300), gt_jaw (T,3),
    gt_tex (50)}, the three one-hot conditions}.  As in the reference, the padding line pads BOTH axes of ``raw_audio`` at
    their end by ``smallest_unit - T % smallest_unit`` (np.pad with one (before, after) pair): one extra zero frame AND one
    extra zero sample column at the default ``smallest_unit = 1``."""
    if isinstance(audio, (str, bytes)) or hasattr(audio, "__fspath__"):
        wavdata, sr = read_audio(audio)
    else:
        wavdata, sr = np.asarray(audio), 16000
    sample = process_audio(wavdata, sr, video_fps=25)
    raw = sample["raw_audio"]
    raw = np.pad(raw, (0, smallest_unit - raw.shape[0] % smallest_unit))
    if silent_frames_start > 0:
        raw = np.concatenate([np.zeros((silent_frames_start, raw.shape[1]), dtype=raw.dtype), raw], axis=0)
    if silent_frames_end > 0:
        raw = np.concatenate([raw, np.zeros((silent_frames_end, raw.shape[1]), dtype=raw.dtype)], axis=0)
    if silence_all:
        raw = np.zeros_like(raw)
    sample["raw_audio"] = raw
    T = raw.shape[0]
    sample["reconstruction"] = {reconstruction_type: {
        "gt_exp": np.zeros((T, 50), dtype=np.float32), "gt_shape": np.zeros((300), dtype=np.float32),
        "gt_jaw": np.zeros((T, 3), dtype=np.float32), "gt_tex": np.zeros((50), dtype=np.float32)}}
    return create_condition(sample, **condition_sizes)


def create_name(int_idx, emo_idx, identity_idx, training_subjects):
    """evaluation_functions.py:163-171."""
    return f"_{training_subjects[identity_idx]}_{EMOTION_NAMES[emo_idx]}_{int_idx}"


def create_high_intensity_emotions(sample, identity_list, emotion_index_list=None, intensity_list=None,
                                   silent_frames_start=0, silent_frames_end=0, silent_emotion_start=0, silent_emotion_end=0,
                                   training_subjects=None, n_emotions=8, n_intensities=3, n_identities=32):
    """evaluation_functions.py:218-275: one deep copy of ``sample`` per (emotion, intensity, identity) triple with its
    conditions repeated over the T frames; the first / last ``silent_frames_*`` frames of the emotion label are switched to
    ``silent_emotion_*`` (as written there: with ``silent_frames_end = 0`` the slice ``[-0:]`` is the WHOLE sequence, so the
    label of every frame becomes ``silent_emotion_end`` - the reference's callers pass the default 0 = Neutral... and so does
    the entry point; kept as is)."""
    emotion_index_list = list(range(n_emotions)) if emotion_index_list is None else emotion_index_list
    if intensity_list is None:
        raise TypeError("intensity_list is required (the reference zips over it)")
    samples = []
    for emo_idx, int_idx, identity_idx in zip(emotion_index_list, intensity_list, identity_list):
        s = create_condition(copy.deepcopy(sample), emotions=[emo_idx], identities=[identity_idx], intensities=[int_idx],
                             n_emotions=n_emotions, n_intensities=n_intensities, n_identities=n_identities)
        T = s["raw_audio"].shape[0]
        for key in CONDITION_KEYS:
            cond = s[key]
            if cond.shape[0] == 1:
                cond = cond.repeat(T, axis=0)
                if key == "gt_expression_label_condition":
                    cond[:silent_frames_start] = 0
                    cond[:silent_frames_start, silent_emotion_start] = 1
            s[key] = cond
        for key in CONDITION_KEYS:          # the second loop of the reference never finds shape[0] == 1 again unless T == 1
            cond = s[key]
            if cond.shape[0] == 1:
                cond = cond.repeat(T, axis=0)
                if key == "gt_expression_label_condition":
                    cond[-silent_frames_end:] = 0
                    cond[-silent_frames_end:, silent_emotion_end] = 1
            s[key] = cond
        if training_subjects is not None:
            s["output_name"] = create_name(int_idx, emo_idx, identity_idx, training_subjects)
        samples.append(s)
    return samples


def recursive_collate(batch, device="cuda"):
    """train_diffusion_prior.py:108-120: nested dicts of numpy arrays -> stacked device tensors; other leaves stay lists."""
    if isinstance(batch[0], dict):
        return {k: recursive_collate([item[k] for item in batch], device) for k in batch[0]}
    if isinstance(batch[0], np.ndarray):
        return torch.from_numpy(np.stack(batch)).to(device)
    return batch
