"""Synthetic inputs shaped like the reference's training data.

Audio: stereo 16 kHz, a few decaying piano-frequency sinusoids + noise, normalised to unit
mean-square over both channels and rounded through fp16 (python.rs:251-261, 806-818).
Labels: (S, 90) soft frame events built by the rule of convert_to_frame_events
(python.rs:423-447: max(exp(-0.05 t), 0.6) while held, blank frame before the attack), clamped
to [0.005, 0.995] (label smoothing, python.rs:822-836).
"""
from __future__ import annotations

import numpy as np

SAMPLE_RATE = 16000
SAMPLES_PER_FRAME = 320
N_EVENTS = 90


def make_clip(T: int, rng: np.random.Generator):
    S = T // SAMPLES_PER_FRAME
    t = np.arange(T, dtype=np.float64) / SAMPLE_RATE
    audio = np.zeros((2, T), dtype=np.float64)
    labels = np.zeros((S, N_EVENTS), dtype=np.float32)
    for _ in range(int(rng.integers(1, 7))):
        key = int(rng.integers(0, 88))
        freq = 440.0 * 2.0 ** ((key + 21 - 69) / 12.0)
        onset = float(rng.uniform(0, 0.8 * T / SAMPLE_RATE))
        dur = float(rng.uniform(0.1, 1.5))
        pan = float(rng.uniform(0.3, 0.7))
        amp = float(rng.uniform(0.3, 1.0))
        env = np.where(t >= onset, np.exp(-3.0 * (t - onset)), 0.0) * (t < onset + dur)
        tone = amp * env * np.sin(2 * np.pi * freq * (t - onset))
        audio[0] += pan * tone
        audio[1] += (1 - pan) * tone
        f0 = int(onset * SAMPLE_RATE / SAMPLES_PER_FRAME)
        f1 = min(S, f0 + max(1, int(dur * SAMPLE_RATE / SAMPLES_PER_FRAME)))
        if 0 < f0 < S:
            labels[f0 - 1, key] = 0.0
        for f in range(f0, f1):
            labels[f, key] = max(np.exp(-0.05 * (f - f0)), 0.6)
    audio += rng.normal(0.0, float(rng.uniform(0.0, 0.25)), size=audio.shape)
    audio /= np.sqrt(np.mean(audio ** 2) + 1e-12)
    audio = audio.astype(np.float16).astype(np.float32)
    labels = np.clip(labels, 0.005, 0.995).astype(np.float32)
    return audio, labels


def make_batch(B: int, T: int, seed: int = 42):
    rng = np.random.default_rng(seed)
    a, l = zip(*(make_clip(T, rng) for _ in range(B)))
    return np.stack(a), np.stack(l)


def make_batch_fast(B: int, T: int, seed: int = 42):
    """Cheap generator for large benchmark batches: same statistics (unit mean-square, fp16
    rounded), white noise + one tone per clip; labels uniform in the smoothed range with ~3 %
    active cells."""
    rng = np.random.default_rng(seed)
    S = T // SAMPLES_PER_FRAME
    t = np.arange(T, dtype=np.float32) / SAMPLE_RATE
    audio = rng.normal(0.0, 0.2, size=(B, 2, T)).astype(np.float32)
    freqs = 440.0 * 2.0 ** ((rng.integers(0, 88, size=B) + 21 - 69) / 12.0)
    audio += np.sin(2 * np.pi * freqs[:, None, None].astype(np.float32) * t[None, None, :])
    audio /= np.sqrt(np.mean(audio ** 2, axis=(1, 2), keepdims=True))
    audio = audio.astype(np.float16).astype(np.float32)
    labels = np.full((B, S, N_EVENTS), 0.005, dtype=np.float32)
    active = rng.random((B, S, N_EVENTS)) < 0.03
    labels[active] = rng.uniform(0.6, 0.995, size=int(active.sum())).astype(np.float32)
    return audio, labels
