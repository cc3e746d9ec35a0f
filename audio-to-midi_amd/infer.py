"""Host glue around the hot path for inference (next-row 3 and 4 of SURVEY.md §8f).

Mirrors, name for name:
  * ``load_and_slice_full_audio``  audio_to_midi_dataset.py:277-294 (+ the RMS normalisation and fp16
    round trip of rust-plugins/src/python.rs:236-264; the reference decodes through an ffmpeg
    subprocess, which is absent here: 16 kHz PCM / float WAV files are read directly)
  * ``predict_and_stitch``         infer.py:37-44
  * ``write_midi_file``            infer.py:46-83 (mido is not installed: a minimal Standard MIDI
    File writer producing the same track content — tempo 120, 4/4, 480 ticks per beat)
  * ``save_checkpoint`` / ``load_checkpoint``  train.py:384-394, infer.py:172-236 (Orbax is not
    readable offline: flat little-endian fp32 parameters in device layout + metadata JSON with the
    reference's ``{model, data_prep}`` keys; ``params.flat_to_equinox`` gives the reference pytree)
"""
from __future__ import annotations

import json
import math
import os
import struct
import wave
from pathlib import Path

import numpy as np

from . import params as _params
from . import post as modelutil
from .rope import precompute_frequencies

SAMPLE_RATE = 16000                # AudioToMidiDatasetLoader.SAMPLE_RATE
MODEL_AUDIO_LENGTH = 5.0           # audio_to_midi_dataset.py:28
MIDI_EVENT_VOCCAB_SIZE = 90        # audio_to_midi_dataset.py:26
NUM_VELOCITY_CATEGORIES = 10       # audio_to_midi_dataset.py:33


def get_data_prep_config():
    """audio_to_midi_dataset.py:41-46"""
    return {"midi_voccab_size": MIDI_EVENT_VOCCAB_SIZE, "max_event_timestamp": MODEL_AUDIO_LENGTH,
            "num_velocity_categories": NUM_VELOCITY_CATEGORIES}


# --------------------------------------------------------------------------- audio in
def read_wav(filename) -> np.ndarray:
    """-> float32 (2, n) in [-1, 1].  16 kHz, mono (duplicated) or stereo, 16/32-bit PCM or 32-bit float."""
    with open(filename, "rb") as f:
        head = f.read(12)
        if head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise ValueError(f"{filename}: not a RIFF/WAVE file")
        fmt = None
        data = None
        while True:
            hdr = f.read(8)
            if len(hdr) < 8:
                break
            cid, size = hdr[:4], struct.unpack("<I", hdr[4:])[0]
            body = f.read(size + (size & 1))[:size]
            if cid == b"fmt ":
                fmt = struct.unpack("<HHIIHH", body[:16])
            elif cid == b"data":
                data = body
    if fmt is None or data is None:
        raise ValueError(f"{filename}: missing fmt/data chunk")
    tag, channels, rate, _, _, bits = fmt
    if rate != SAMPLE_RATE:
        raise ValueError(f"{filename}: sample rate {rate} != {SAMPLE_RATE} (resample first; the reference uses ffmpeg)")
    if tag == 1 and bits == 16:
        x = np.frombuffer(data, dtype="<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        x = np.frombuffer(data, dtype="<i4").astype(np.float32) / 2147483648.0
    elif tag == 3 and bits == 32:
        x = np.frombuffer(data, dtype="<f4").astype(np.float32)
    else:
        raise ValueError(f"{filename}: unsupported WAV encoding (format {tag}, {bits} bit)")
    x = x[: len(x) // channels * channels].reshape(-1, channels).T
    if channels == 1:
        x = np.repeat(x, 2, axis=0)
    elif channels > 2:
        x = x[:2]
    return np.ascontiguousarray(x)


def normalize_audio(samples: np.ndarray) -> np.ndarray:
    """python.rs:236-264: unit mean-square over both channels (skipped for near-silent input),
    then a round trip through fp16."""
    left, right = samples[0].astype(np.float32), samples[1].astype(np.float32)
    total_max = max(float(np.abs(left).max(initial=0.0)), float(np.abs(right).max(initial=0.0)))
    if total_max <= 0.05:
        out = np.stack([left, right])
    else:
        total = float(left.size + right.size)
        variance = float(np.sum(left.astype(np.float64) ** 2 + right.astype(np.float64) ** 2) / total)
        adjustment = math.sqrt(1.0 / variance)
        out = np.stack([left.astype(np.float64) * adjustment, right.astype(np.float64) * adjustment])
    return out.astype(np.float16).astype(np.float32)


def load_full_audio(filename, sample_rate: int = SAMPLE_RATE) -> np.ndarray:
    """modelutil.load_full_audio (python.rs:373-394): (2, n) normalised, fp16-rounded samples."""
    if sample_rate != SAMPLE_RATE:
        raise ValueError("only 16 kHz is supported")
    return normalize_audio(read_wav(filename))


def slice_full_audio(audio_samples: np.ndarray, overlap: float = 0.25):
    """audio_to_midi_dataset.py:281-292 on already-loaded samples."""
    window_size = round(MODEL_AUDIO_LENGTH * SAMPLE_RATE)
    overlap_s = round(overlap * SAMPLE_RATE)
    step = window_size - overlap_s
    n_windows = max(1, math.ceil((audio_samples.shape[1] - overlap_s) / step))
    windows = []
    for i in range(n_windows):
        w = audio_samples[:, i * step: i * step + window_size]
        w = np.pad(w, ((0, 0), (0, window_size - w.shape[1])))
        windows.append(w)
    return np.stack(windows).astype(np.float32), MODEL_AUDIO_LENGTH


def load_and_slice_full_audio(filename, overlap: float = 0.25):
    """AudioToMidiDatasetLoader.load_and_slice_full_audio (audio_to_midi_dataset.py:277-294)."""
    return slice_full_audio(load_full_audio(str(filename), SAMPLE_RATE), overlap)


# --------------------------------------------------------------------------- model out
def stitch_output_probs(all_probs, duration_per_frame: float, overlap: float):
    """infer.py:34-35"""
    return modelutil.stitch_probs(np.stack(all_probs), overlap, duration_per_frame)


def predict_and_stitch(model, state, samples, window_duration: float, overlap: float = 0.0):
    """infer.py:37-44.  The reference precomputes 300 RoPE positions for its 250-frame windows."""
    frames = samples.shape[-1] // 320
    rope_freqs = precompute_frequencies(64, max(300, frames))
    _logits, probs = model.predict(state, samples, rope_freqs)
    probs = probs.float().cpu().numpy().astype(np.float32)
    duration_per_frame = window_duration / probs.shape[1]
    return probs, stitch_output_probs(probs, duration_per_frame, overlap), duration_per_frame


# --------------------------------------------------------------------------- MIDI out
def _varlen(n: int) -> bytes:
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(out))


def midi_track_events(events, duration_per_frame: float, ticks_per_beat: int = 480):
    """The (delta_ticks, kind, key, velocity) list infer.py:58-80 feeds to mido."""
    tempo_us = 500000  # mido.bpm2tempo(120)

    def frame_to_midi_time(frame: int) -> int:   # mido.second2tick
        seconds = frame * duration_per_frame
        scale = tempo_us * 1e-6 / ticks_per_beat
        return int(round(seconds / scale))

    ooo = []
    for attack_frame, key, duration_frame, velocity in events:
        t_on = frame_to_midi_time(attack_frame)
        t_off = frame_to_midi_time(attack_frame + duration_frame)
        midi_key = key + 21
        midi_velocity = int(round((velocity / NUM_VELOCITY_CATEGORIES) * 127))
        ooo.append((t_on, "note_on", midi_key, midi_velocity))
        ooo.append((t_off, "note_off", midi_key, midi_velocity))
    out, current = [], 0
    for time, kind, key, velocity in sorted(ooo):
        out.append((time - current, kind, key, velocity))
        current = time
    return out


def write_midi_file(events, duration_per_frame: float, output_file: str):
    """infer.py:46-83: one track, tempo 120, 4/4, note = key + 21, velocity = round(v / 10 * 127)."""
    ticks_per_beat = 480   # mido.MidiFile default
    track = bytearray()
    track += b"\x00\xff\x51\x03" + (500000).to_bytes(3, "big")                 # set_tempo
    track += b"\x00\xff\x58\x04" + bytes([4, 2, 24, 8])                          # time_signature 4/4
    for delta, kind, key, velocity in midi_track_events(events, duration_per_frame, ticks_per_beat):
        status = 0x90 if kind == "note_on" else 0x80
        track += _varlen(delta) + bytes([status, key & 0x7F, velocity & 0x7F])
    track += b"\x00\xff\x2f\x00"                                                 # end_of_track
    with open(output_file, "wb") as f:
        f.write(b"MThd" + struct.pack(">IHHH", 6, 1, 1, ticks_per_beat))
        f.write(b"MTrk" + struct.pack(">I", len(track)) + bytes(track))


def audio_to_midi(model, path, output=None, overlap: float = 0.5, state=None):
    """The non-validation branch of audio_to_midi.py:33-57."""
    audio_file = Path(path)
    if not audio_file.exists():
        raise FileNotFoundError(f"The specified audio file {audio_file} does not exist!")
    sample_windows, window_duration = load_and_slice_full_audio(audio_file, overlap=overlap)
    individual_probs, stitched_probs, duration_per_frame = predict_and_stitch(
        model, state, sample_windows, window_duration, overlap=overlap)
    events = modelutil.extract_events(stitched_probs)
    if output:
        write_midi_file(events, duration_per_frame, str(output))
    return events, stitched_probs, duration_per_frame


# --------------------------------------------------------------------------- checkpoints
def save_checkpoint(directory, flat_params: np.ndarray, conf: dict, step: int = 0, optimizer=None):
    """Flat weight file (device layout, little-endian fp32) + metadata JSON (model.py:36-41).  ``optimizer``: optional
    ``(m, v, count)`` (OutputSequenceGenerator.optimizer_state) written beside it as ``opt_{step}.f32`` = [m | v], the count
    into the metadata: what ``train(resume=True)`` restores.  Every file is written under a temporary name and renamed, the
    metadata LAST: a crash in between leaves no checkpoint that ``load_checkpoint(step=None)`` would pick and then reject."""
    os.makedirs(directory, exist_ok=True)
    flat = np.ascontiguousarray(flat_params, dtype="<f4")
    if flat.size != _params.param_count(conf):
        raise ValueError("parameter count does not match the configuration")

    def put(name, write):
        tmp = os.path.join(directory, name + ".tmp")
        write(tmp)
        os.replace(tmp, os.path.join(directory, name))

    put(f"params_{step}.f32", lambda t: flat.tofile(t))
    meta = {"model": {k: v for k, v in conf.items()}, "data_prep": get_data_prep_config(), "step": int(step),
            "n_params": int(flat.size), "layout": "device (audio_to_midi_amd.params.device_layout)"}
    if optimizer is not None:
        m, v, count = optimizer
        mv = np.concatenate([np.ascontiguousarray(m, dtype="<f4").reshape(-1), np.ascontiguousarray(v, dtype="<f4").reshape(-1)])
        if mv.size != 2 * flat.size:
            raise ValueError("optimizer moments do not match the parameter count")
        put(f"opt_{step}.f32", lambda t: mv.tofile(t))
        meta["optimizer"] = {"applied_updates": int(count), "file": f"opt_{step}.f32", "layout": "[m | v], device layout"}

    def write_meta(t):
        with open(t, "w") as f:
            json.dump(meta, f, indent=1)
    put(f"metadata_{step}.json", write_meta)


def load_checkpoint(directory, step=None, with_optimizer: bool = False):
    """-> (flat fp32 params, conf, step) [, (m, v, count) or None].  step=None picks the newest COMPLETE checkpoint (one whose
    metadata exists; infer.py:172 load_newest_checkpoint)."""
    if step is None:
        steps = [int(p[len("metadata_"):-len(".json")]) for p in os.listdir(directory)
                 if p.startswith("metadata_") and p.endswith(".json")
                 and os.path.exists(os.path.join(directory, "params_" + p[len("metadata_"):-len(".json")] + ".f32"))]
        if not steps:
            raise FileNotFoundError(f"no checkpoint in {directory}")
        step = max(steps)
    with open(os.path.join(directory, f"metadata_{step}.json")) as f:
        meta = json.load(f)
    flat = np.fromfile(os.path.join(directory, f"params_{step}.f32"), dtype="<f4")
    if flat.size != meta["n_params"] or flat.size != _params.param_count(meta["model"]):
        raise ValueError("checkpoint is inconsistent with its metadata")
    if not with_optimizer:
        return flat, meta["model"], int(step)
    opt = None
    if "optimizer" in meta and os.path.exists(os.path.join(directory, meta["optimizer"]["file"])):
        mv = np.fromfile(os.path.join(directory, meta["optimizer"]["file"]), dtype="<f4")
        if mv.size != 2 * flat.size:
            raise ValueError("optimizer file is inconsistent with the checkpoint")
        opt = (mv[:flat.size], mv[flat.size:], int(meta["optimizer"]["applied_updates"]))
    return flat, meta["model"], int(step), opt
