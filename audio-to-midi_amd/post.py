"""Host mirror of the reference's `modelutil` post-processing API (rust-plugins/src/python.rs:962-1005),
backed by libamt_post.so (C++; include/amt_post.h)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib


class _MidiEvent(C.Structure):
    _fields_ = [("attack_time", C.c_uint64), ("note", C.c_uint8), ("duration", C.c_uint64), ("velocity", C.c_uint8)]


class _MidiEventList(C.Structure):
    _fields_ = [("ptr", C.POINTER(_MidiEvent)), ("length", C.c_size_t), ("_capacity", C.c_size_t)]


class _MLMultiArrayWrapper3(C.Structure):
    _fields_ = [("strides", C.c_uint64 * 3), ("dims", C.c_uint64 * 3), ("data", C.c_void_p)]


def _L():
    L = _lib.load_post()
    if not getattr(L, "_amt_typed", False):
        L.amt_stitched_frames.restype = C.c_int64
        L.amt_stitched_frames.argtypes = [C.c_int64, C.c_int64, C.c_double, C.c_double]
        L.amt_stitch_probs.restype = C.c_int32
        L.amt_stitch_probs.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_void_p]
        L.amt_extract_events.restype = C.POINTER(_MidiEventList)
        L.amt_extract_events.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        L.extract_midi_events.restype = C.POINTER(_MidiEventList)
        L.extract_midi_events.argtypes = [_MLMultiArrayWrapper3, C.c_double, C.c_double]
        L.free_midi_events.restype = None
        L.free_midi_events.argtypes = [C.POINTER(_MidiEventList)]
        L.amt_convert_to_frame_events.restype = C.c_int32
        L.amt_convert_to_frame_events.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L._amt_typed = True
    return L


def _events(lst):
    n = lst.contents.length
    out = [(int(lst.contents.ptr[i].attack_time), int(lst.contents.ptr[i].note), int(lst.contents.ptr[i].duration),
            int(lst.contents.ptr[i].velocity)) for i in range(n)]
    _L().free_midi_events(lst)
    return out


def stitch_probs(all_probs, overlap: float, duration_per_frame: float) -> np.ndarray:
    """modelutil.stitch_probs (python.rs:962-969): [windows, frames, events] -> [stitched_frames, events] f32"""
    a = np.ascontiguousarray(all_probs, dtype=np.float32)
    if a.ndim != 3:
        raise TypeError("all_probs must be [windows, frames, events]")
    W, F, E = a.shape
    frames = _L().amt_stitched_frames(W, F, float(overlap), float(duration_per_frame))
    out = np.zeros((max(frames, 0), E), dtype=np.float32)
    if _L().amt_stitch_probs(a.ctypes.data, W, F, E, float(overlap), float(duration_per_frame), out.ctypes.data) != 0:
        raise IndexError("stitch_probs: overlap / duration_per_frame index past the stitched buffer")
    return out


def extract_events(probs):
    """modelutil.extract_events (python.rs:971-978): [frames, notes] -> [(attack, key, duration, velocity)]"""
    a = np.ascontiguousarray(probs, dtype=np.float32)
    if a.ndim != 2:
        raise TypeError("probs must be [frames, notes]")
    return _events(_L().amt_extract_events(a.ctypes.data, a.shape[0], a.shape[1]))


def extract_midi_events_f16(all_probs_f16, overlap: float, duration_per_frame: float):
    """The C entry point of cbinds.rs:51-75 on an f16 [windows, frames, events] array."""
    a = np.ascontiguousarray(all_probs_f16, dtype=np.float16)
    w = _MLMultiArrayWrapper3()
    for i in range(3):
        w.dims[i] = a.shape[i]
        w.strides[i] = a.strides[i] // 2
    w.data = a.ctypes.data
    return _events(_L().extract_midi_events(w, float(overlap), float(duration_per_frame)))


def to_frame_events(events, model_output_size: int, start_frame: int = 0, num_frames_with_backing_samples=None):
    """convert_to_frame_events (python.rs:423-447)"""
    ev = np.ascontiguousarray(np.asarray(events, dtype=np.uint32).reshape(-1, 4))
    out = np.zeros((model_output_size, 90), dtype=np.float32)
    nb = model_output_size if num_frames_with_backing_samples is None else num_frames_with_backing_samples
    if _L().amt_convert_to_frame_events(ev.ctypes.data, ev.shape[0], model_output_size, start_frame, nb,
                                        out.ctypes.data) != 0:
        raise ValueError("event key out of range")
    return out
