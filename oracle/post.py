"""Oracle for the post-processing next-row: pure-Python restatement of the Rust plugin's
stitch_probs / extract_events / convert_to_frame_events.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (the Rust crate has no #[test]; it cannot be built
here: cargo/rustc are absent).  Follows rust-plugins/src/common.rs:13-144 and python.rs:423-447.
"""
import math

import numpy as np


def stitch_probs(all_probs, overlap, duration_per_frame):
    """common.rs:13-45"""
    W, F, E = all_probs.shape
    ovf = float(overlap) / float(duration_per_frame)
    out_frames = W * F - int(ovf) * (W - 1)
    st = np.zeros((out_frames, E), dtype=np.float32)
    base = 0.0
    for w in range(W):
        for f in range(F):
            row = int(base) + f
            for e in range(E):
                p = np.float32(all_probs[w, f, e])
                if w > 0 and f <= math.ceil(ovf):
                    blend = float(f) / ovf
                    st[row, e] = np.float32((1.0 - blend) * float(st[row, e]) + blend * float(p))
                else:
                    st[row, e] = p
        base += float(F) - ovf
    return st


def extract_events(probs):
    """common.rs:47-144 -> sorted list of (attack_frame, key, duration, velocity)"""
    probs = np.asarray(probs, dtype=np.float32)
    num_frames, num_notes = probs.shape
    f32 = np.float32
    events = []
    playing = [None] * num_notes

    def duration(end, start):
        return max(int(end) - int(start), 1)

    def activation_prob(frame, key):
        a = probs[frame, key]
        for i in range(frame + 1, num_frames):
            if probs[i, key] > a:
                a = probs[i, key]
            elif i - frame > 10:
                break
        return a

    for frame in range(num_frames):
        for key in range(num_notes):
            cur = playing[key]
            if cur is not None:
                started_at, ap = cur
                if probs[frame, key] < f32(0.1):
                    events.append((started_at, key, duration(frame, started_at), 7))
                    playing[key] = None
                else:
                    since = f32(frame) - f32(started_at)
                    should = False
                    if since > f32(5.0):
                        prev = f32(0.0)
                        for i in range(frame - 6, frame):
                            prev = f32(prev + probs[i, key])
                        prev = f32(prev / f32(6))
                        nxt = f32(0.0)
                        for i in range(frame, min(frame + 6, num_frames)):
                            nxt = f32(nxt + probs[i, key])
                        nxt = f32(nxt / f32(6))
                        should = f32(nxt - prev) > f32(0.1)
                    if frame < num_frames - 1 and probs[frame, key] < probs[frame + 1, key]:
                        continue
                    if probs[frame, key] > f32(0.4) and should:
                        events.append((started_at, key, duration(frame - 1, started_at), 7))
                        playing[key] = (frame, activation_prob(frame, key))
            elif probs[frame, key] > f32(0.5):
                playing[key] = (frame, activation_prob(frame, key))
    for key in range(num_notes):
        if playing[key] is not None:
            events.append((playing[key][0], key, duration(num_frames, playing[key][0]), 7))
    events.sort()
    return events


def convert_to_frame_events(events, model_output_size, start_frame, num_frames_with_backing_samples):
    """python.rs:423-447"""
    frames = np.zeros((model_output_size, 90), dtype=np.float32)
    for attack, key, dur, _vel in events:
        fs = int(attack) - start_frame
        fe = fs + int(dur)
        if 0 < fs < model_output_size:
            frames[fs - 1, key] = 0.0
        for f in range(max(fs, 0), min(fe, model_output_size, num_frames_with_backing_samples)):
            t = np.float32(f) - np.float32(fs)
            frames[f, key] = max(np.exp(np.float32(-0.05) * t, dtype=np.float32), np.float32(0.6))
    return frames
