"""Host glue for inference: windowing, audio normalisation,
Standard MIDI File writer and checkpoint I/O.  Known answers are derived by hand from the
reference's source text (audio_to_midi_dataset.py:277-294, infer.py:46-83, python.rs:236-264)."""
import math
import struct
import wave

import numpy as np
import pytest

import audio_to_midi_amd as amt
from audio_to_midi_amd import infer, params
from oracle import np_model


def _write_wav(path, x, rate=16000):
    """x: (channels, n) float in [-1, 1] -> 16-bit PCM"""
    pcm = np.clip(np.round(x.T * 32767.0), -32768, 32767).astype("<i2")
    with wave.open(str(path), "wb") as w:
        w.setnchannels(x.shape[0]); w.setsampwidth(2); w.setframerate(rate)
        w.writeframes(pcm.tobytes())
    return pcm.T.astype(np.float32) / 32768.0


def test_read_wav_stereo_mono_and_errors(tmp_path):
    rng = np.random.default_rng(0)
    x = rng.uniform(-0.5, 0.5, (2, 1000))
    ref = _write_wav(tmp_path / "s.wav", x)
    got = infer.read_wav(tmp_path / "s.wav")
    assert got.shape == (2, 1000) and got.dtype == np.float32 and np.array_equal(got, ref)
    ref1 = _write_wav(tmp_path / "m.wav", x[:1])
    got1 = infer.read_wav(tmp_path / "m.wav")
    assert got1.shape == (2, 1000) and np.array_equal(got1[0], ref1[0]) and np.array_equal(got1[1], ref1[0])
    _write_wav(tmp_path / "r.wav", x, rate=44100)
    with pytest.raises(ValueError):
        infer.read_wav(tmp_path / "r.wav")
    (tmp_path / "bad.wav").write_bytes(b"not a wave file at all")
    with pytest.raises(ValueError):
        infer.read_wav(tmp_path / "bad.wav")


def test_normalize_audio_unit_mean_square_and_fp16_round_trip():
    rng = np.random.default_rng(1)
    x = rng.normal(0, 0.2, (2, 4000)).astype(np.float32)
    y = infer.normalize_audio(x)
    assert y.dtype == np.float32 and np.array_equal(y, y.astype(np.float16).astype(np.float32))
    assert abs(float(np.mean(y.astype(np.float64) ** 2)) - 1.0) < 2e-3
    # python.rs:241-251: quiet input (max |x| <= 0.05) is only rounded to fp16, never scaled
    q = (x * 0.01).astype(np.float32)
    assert np.abs(q).max() <= 0.05
    assert np.array_equal(infer.normalize_audio(q), q.astype(np.float16).astype(np.float32))
    # exact restatement of the Rust fold, one sample pair at a time
    total = 2.0 * x.shape[1]
    var = 0.0
    for l, r in zip(x[0], x[1]):
        var += (float(l) ** 2 + float(r) ** 2) / total
    adj = math.sqrt(1.0 / var)
    want = np.stack([(x[0].astype(np.float64) * adj), (x[1].astype(np.float64) * adj)]).astype(np.float16)
    assert np.mean(y != want.astype(np.float32)) < 1e-3   # summation order differs in the last bit only


@pytest.mark.parametrize("seconds,overlap,n_windows", [(5.0, 0.5, 1), (5.01, 0.5, 2), (60.0, 0.5, 14), (12.0, 0.25, 3),
                                                       (9.5, 0.5, 2), (9.51, 0.5, 3)])
def test_slice_full_audio_window_count_and_content(seconds, overlap, n_windows):
    """audio_to_midi_dataset.py:281-292: n = ceil((len - ov) / (80000 - ov)); SURVEY §8d cfg4: 14 windows @60 s"""
    n = round(seconds * 16000)
    x = np.arange(2 * n, dtype=np.float32).reshape(2, n)
    w, dur = infer.slice_full_audio(x, overlap)
    assert dur == 5.0 and w.shape == (n_windows, 2, 80000)
    step = 80000 - round(overlap * 16000)
    for i in range(n_windows):
        seg = x[:, i * step: i * step + 80000]
        assert np.array_equal(w[i, :, :seg.shape[1]], seg)
        assert not w[i, :, seg.shape[1]:].any()       # zero-padded tail


def _parse_smf(data):
    assert data[:4] == b"MThd"
    hlen, fmt, ntrk, tpb = struct.unpack(">IHHH", data[4:14])
    assert data[14:18] == b"MTrk"
    tlen = struct.unpack(">I", data[18:22])[0]
    body = data[22:22 + tlen]
    assert len(data) == 22 + tlen
    i, msgs = 0, []
    while i < len(body):
        delta = 0
        while True:
            b = body[i]; i += 1
            delta = (delta << 7) | (b & 0x7F)
            if not b & 0x80:
                break
        st = body[i]; i += 1
        if st == 0xFF:
            kind, ln = body[i], body[i + 1]
            msgs.append((delta, "meta", kind, bytes(body[i + 2:i + 2 + ln])))
            i += 2 + ln
        else:
            msgs.append((delta, st, body[i], body[i + 1]))
            i += 2
    return (hlen, fmt, ntrk, tpb), msgs


def test_write_midi_file_known_answer(tmp_path):
    """Two notes, 20 ms per frame (5 s / 250 frames), tempo 120 => 1 tick = 1/960 s (infer.py:46-83)."""
    events = [(10, 40, 20, 7), (5, 0, 300, 10)]
    out = tmp_path / "o.mid"
    infer.write_midi_file(events, 5.0 / 250, str(out))
    head, msgs = _parse_smf(out.read_bytes())
    assert head == (6, 1, 1, 480)
    assert msgs[0] == (0, "meta", 0x51, (500000).to_bytes(3, "big"))
    assert msgs[1] == (0, "meta", 0x58, bytes([4, 2, 24, 8]))
    assert msgs[-1] == (0, "meta", 0x2F, b"")
    # frames 5, 10, 30, 305 -> ticks 96, 192, 576, 5856; velocity round(v / 10 * 127)
    assert msgs[2:-1] == [(96, 0x90, 21, 127), (96, 0x90, 61, 89), (384, 0x80, 61, 89), (5280, 0x80, 21, 127)]


def test_midi_track_sort_matches_tuple_order():
    """sorted() on (time, type, key, velocity): at equal times 'note_off' sorts before 'note_on'"""
    ev = infer.midi_track_events([(0, 3, 10, 5), (10, 3, 4, 5)], 0.02)
    assert [e[1] for e in ev] == ["note_on", "note_off", "note_on", "note_off"]
    assert [e[0] for e in ev] == [0, 192, 0, 77]


def test_checkpoint_round_trip(tmp_path, small_conf):
    p = np_model.init_params(small_conf, 5, gamma=1.0)
    flat = params.equinox_to_flat(p, small_conf)
    infer.save_checkpoint(tmp_path, flat, small_conf, step=12)
    infer.save_checkpoint(tmp_path, flat * 2, small_conf, step=40)
    got, conf, step = infer.load_checkpoint(tmp_path)
    assert step == 40 and conf == small_conf and np.array_equal(got, flat * 2)
    got, _, step = infer.load_checkpoint(tmp_path, step=12)
    assert step == 12 and np.array_equal(got, flat)
    back = params.flat_to_equinox(got, conf)
    assert all(np.array_equal(back[k], v.astype(np.float32)) for k, v in p.items())
    with pytest.raises(ValueError):
        infer.save_checkpoint(tmp_path, flat[:-1], small_conf)
    with pytest.raises(FileNotFoundError):
        infer.load_checkpoint(tmp_path / "nope")


def test_checkpoint_is_complete_or_absent_and_carries_the_optimizer_state(tmp_path, small_conf):
    """a checkpoint is written under temporary names and renamed, the metadata last, so the newest one
    `load_checkpoint(step=None)` picks is always complete; AdamW's moments and the count of applied updates travel with
    it (the reference restores its optimizer state through its checkpoint manager, train.py:799-831)."""
    import os
    from audio_to_midi_amd import params as P
    n = P.param_count(small_conf)
    flat = np.arange(n, dtype=np.float32) * 1e-6
    infer.save_checkpoint(tmp_path, flat, small_conf, step=20, optimizer=(flat + 1, flat + 2, 17))
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".tmp")]
    got, conf, step, opt = infer.load_checkpoint(tmp_path, with_optimizer=True)
    assert step == 20 and np.array_equal(got, flat) and opt[2] == 17
    assert np.array_equal(opt[0], flat + 1) and np.array_equal(opt[1], flat + 2)
    # a crash between the weight file and the metadata of a LATER step: that step is not a checkpoint
    (flat * 3).astype("<f4").tofile(os.path.join(tmp_path, "params_40.f32"))
    got, _, step = infer.load_checkpoint(tmp_path)
    assert step == 20 and np.array_equal(got, flat)
    # a checkpoint without optimizer state loads with opt = None
    infer.save_checkpoint(tmp_path, flat, small_conf, step=60)
    assert infer.load_checkpoint(tmp_path, with_optimizer=True)[3] is None
    with pytest.raises(ValueError):
        infer.save_checkpoint(tmp_path, flat, small_conf, step=80, optimizer=(flat[:-1], flat, 1))
