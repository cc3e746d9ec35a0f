"""Post-processing next-row (C++ libamt_post.so) against the Python restatement of the Rust
plugin: bit-exact events, exact stitched floats, hand-built known answers."""
import numpy as np

from audio_to_midi_amd import post
from oracle import post as opost


def _track(rng, frames=300, notes=90, n_notes=25):
    p = rng.uniform(0.0, 0.08, (frames, notes)).astype(np.float32)
    for _ in range(n_notes):
        k = rng.integers(0, notes); s = rng.integers(0, frames - 5); d = rng.integers(2, 80)
        e = min(frames, s + d)
        p[s:e, k] = np.maximum(np.exp(-0.05 * np.arange(e - s)), 0.6) * rng.uniform(0.7, 1.0)
        if rng.random() < 0.4 and e - s > 20:          # re-attack inside a held note
            m = s + (e - s) // 2
            p[m - 1, k] = 0.45
            p[m:e, k] = np.maximum(np.exp(-0.05 * np.arange(e - m)), 0.6) * 0.98
    return p


def test_single_note_known_answer():
    p = np.zeros((60, 90), dtype=np.float32)
    p[10:30, 40] = 0.9
    assert post.extract_events(p) == [(10, 40, 20, 7)]
    assert opost.extract_events(p) == [(10, 40, 20, 7)]
    p[59, 3] = 0.7                                    # still playing at the end: closed with duration >= 1
    assert post.extract_events(p) == [(10, 40, 20, 7), (59, 3, 1, 7)]


def test_extract_events_bit_exact_on_random_tracks():
    rng = np.random.default_rng(0)
    for i in range(6):
        p = _track(rng, frames=int(rng.integers(40, 400)))
        assert post.extract_events(p) == opost.extract_events(p), i
    assert post.extract_events(np.zeros((0, 90), np.float32)) == []
    assert post.extract_events(np.zeros((5, 90), np.float32)) == []


def test_stitch_known_answers_and_exactness():
    const = np.full((4, 250, 90), 0.37, dtype=np.float32)
    s = post.stitch_probs(const, 0.5, 0.02)
    assert s.shape == (4 * 250 - 25 * 3, 90)           # common.rs:20-22
    assert np.allclose(s, 0.37, atol=1e-7)
    rng = np.random.default_rng(1)
    for W, F, ov, dpf in ((3, 250, 0.5, 0.02), (5, 50, 0.2, 0.02), (1, 250, 0.5, 0.02), (2, 40, 0.13, 0.02)):
        a = rng.uniform(0, 1, (W, F, 7)).astype(np.float32)
        assert np.array_equal(post.stitch_probs(a, ov, dpf), opost.stitch_probs(a, ov, dpf))


def test_cbinds_entry_point_matches_two_step_path():
    rng = np.random.default_rng(2)
    wins = np.stack([_track(rng, 250) for _ in range(3)]).astype(np.float16)
    ev = post.extract_midi_events_f16(wins, 0.5, 0.02)
    st = opost.stitch_probs(wins, 0.5, 0.02)
    assert ev == opost.extract_events(st)
    assert len(ev) > 5


def test_frame_events_round_trip():
    events = [(10, 40, 20, 7), (5, 3, 100, 7), (200, 88, 500, 7), (31, 40, 4, 7)]
    a = post.to_frame_events(events, 250, 0, 250)
    b = opost.convert_to_frame_events(events, 250, 0, 250)
    assert np.allclose(a, b, rtol=0, atol=1e-6)      # f32 exp: libm vs numpy differ by <= 1 ulp
    assert (a == 0).tolist() == (b == 0).tolist()
    assert a[10, 40] == 1.0 and a[9, 40] == 0.0 and a[30, 40] == 0.0 and a[249, 88] > 0
    # labels built this way are recovered by extract_events (attack frames and keys)
    got = post.extract_events(np.clip(a, 0.005, 0.995))
    assert got == opost.extract_events(np.clip(a, 0.005, 0.995))
    assert {(10, 40), (5, 3), (200, 88), (31, 40)} <= {(e[0], e[1]) for e in got}


def test_post_library_under_address_sanitizer(tmp_path):
    """SURVEY.md section 5 ("race detection / sanitizers"): an AddressSanitizer host build of the C++ post-processing code
    (csrc/post.cpp) replays the golden vectors and the edge cases of this file in a child process (libasan preloaded, the
    instrumented library loaded through AMT_POST_LIB).  Any heap overflow / use-after-free in stitch_probs,
    extract_events, extract_midi_events / free_midi_events or convert_to_frame_events aborts the child."""
    import os
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gxx = shutil.which("g++")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if gxx is None or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no g++ / libasan in this environment")
    so = tmp_path / "libamt_post_asan.so"
    src = os.path.join(root, "audio-to-midi_amd", "csrc", "post.cpp")
    r = subprocess.run([gxx, "-O1", "-g", "-std=c++17", "-fsanitize=address", "-fno-omit-frame-pointer", "-shared", "-fPIC",
                        "-o", str(so), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    code = r"""
import os, sys
import numpy as np
sys.path.insert(0, %r)
from audio_to_midi_amd import post
g = np.load(os.path.join(%r, "tests", "golden", "post.npz"))
assert np.array_equal(post.stitch_probs(g["windows"], 0.5, 0.02), g["stitched"])
ev = post.extract_events(g["track"])
assert np.array_equal(np.array(ev, dtype=np.int64).reshape(-1, 4), g["events"])
ev16 = post.extract_midi_events_f16(g["windows"].astype(np.float16), 0.5, 0.02)    # the cbinds.rs entry point (f16, stitch + extract)
assert len(ev16) > 0
assert post.extract_events(np.zeros((0, 90), np.float32)) == []                      # empty track
assert post.extract_events(np.full((3, 90), 0.9, np.float32))[0][2] == 3             # note held to the end of a short track
fr = post.to_frame_events(ev[:50], 400, 0, 380)
assert fr.shape == (400, 90)
try:
    post.stitch_probs(np.zeros((2, 10, 90), np.float32), 5.0, 0.02)                   # overlap longer than a window
except IndexError:
    pass
print("ASAN_CHILD_OK")
""" % (root, root)
    env = dict(os.environ, LD_PRELOAD=asan, AMT_POST_LIB=str(so), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "ASAN_CHILD_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "AddressSanitizer" not in r.stderr
