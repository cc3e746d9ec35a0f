"""Round 3 ("Next round" item 1): parity of the HIP path at the BENCHMARK shape, on the production dtype.

Full depth (39 Blocks, 16 layers), 10-s clips (S = 500 frames: RoPE beyond the reference's 300-row table, 62 local
windows with 2 + 2 pad rows), train mode with the device's dropout masks rebuilt on the host and fed to the float64
oracle (oracle/torch_model.py autograd), loss + EVERY gradient leaf; two weight sets: layer scale gamma = 1.0 (the CNN
branches contribute) and the reference's own init gamma = 1e-6 (model.py:157-158).

Tolerances.  SURVEY.md 8(c) states: fp32 |dlogit| <= 1e-4; bf16 |dlogit| <= 3e-2, loss rel <= 1e-2.
Measured on MI355X (scratch/parity_probe.py, gpurun_out/r3a/parity_probe.log; B = 2 x 10 s, rms(logit) = 0.54-0.61):

                          |dlogit| max   loss rel   grads, whole vector   worst leaf (norm > 1e-3 of the largest)
  fp32  gamma 1   train     2.4e-6        6e-8          2.6e-7               1.5e-6
  fp32  gamma 1e-6 train    2.1e-6        1e-8          2.6e-7               5.7e-7
  bf16  gamma 1   train     3.7e-2        1.7e-5        9.7e-4               3.2e-2   (layers.0.3.depth_conv.weight, 1.4e-3 of the top norm)
  bf16  gamma 1   eval      3.7e-2        1.3e-5        1.3e-3               4.1e-2
  bf16  gamma 1e-6 train    3.5e-2        9.5e-6        9.6e-4               9.9e-3
  bf16  gamma 1e-6 eval     2.9e-2        4.0e-6        9.8e-4               1.1e-2

so fp32 holds SURVEY's bound with a factor 40 to spare (asserted here: 2e-5), and bf16 holds 5e-2 on the logits - SURVEY's
3e-2 is met only by the gamma = 1e-6 inference pass (2.9e-2); the maximum over 90 000 logits of a 55-layer bf16 network
sits at 3.5-3.7e-2.  The loss and the gradients are far inside their bounds (asserted: loss 1e-3, whole gradient 3e-3,
every leaf with a non-negligible norm 6e-2 - round 2 asserted 8e-2 on a reduced model and skipped nothing else).
The oracle sees the bf16-rounded GEMM weights the device computes with (util.bf16_round_gemm_weights), so the bf16 rows
measure the arithmetic, not the weight rounding.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import audio_to_midi_amd as amt  # noqa: E402
from audio_to_midi_amd import synth  # noqa: E402
from oracle import np_model, post as opost, torch_model  # noqa: E402
from util import bf16_round_gemm_weights, dropout_masks_for  # noqa: E402

#            |dlogit|  loss rel  whole-gradient rel-L2  per-leaf rel-L2 (leaves with norm > 1e-3 of the largest)
SURVEY_BF16_LOGIT_TOL = 3e-2
TOL = {"float32": (2e-5, 1e-6, 2e-6, 2e-5),
       "bfloat16": (SURVEY_BF16_LOGIT_TOL, 1e-3, 1.5e-3, 3e-2)}


def _compare(model, out, loss_ref, logits_ref, grads_ref, dtype):
    tol_logit, tol_loss, tol_vec, tol_leaf = TOL[dtype]
    err = float(np.abs(out.logits.cpu().numpy() - logits_ref).max())
    assert err <= tol_logit, f"logits max abs err {err}"
    assert abs(out.loss - loss_ref) <= tol_loss * abs(loss_ref), (out.loss, loss_ref)
    g = model.grads_to_equinox(out.grads_flat.cpu().numpy())
    assert set(g) == set(grads_ref)
    top = max(float(np.linalg.norm(v)) for v in grads_ref.values())
    num = den = 0.0
    bad = {}
    for k, ref in grads_ref.items():
        d, n = float(np.linalg.norm(g[k] - ref)), float(np.linalg.norm(ref))
        num += d * d; den += n * n
        if n > 1e-3 * top and d > tol_leaf * n:
            bad[k] = (d / n, n / top)
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1][0])[:8]
    assert np.sqrt(num / den) <= tol_vec, np.sqrt(num / den)
    return err


@pytest.mark.parametrize("gamma", [1.0, 1e-6])
@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_full_depth_10s_training_step_matches_the_oracle(dtype, gamma):
    """BASELINE configs[1] shape per clip (full depth, 10 s, train mode), B = 2: loss, logits and every gradient leaf
    against float64 autograd with the same dropout masks (model.py:740-769, train.py:49-62)."""
    conf = np_model.MODEL_CONFIG
    S, B, seed = 500, 2, 20240607
    params = np_model.init_params(conf, seed=1234, gamma=gamma)
    audio, labels = synth.make_batch(B, 320 * S, seed=42)
    model = amt.OutputSequenceGenerator(conf, dtype=dtype)
    model.load_equinox_params(params)
    out = model.loss_and_grads(torch.from_numpy(audio), torch.from_numpy(labels), seed=seed, train=True, return_logits=True)
    assert out.grads_valid
    p_or = params if dtype == "float32" else bf16_round_gemm_weights(params)
    masks = dropout_masks_for(seed, conf, B, S)
    loss_ref, logits_ref, grads_ref = torch_model.loss_and_grads(p_or, audio, labels, conf, dropout_masks=masks)
    _compare(model, out, loss_ref, logits_ref, grads_ref, dtype)


def test_full_depth_10s_inference_meets_the_survey_tolerance_with_the_reference_init():
    """The one case where SURVEY 8(c)'s 3e-2 holds for bf16 at full depth: the reference's gamma = 1e-6 init, inference.
    Stated so that a regression of the bf16 forward shows against the survey's own number (measured: 2.9e-2)."""
    conf = np_model.MODEL_CONFIG
    params = np_model.init_params(conf, seed=1234, gamma=1e-6)
    audio, _ = synth.make_batch(2, 160000, seed=42)
    model = amt.OutputSequenceGenerator(conf, dtype="bfloat16")
    model.load_equinox_params(params)
    logits, _ = model.predict(None, torch.from_numpy(audio))
    pt = torch_model.to_torch(bf16_round_gemm_weights(params), torch.float64)
    ref, _ = torch_model.forward(pt, torch.from_numpy(audio).double(), conf)
    err = float(np.abs(logits.cpu().numpy() - ref.numpy()).max())
    assert err <= SURVEY_BF16_LOGIT_TOL, err


def test_production_qkv_backward_tile_is_pinned_in_fp32(small_conf, monkeypatch):
    """qkv_bwd_fused runs 32-token tiles in bf16 and 16-token tiles in f32, so the f32 parity runs did
    not exercise the production tile shape.  AMT_QKV_BWD_TT=32 forces the 32-token tile in an f32 context (same code,
    same LDS addresses): gradients against float64 autograd at the f32 tolerance, and equal to the 16-token form."""
    conf = dict(small_conf)
    S, B = 77, 3                       # 231 tokens: ragged last tile for both tile lengths
    params = np_model.init_params(conf, seed=21, gamma=1.0)
    audio, labels = synth.make_batch(B, 320 * S, seed=9)
    outs = {}
    for tt in ("16", "32"):
        monkeypatch.setenv("AMT_QKV_BWD_TT", tt)
        model = amt.OutputSequenceGenerator(conf, dtype="float32")
        model.load_equinox_params(params)
        outs[tt] = model.loss_and_grads(torch.from_numpy(audio), torch.from_numpy(labels), seed=5, train=True, return_logits=True)
    monkeypatch.delenv("AMT_QKV_BWD_TT")
    masks = dropout_masks_for(5, conf, B, S)
    loss_ref, logits_ref, grads_ref = torch_model.loss_and_grads(params, audio, labels, conf, dropout_masks=masks)
    _compare(model, outs["32"], loss_ref, logits_ref, grads_ref, "float32")
    assert float((outs["32"].grads_flat - outs["16"].grads_flat).norm() / outs["16"].grads_flat.norm()) < 1e-5


# ------------------------------------------------------------------------------------------------ events, bf16
def _margin_mixed_track(st_ref, st_dev, m):
    """The track extract_events is run on for the reference side of a margin-filtered comparison (events are
    bit-exact "on tracks whose probabilities keep a margin from the thresholds").  A cell takes the DEVICE's value where
    a decision of common.rs:47-144 that reads it is not robust against a perturbation of m per cell, the oracle's value
    everywhere else.  With p = oracle probability, d[f] = mean(p[f .. f+6)) - mean(p[f-6 .. f)):
      * thresholds 0.1 / 0.4 / 0.5 (common.rs:51-54):           |p - t| <= m
      * re-attack test d > 0.1 (common.rs:96-108), which can only fire where p[f] > 0.4:
                                                                 |d - 0.1| <= 2 m and p[f] > 0.4 - m   -> the 12 cells of the two windows
      * "still rising" p[f] < p[f+1] (common.rs:111), which only gates a re-attack:
                                 |p[f] - p[f+1]| <= 2 m and d > 0.1 - 2 m and p[f] > 0.4 - m          -> both cells
    Every decision that reads only unflagged cells comes out the same on both tracks whenever |device - oracle| <= m in
    every cell; a flagged decision reads device values only.  (The activation probability, common.rs:68-84, never
    reaches an event: the velocity is the constant 7.)"""
    F, K = st_ref.shape
    p = st_ref.astype(np.float64)
    near = np.zeros((F, K), dtype=bool)
    for t in (0.5, 0.4, 0.1):
        near |= np.abs(p - t) <= m
    c = np.concatenate([np.zeros((1, K)), np.cumsum(p, axis=0)])
    d = np.full((F, K), -np.inf)
    for f in range(6, F):
        hi = min(f + 6, F)
        d[f] = (c[hi] - c[f]) / 6.0 - (c[f] - c[f - 6]) / 6.0       # / 6 also at the end of the track, as common.rs does
    can_fire = p > 0.4 - m
    win = (np.abs(d - 0.1) <= 2 * m) & can_fire
    for f in np.nonzero(win.any(axis=1))[0]:
        near[f - 6:min(f + 6, F), win[f]] = True
    tie = np.zeros((F, K), dtype=bool)
    tie[:-1] = (np.abs(p[:-1] - p[1:]) <= 2 * m) & (d[:-1] > 0.1 - 2 * m) & can_fire[:-1]
    near |= tie
    near[1:] |= tie[:-1]
    return np.where(near, st_dev, st_ref), near


# per track: (bound on the margin m, bound on the share of cells inside the margin, least unfiltered agreement, least events)
EVENT_BOUNDS = {0.0: (3.0e-2, 0.65, 0.95, 1000), -3.0: (3.0e-2, 0.45, 0.95, 100), -6.0: (3.0e-2, 0.20, 0.95, 20)}


@pytest.mark.parametrize("head_bias", [0.0, -3.0, -6.0])
def test_bf16_events_are_bit_exact_outside_the_bf16_margin(tmp_path, head_bias):
    """BASELINE north_star "bit-exact MIDI events", production dtype: WAV -> windows -> bf16 model ->
    stitch -> extract_events against the float64 oracle + the Python restatement of common.rs.  The margin is DERIVED:
    m = the measured max |p_bf16 - p_f64| on the stitched track (the head weights are scaled x 6 to spread the sigmoids, which
    scales the logit error by 6 as well).  Round 3: m = 4.9e-2 (asserted <= 6e-2), 75-81 % of the dense track's cells inside
    the margin.  Round 4 (residual stream and audio in fp16, csrc/common.h: Res<T>): m = 2.4e-2 / 2.1e-2, asserted <= 3e-2;
    57 % of the cells of the dense track (head bias 0: 4200 events in 700 frames, every key flickering around the
    thresholds - the worst case for a threshold detector), 12 % of the sparse one (bias -6) inside the margin; a third track
    in between (bias -3).  Events of the device track equal, bit for bit, the events of the oracle track with the
    undecidable cells (see _margin_mixed_track) taken from the device; without any filter the two event lists share >= 95 %
    of their events on every track (measured 97 %)."""
    import wave
    from audio_to_midi_amd import infer
    conf = np_model.MODEL_CONFIG
    params = np_model.init_params(conf, seed=1234, gamma=1.0)
    for k in ("decoder.decoder_pooling.weight", "decoder.decoder_pooling.bias"):
        params[k] = params[k] * 6.0
    params["decoder.decoder_pooling.bias"] = params["decoder.decoder_pooling.bias"] + head_bias
    audio, _ = synth.make_batch(1, 12 * 16000, seed=42)
    pcm = np.clip(np.round(audio[0].T / np.abs(audio).max() * 30000.0), -32768, 32767).astype("<i2")
    path = tmp_path / "clip.wav"
    with wave.open(str(path), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
    model = amt.OutputSequenceGenerator(conf, dtype="bfloat16")
    model.load_equinox_params(params)
    events, stitched, dpf = infer.audio_to_midi(model, path, tmp_path / "clip.mid", overlap=0.5)
    windows, _ = infer.load_and_slice_full_audio(path, overlap=0.5)
    pt = torch_model.to_torch(bf16_round_gemm_weights(params), torch.float64)
    _, probs_ref = torch_model.forward(pt, torch.from_numpy(windows).double(), conf)
    st_ref = opost.stitch_probs(probs_ref.numpy().astype(np.float32), 0.5, dpf)
    m_max, near_max, agree_min, n_min = EVENT_BOUNDS[head_bias]
    m = float(np.abs(stitched - st_ref).max())
    assert m <= m_max, m
    mixed, near = _margin_mixed_track(st_ref, stitched, m)
    ev_mixed = opost.extract_events(mixed.astype(np.float32))
    assert len(events) > n_min
    assert events == ev_mixed
    ev_ref = opost.extract_events(st_ref)
    common = len(set(events) & set(ev_ref))
    assert common >= agree_min * max(len(ev_ref), len(events)), (common, len(ev_ref), len(events))
    # how much of the track is decided by the oracle alone (the numbers are in DESIGN.md; a weak bound here)
    print(f"bf16 events, head bias {head_bias}: margin {m:.3e}, cells within the margin {near.mean():.3f}, events {len(events)} "
          f"(oracle {len(ev_ref)}, identical without any filter {common})")
    assert near.mean() < near_max, near.mean()


def _fit_head(params, conf, n_clips=8, seed=300):
    """A detector head that behaves like a trained one WITHOUT a training run: the decoder (LayerNorm + Linear, model.py:185-198)
    is fitted by L-BFGS - the reference's own loss, BCE against the soft labels of python.rs:423-447 - on the float64 oracle's
    decoder inputs for `n_clips` synthetic 5-s clips; everything below the decoder keeps its seeded init.  The result: most
    cells confidently off, held notes confidently on with the label's attack / decay shape - the regime extract_events
    (common.rs:47-144) is built for, unlike the x 6-scaled random head of the test above."""
    rng = np.random.default_rng(seed)
    clips = [synth.make_clip(80000, rng) for _ in range(n_clips)]
    audio = np.stack([c[0] for c in clips])
    labels = np.stack([c[1] for c in clips])
    taps = {}

    def tap(name, x):
        if name == "head.hn":
            taps[name] = x.detach()
        return x
    with torch.no_grad():
        torch_model.forward(torch_model.to_torch(params, torch.float64), torch.from_numpy(audio).double(), conf, tap=tap)
    Z = taps["head.hn"].reshape(-1, taps["head.hn"].shape[-1])                       # [n_clips * 250, 256]
    Y = torch.from_numpy(labels).double().reshape(-1, labels.shape[-1])
    W = torch.zeros(Y.shape[1], Z.shape[1], dtype=torch.float64, requires_grad=True)
    b = torch.full((Y.shape[1],), -4.0, dtype=torch.float64, requires_grad=True)
    opt = torch.optim.LBFGS([W, b], lr=1.0, max_iter=300, history_size=30, line_search_fn="strong_wolfe")

    def closure():
        opt.zero_grad()
        loss = torch.nn.functional.binary_cross_entropy_with_logits(Z @ W.T + b, Y, reduction="mean") + 1e-5 * (W ** 2).sum()
        loss.backward()
        return loss
    opt.step(closure)
    out = dict(params)
    out["decoder.decoder_pooling.weight"] = W.detach().numpy().astype(np.float32)
    out["decoder.decoder_pooling.bias"] = b.detach().numpy().astype(np.float32)
    return out, clips


def test_bf16_events_of_a_fitted_detector_without_any_filter(tmp_path):
    """the bf16 'bit-exact events' claim on a track that looks like a trained model's, with NO margin
    substitution and no head scaling.  The decoder head is FITTED (see _fit_head) on the oracle's features; then, for five 12-s
    recordings - two assembled from fitting clips, three from held-out clips - WAV -> windows -> bf16 device model -> stitch ->
    extract_events is compared with the float64 oracle + the Python restatement of common.rs, event list against event list.
    Asserted: the share of identical events, the share of cells inside the measured bf16 margin of some threshold (what the
    margin-mixed test above substitutes), and the margin itself.  Reference: /root/reference/rust-plugins/src/common.rs:47-144."""
    import wave
    from audio_to_midi_amd import infer
    conf = np_model.MODEL_CONFIG
    base = np_model.init_params(conf, seed=1234, gamma=1.0)
    params, clips = _fit_head(bf16_round_gemm_weights(base), conf)
    # the device gets the fp32 head; a bf16 context multiplies with its bf16 rounding, and so does the oracle below
    dev_params = dict(base)
    dev_params["decoder.decoder_pooling.weight"] = params["decoder.decoder_pooling.weight"]
    dev_params["decoder.decoder_pooling.bias"] = params["decoder.decoder_pooling.bias"]
    model = amt.OutputSequenceGenerator(conf, dtype="bfloat16")
    model.load_equinox_params(dev_params)
    pt = torch_model.to_torch(bf16_round_gemm_weights(dev_params), torch.float64)
    rng = np.random.default_rng(4242)
    held_out = [synth.make_clip(80000, rng) for _ in range(9)]
    recordings = {"fit-a": clips[0:3], "fit-b": clips[3:6], "held-out-a": held_out[0:3], "held-out-b": held_out[3:6], "held-out-c": held_out[6:9]}
    total_ref = total_dev = total_common = 0
    worst_m, near_cells, all_cells = 0.0, 0, 0
    for name, parts in recordings.items():
        audio = np.concatenate([p[0] for p in parts], axis=1)[:, :12 * 16000]
        pcm = np.clip(np.round(audio.T / np.abs(audio).max() * 30000.0), -32768, 32767).astype("<i2")
        path = tmp_path / f"{name}.wav"
        with wave.open(str(path), "wb") as w:
            w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
        events, stitched, dpf = infer.audio_to_midi(model, path, tmp_path / f"{name}.mid", overlap=0.5)
        windows, _ = infer.load_and_slice_full_audio(path, overlap=0.5)
        with torch.no_grad():
            _, probs_ref = torch_model.forward(pt, torch.from_numpy(windows).double(), conf)
        st_ref = opost.stitch_probs(probs_ref.numpy().astype(np.float32), 0.5, dpf)
        ev_ref = opost.extract_events(st_ref)
        m = float(np.abs(stitched - st_ref).max())
        near = np.zeros(st_ref.shape, dtype=bool)
        for t in (0.5, 0.4, 0.1):
            near |= np.abs(st_ref.astype(np.float64) - t) <= m
        common = len(set(events) & set(ev_ref))
        print(f"fitted detector, {name}: margin {m:.3e}, cells within the margin of a threshold {near.mean():.4f}, cells on (p > 0.5) "
              f"{(st_ref > 0.5).mean():.4f}, events device {len(events)} / oracle {len(ev_ref)} / identical {common}")
        total_ref += len(ev_ref); total_dev += len(events); total_common += common
        worst_m = max(worst_m, m); near_cells += int(near.sum()); all_cells += near.size
    share = total_common / max(1, max(total_ref, total_dev))
    print(f"fitted detector, all recordings: identical events {total_common} of {max(total_ref, total_dev)} = {share:.4f}, "
          f"cells within the margin {near_cells / all_cells:.4f}, worst margin {worst_m:.3e}")
    assert total_ref >= 30, total_ref                      # the fitted head does detect notes (measured: 26 in the first three recordings)
    assert share >= 0.97, share
    assert near_cells / all_cells < 0.05
    assert worst_m <= 6e-2, worst_m
