"""End-to-end parity of the HIP path (through the C ABI) against the oracle on the same seeded
inputs: frame logits, loss, every gradient leaf; fp32 verification mode and bf16 production mode.

Tolerances (stated per SURVEY.md §8c):
  fp32 HIP vs float64 oracle : |dlogit| <= 2e-4,  loss rel <= 1e-5, per-leaf grad rel-L2 <= 2e-3
  bf16 HIP vs float64 oracle : |dlogit| <= 6e-2 (reduced model) / 1e-1 (full depth),
                               loss rel <= 1e-2, per-leaf grad rel-L2 <= 8e-2 (leaves with a
                               non-negligible norm)
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import audio_to_midi_amd as amt  # noqa: E402
from audio_to_midi_amd import params as aparams  # noqa: E402
from audio_to_midi_amd import synth  # noqa: E402
from oracle import np_model, torch_model  # noqa: E402
from util import keep_mask  # noqa: E402


def _grad_report(model, grads_flat, grads_ref):
    g = model.grads_to_equinox(grads_flat.cpu().numpy())
    rep = {}
    gmax = max(float(np.linalg.norm(v)) for v in grads_ref.values())
    for k, ref in grads_ref.items():
        n = float(np.linalg.norm(ref))
        rep[k] = (float(np.linalg.norm(g[k] - ref)) / (n + 1e-30), n / gmax)
    return rep


@pytest.mark.parametrize("dtype,S", [("float32", 50), ("float32", 32), ("bfloat16", 50)])
def test_reduced_model_forward_backward(small_conf, dtype, S):
    conf = small_conf
    T, B = 320 * S, 2
    params = np_model.init_params(conf, seed=7, gamma=1.0)
    audio, labels = synth.make_batch(B, T, seed=3)
    model = amt.OutputSequenceGenerator(conf, dtype=dtype)
    model.load_equinox_params(params)
    # inference entry point == training entry point with train=False
    logits_p, probs_p = model.predict(None, torch.from_numpy(audio))
    out = model.loss_and_grads(torch.from_numpy(audio), torch.from_numpy(labels), scale=1.0, train=False,
                               return_logits=True)
    assert out.grads_valid
    if dtype == "bfloat16":   # the oracle sees the same bf16-rounded parameters the device computes with
        params = torch_model.round_gemm_weights(params)
    loss_ref, logits_ref, grads_ref = torch_model.loss_and_grads(params, audio, labels, conf)
    logits = out.logits.cpu().numpy()
    assert np.array_equal(logits, logits_p.cpu().numpy())
    assert np.allclose(probs_p.cpu().numpy(), 1 / (1 + np.exp(-logits)), atol=1e-6)
    tol_logit, tol_loss, tol_grad = (2e-4, 1e-5, 2e-3) if dtype == "float32" else (3e-2, 1e-3, 4e-2)
    err = np.abs(logits - logits_ref).max()
    assert err <= tol_logit, f"logits max abs err {err}"
    assert abs(out.loss - loss_ref) <= tol_loss * abs(loss_ref), (out.loss, loss_ref)
    rep = _grad_report(model, out.grads_flat, grads_ref)
    bad = {k: v for k, v in rep.items() if v[0] > tol_grad and v[1] > 1e-3}
    assert not bad, f"gradient leaves off: {sorted(bad.items(), key=lambda kv: -kv[1][0])[:8]}"
    # loss scaling is transparent (train.py:319): grads come back unscaled
    out2 = model.loss_and_grads(torch.from_numpy(audio), torch.from_numpy(labels), scale=8.0, train=False)
    assert abs(out2.scaled_loss - 8 * out.loss) <= 1e-4 * abs(8 * out.loss)
    d = (out2.grads_flat - out.grads_flat).norm() / out.grads_flat.norm()
    assert float(d) < (1e-4 if dtype == "float32" else 3e-2)


def test_dropout_training_step_matches_oracle_with_same_masks(small_conf):
    """train=True: attention-weight + FFN-output dropout (model.py:237,255) with the device's
    counter-based masks rebuilt on the host and fed to the oracle."""
    conf = small_conf
    S, B, H = 50, 2, 4
    T = 320 * S
    params = np_model.init_params(conf, seed=11, gamma=1.0)
    audio, labels = synth.make_batch(B, T, seed=5)
    model = amt.OutputSequenceGenerator(conf, dtype="float32")
    model.load_equinox_params(params)
    seed = 20240607
    out = model.loss_and_grads(torch.from_numpy(audio), torch.from_numpy(labels), seed=seed, train=True,
                               return_logits=True)
    p = conf["transformer_dropout_rate"]
    pl, pr, nwin, _ = np_model.local_window_geometry(S)
    masks = []
    for li in range(2 * conf["num_transformer_layers"]):
        if li % 2 == 0:
            k_, kp = keep_mask(seed, 2 * li, np.arange(B * H * nwin * 256).reshape(B, H, nwin, 16, 16), p)
            attn = torch.tensor(k_).permute(0, 2, 1, 3, 4)
        else:
            k_, kp = keep_mask(seed, 2 * li, np.arange(B * H * S * S).reshape(B, H, S, S), p)
            attn = torch.tensor(k_)
        f_, _ = keep_mask(seed, 2 * li + 1, np.arange(B * S * 256).reshape(B, S, 256), p)
        masks.append(dict(attn=attn, ffn=torch.tensor(f_), keep=kp))
    loss_ref, logits_ref, grads_ref = torch_model.loss_and_grads(params, audio, labels, conf, dropout_masks=masks)
    assert np.abs(out.logits.cpu().numpy() - logits_ref).max() <= 2e-4
    assert abs(out.loss - loss_ref) <= 1e-5 * abs(loss_ref)
    rep = _grad_report(model, out.grads_flat, grads_ref)
    bad = {k: v for k, v in rep.items() if v[0] > 2e-3 and v[1] > 1e-3}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1][0])[:8]
    # a different seed gives a different (but equally valid) step; inference ignores the seed
    out_b = model.loss_and_grads(torch.from_numpy(audio), torch.from_numpy(labels), seed=seed + 1, train=True)
    assert out_b.loss != out.loss


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_full_model_forward_5s(dtype):
    """BASELINE configs[0] shape: one 5 s clip (2, 80000) -> (250, 90), full depth."""
    conf = np_model.MODEL_CONFIG
    params = np_model.init_params(conf, seed=1234, gamma=1.0)
    audio, _ = synth.make_batch(1, 80000, seed=42)
    model = amt.OutputSequenceGenerator(conf, dtype=dtype)
    model.load_equinox_params(params)
    logits, probs = model.predict(None, torch.from_numpy(audio))
    # (bf16: the oracle multiplies with the bf16-rounded GEMM weights the context multiplies with)
    pt = torch_model.to_torch(params if dtype == "float32" else torch_model.round_gemm_weights(params), torch.float64)
    ref, _ = torch_model.forward(pt, torch.from_numpy(audio).double(), conf)
    err = np.abs(logits.cpu().numpy() - ref.numpy()).max()
    assert logits.shape == (1, 250, 90)
    assert err <= (5e-4 if dtype == "float32" else 3e-2), err      # 1e-4 (5 s: 5e-4 asserted) / 3e-2


def test_size_independent_properties_at_bench_size():
    """cfg2 shape (B=8 x 10 s, bf16): properties that need no oracle at this size."""
    conf = np_model.MODEL_CONFIG
    model = amt.OutputSequenceGenerator(conf, key=1234, dtype="bfloat16", gamma=1.0)
    audio, labels = synth.make_batch_fast(8, 160000, seed=1)
    a, y = torch.from_numpy(audio), torch.from_numpy(labels)
    o1 = model.loss_and_grads(a, y, seed=3, train=True, return_logits=True)
    assert o1.grads_valid and np.isfinite(o1.loss)
    # clips are independent (the reference vmaps the model, train.py:56-58): permuting the batch
    # permutes the logits and leaves the loss unchanged
    perm = torch.tensor([3, 1, 7, 0, 2, 6, 5, 4])
    o2 = model.loss_and_grads(a[perm], y[perm], train=False, return_logits=True)
    o3 = model.loss_and_grads(a, y, train=False, return_logits=True)
    assert torch.equal(o2.logits, o3.logits[perm])
    assert abs(o2.loss - o3.loss) <= 1e-5 * abs(o3.loss)
    # forward is deterministic; gradients are reproducible up to atomic summation order
    o4 = model.loss_and_grads(a, y, train=False, return_logits=True)
    assert torch.equal(o3.logits, o4.logits)
    assert float((o3.grads_flat - o4.grads_flat).norm() / o3.grads_flat.norm()) < 1e-5
    # the training step (dropout masks are a pure function of the seed) is reproducible as well
    t1 = model.loss_and_grads(a, y, seed=9, train=True)
    t2 = model.loss_and_grads(a, y, seed=9, train=True)
    assert float((t1.grads_flat - t2.grads_flat).norm() / t1.grads_flat.norm()) < 1e-5
    # loss equals the BCE of the returned logits
    z = o3.logits.double()
    l = torch.nn.functional.binary_cross_entropy_with_logits(z, y.cuda().double(), reduction="sum") / 8
    assert abs(float(l) - o3.loss) <= 1e-4 * abs(o3.loss)


def test_error_behaviour():
    model = amt.OutputSequenceGenerator(np_model.MODEL_CONFIG, dtype="bfloat16")
    with pytest.raises(ValueError):
        model.predict(None, torch.zeros(1, 1, 6400))        # mono
    with pytest.raises(ValueError):
        model.predict(None, torch.zeros(1, 2, 6401))        # not a whole number of frames
    with pytest.raises(ValueError):
        model.loss_and_grads(torch.zeros(2, 2, 6400), torch.zeros(2, 21, 90))
    with pytest.raises(ValueError):                           # reference: 300-row table vs 500 frames
        model.predict(None, torch.zeros(1, 2, 160000), amt.precompute_frequencies(64, 300))
    bad = dict(np_model.MODEL_CONFIG); bad["dims"] = [8, 16, 32, 64, 128, 256, 512]
    with pytest.raises(ValueError):
        amt.OutputSequenceGenerator(bad)


def test_cfg1_wav_to_midi_events_bit_exact(tmp_path):
    """BASELINE configs[0]: WAV -> windows -> model -> stitch -> extract_events -> .mid
    (audio_to_midi.py:33-57).  fp32 device path vs the float64 oracle forward + the Python restatement
    of the Rust post-processing: stitched probabilities within 2e-4, events bit-exact on the
    margin-filtered track (thresholds 0.5 / 0.4 / 0.1, common.rs:51-54)."""
    import wave
    from audio_to_midi_amd import infer
    from oracle import post as opost
    conf = np_model.MODEL_CONFIG
    params = np_model.init_params(conf, seed=1234, gamma=1.0)
    for k in ("decoder.decoder_pooling.weight", "decoder.decoder_pooling.bias"):
        params[k] = params[k] * 6.0     # spread the sigmoids over (0, 1) so that notes start and stop
    audio, _ = synth.make_batch(1, 12 * 16000, seed=42)
    pcm = np.clip(np.round(audio[0].T / np.abs(audio).max() * 30000.0), -32768, 32767).astype("<i2")
    path = tmp_path / "clip.wav"
    with wave.open(str(path), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
    model = amt.OutputSequenceGenerator(conf, dtype="float32")
    model.load_equinox_params(params)
    out = tmp_path / "clip.mid"
    events, stitched, dpf = infer.audio_to_midi(model, path, out, overlap=0.5)
    # oracle path on the same windows
    windows, dur = infer.load_and_slice_full_audio(path, overlap=0.5)
    assert windows.shape == (3, 2, 80000) and dpf == 5.0 / 250
    pt = torch_model.to_torch(params, torch.float64)
    _, probs_ref = torch_model.forward(pt, torch.from_numpy(windows).double(), conf)
    probs_ref = probs_ref.numpy().astype(np.float32)
    st_ref = opost.stitch_probs(probs_ref, 0.5, dpf)
    assert stitched.shape == st_ref.shape == (250 * 3 - 25 * 2, 90)
    tol = 2e-4
    assert np.abs(stitched - st_ref).max() <= tol
    # margin filter: cells whose oracle probability lies within the tolerance of a
    # threshold cannot be decided by either side; they take the device's value (which is within the
    # tolerance of the oracle's), every other cell is the oracle's own
    near = np.zeros(st_ref.shape, dtype=bool)
    for t in (0.5, 0.4, 0.1):
        near |= np.abs(st_ref - t) <= tol
    assert near.mean() < 2e-3
    st_ref = np.where(near, stitched, st_ref)
    ev_ref = opost.extract_events(st_ref)
    assert len(ev_ref) > 10
    assert events == ev_ref
    data = out.read_bytes()
    assert data[:4] == b"MThd" and data[14:18] == b"MTrk" and data[-4:] == b"\x00\xff\x2f\x00"
    assert len(data) == 22 + int.from_bytes(data[18:22], "big")


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_chain_kernels_equal_unfused_launch_sequence(small_conf, dtype, monkeypatch):
    """csrc/chain.hip evaluates whole operator chains per token tile with the roundings and the
    accumulation order of the unfused launches (AMT_FUSE=0): logits and loss agree bit for bit,
    gradients up to the atomic summation order of the weight-gradient kernels (float32) / isolated
    bf16 roundings (bfloat16)."""
    conf = dict(small_conf)
    conf["depths"] = [1, 1, 2, 1, 1, 2, 1]
    params = np_model.init_params(conf, seed=21, gamma=1.0)
    audio, labels = synth.make_batch(3, 320 * 77, seed=9)      # 231 frames: ragged last tiles everywhere
    outs = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("AMT_FUSE", fuse)
        model = amt.OutputSequenceGenerator(conf, dtype=dtype)
        model.load_equinox_params(params)
        outs.append(model.loss_and_grads(torch.from_numpy(audio), torch.from_numpy(labels), seed=5, train=True,
                                         return_logits=True))
        model.close()
    a, b = outs
    assert torch.equal(a.logits, b.logits)
    assert abs(a.loss - b.loss) <= 1e-6 * abs(b.loss)      # summed with float atomics
    rel = float((a.grads_flat - b.grads_flat).norm() / b.grads_flat.norm())
    # float32: both paths agree to the summation-order noise of the weight-gradient atomics.  bfloat16: the
    # forward is bit-identical; in the backward the transformer chains differ from the unfused launches by
    # isolated 1-ulp bf16 roundings of the residual gradient (fp32 values equal to ~1e-7, a different
    # side of a rounding boundary once in ~1e5 elements), which the layers below amplify to ~2e-4.
    assert rel < (1e-5 if dtype == "float32" else 1e-3), rel


def test_hipgraph_replay_reproduces_the_eager_step(small_conf):
    """The training step is captured once and replayed (bench.py, train.py).  Every replay must
    reproduce the eager step: captured hipMemsetAsync nodes did not on ROCm 7.2 (gradients
    accumulated from the second replay on), which is why the step zeroes its buffers with kernels."""
    model = amt.OutputSequenceGenerator(small_conf, key=3, dtype="bfloat16", gamma=1.0)
    audio, labels = synth.make_batch(4, 320 * 100, seed=2)
    a, y = torch.from_numpy(audio).to("cuda", torch.bfloat16), torch.from_numpy(labels).cuda()
    gbuf = torch.full((model.n_params,), 7.0, dtype=torch.float32, device="cuda")

    def step():
        return model.loss_and_grads(a, y, seed=5, train=True, grads_out=gbuf, sync=False)

    step(); torch.cuda.synchronize()
    ref = gbuf.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    loss_ref = float(step()[0][0]); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    keep = []
    with torch.cuda.graph(g):
        keep.append(step())
    for _ in range(4):
        gbuf.fill_(123.0)                      # whatever the buffer holds must not leak into the result
        keep[0][0].fill_(-1.0)                 # (loss, scaled_loss) tensor written by the captured step
        g.replay()
        torch.cuda.synchronize()
        assert float((gbuf - ref).norm() / ref.norm()) < 1e-5
        assert abs(float(keep[0][0][0]) - loss_ref) <= 1e-5 * abs(loss_ref)      # the loss is an atomic sum: last bits only


def test_two_phase_step_equals_one_call(small_conf):
    """amt_forward_backward_phase 1 + 2 (the data-parallel overlap form) issues the launches of
    amt_forward_backward: same loss, same gradients; after phase 1 alone the decoder / transformer /
    final-norm range is already final and the CNN range of the caller's buffer has not been touched."""
    model = amt.OutputSequenceGenerator(small_conf, key=4, dtype="bfloat16", gamma=1.0)
    audio, labels = synth.make_batch(3, 320 * 90, seed=6)
    a = torch.from_numpy(audio).to("cuda", torch.bfloat16)
    y = torch.from_numpy(labels).cuda()
    ref = model.loss_and_grads(a, y, seed=11, train=True)
    split = model.grad_bucket_split()
    assert 0 < split < model.n_params
    g = torch.full((model.n_params,), 3.0, dtype=torch.float32, device="cuda")
    loss = torch.zeros(2, device="cuda")
    valid = torch.zeros(1, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):       # the phase entry point converts nothing: bf16 samples are not its input type
        model.loss_and_grads_phase(1, a, y, g, loss, valid, seed=11, train=True)
    a = a.to(model.audio_dtype)
    model.loss_and_grads_phase(1, a, y, g, loss, valid, seed=11, train=True)
    torch.cuda.synchronize()
    assert float((g[:split] - 3.0).abs().max()) == 0.0       # phase 2 writes this range
    assert float((g[split:] - ref.grads_flat[split:]).norm() / ref.grads_flat[split:].norm()) < 1e-5
    assert abs(float(loss[0]) - ref.loss) <= 1e-6 * abs(ref.loss) and int(valid.item()) == 1
    model.loss_and_grads_phase(2, a, y, g, loss, valid, seed=11, train=True)
    torch.cuda.synchronize()
    assert float((g - ref.grads_flat).norm() / ref.grads_flat.norm()) < 1e-5
    assert int(valid.item()) == 1
    with pytest.raises(RuntimeError):
        model.loss_and_grads_phase(4, a, y, g, loss, valid)
    with pytest.raises(ValueError):       # phase 3 (whole step + first-half signal) is the fused protocol's: no caller buffer
        model.loss_and_grads_phase(3, a, y, g, loss, valid)


def test_streaming_inference_windows_are_independent():
    """cfg4 shape (448 windows x 5 s in one call): every window's logits equal the logits of the same window
    predicted in a small batch (the reference vmaps the model over windows, infer.py:40), and the stitched
    track of a clip has n * 250 - 25 * (n - 1) frames (common.rs:20-22)."""
    from audio_to_midi_amd import infer
    model = amt.OutputSequenceGenerator(np_model.MODEL_CONFIG, key=1234, dtype="bfloat16", gamma=1.0)
    rng = np.random.default_rng(3)
    full = rng.standard_normal((2, 60 * 16000)).astype(np.float32)
    windows, dur = infer.slice_full_audio(infer.normalize_audio(full), 0.5)
    assert windows.shape == (14, 2, 80000)
    batch = torch.from_numpy(np.tile(windows, (32, 1, 1)))
    logits, probs = model.predict(None, batch)
    assert logits.shape == (448, 250, 90)
    small, _ = model.predict(None, batch[[0, 13, 447]])
    assert torch.equal(logits[[0, 13, 447]], small)
    assert torch.equal(logits[:14], logits[14:28])                 # identical clips -> identical windows
    stitched = infer.stitch_output_probs(probs[:14].cpu().numpy(), dur / 250, 0.5)
    assert stitched.shape == (14 * 250 - 25 * 13, 90)


def test_fp8_attention_step_stays_close_to_the_bf16_step():
    """BASELINE configs[4] (fp8 MFMA attention, bf16 everything else) at full depth: with amt_set_attn_fp8 the
    loss and the gradients of the step stay within fp8 rounding of the bf16 step (measured: loss 3e-5 relative,
    gradients 1e-3 relative L2, logits 0.024 absolute on an rms of 0.54), on the first call (default scale)
    and on later ones (delayed scaling); switching it off restores the bf16 step exactly; fp32 contexts refuse."""
    ref = amt.OutputSequenceGenerator(np_model.MODEL_CONFIG, key=1234, dtype="bfloat16", gamma=1.0)
    f8 = amt.OutputSequenceGenerator(np_model.MODEL_CONFIG, key=1234, dtype="bfloat16", gamma=1.0)
    f8.set_attn_fp8(True)
    audio, labels = synth.make_batch(2, 80000, seed=6)
    a = torch.from_numpy(audio).to("cuda", torch.bfloat16)
    y = torch.from_numpy(labels).cuda()
    r = ref.loss_and_grads(a, y, seed=11, train=True)
    for it in range(2):
        q = f8.loss_and_grads(a, y, seed=11, train=True)
        assert q.grads_valid
        assert abs(q.loss - r.loss) < 5e-4 * abs(r.loss), (it, q.loss, r.loss)
        gd = float((q.grads_flat - r.grads_flat).norm() / r.grads_flat.norm())
        assert 0.0 < gd < 1e-2, (it, gd)
    lr, _ = ref.predict(None, torch.from_numpy(audio))
    lq, _ = f8.predict(None, torch.from_numpy(audio))
    assert float((lr - lq).abs().max()) < 0.1
    f8.set_attn_fp8(False)
    q = f8.loss_and_grads(a, y, seed=11, train=True)
    assert float((q.grads_flat - r.grads_flat).norm() / r.grads_flat.norm()) < 1e-5
    f32m = amt.OutputSequenceGenerator(np_model.MODEL_CONFIG, key=1, dtype="float32", gamma=1.0)
    with pytest.raises(RuntimeError):
        f32m.set_attn_fp8(True)


def test_gradient_buffer_alignment_and_fp8_under_graph_replay(small_conf):
    """(1) The caller's gradient pointer may have any 4-byte alignment: the library accumulates in its own aligned
    buffer and the last pass writes the caller's (csrc/model.hip: run_backward), so shifted views give the same
    gradients.  (2) The fp8 attention switch works inside a captured step: the |max| roll is a kernel of the step,
    so replays keep updating the delayed scales and stay finite and close to the eager fp8 step."""
    model = amt.OutputSequenceGenerator(small_conf, key=4, dtype="bfloat16", gamma=1.0)
    audio, labels = synth.make_batch(2, 320 * 64, seed=8)
    a = torch.from_numpy(audio).to("cuda", torch.bfloat16)
    y = torch.from_numpy(labels).cuda()
    ref = model.loss_and_grads(a, y, seed=5, train=True)
    big = torch.zeros(model.n_params + 8, dtype=torch.float32, device="cuda")
    for off in (1, 4, 7):
        big.zero_()
        view = big[off: off + model.n_params]
        out = model.loss_and_grads(a, y, seed=5, train=True, grads_out=view)
        assert float((out.grads_flat - ref.grads_flat).norm() / ref.grads_flat.norm()) < 1e-5, off
        assert float(big[:off].abs().max()) == 0.0 and float(big[off + model.n_params:].abs().max()) == 0.0

    model.set_attn_fp8(True)
    eager = model.loss_and_grads(a, y, seed=5, train=True)
    assert eager.grads_valid
    g = torch.zeros(model.n_params, dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        model.loss_and_grads(a, y, seed=5, train=True, grads_out=g, sync=False)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        model.loss_and_grads(a, y, seed=5, train=True, grads_out=g, sync=False)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(g).all())
    assert float((g - eager.grads_flat).norm() / eager.grads_flat.norm()) < 2e-2
    assert float((g - ref.grads_flat).norm() / ref.grads_flat.norm()) < 5e-2


# ------------------------------------------------------------------ round 2: device gate, device seed, re-planning, teams
def test_adamw_is_gated_on_device_flags_and_rejects_step_zero(small_conf):
    """amt_adamw_step_gated: a closed gate (int flag 0, or float flag < 1 after the mean all-reduce, or NaN) leaves
    parameters, m and v untouched (the skip of train.py:369-377 without a host sync); step < 1 is an error (the
    bias corrections 1 - b^step would be zero)."""
    model = amt.OutputSequenceGenerator(small_conf, key=5, dtype="float32", gamma=1.0)
    model.init_optimizer()
    g = torch.randn(model.n_params, device="cuda") * 1e-2
    p0 = model.weights_tensor().clone()
    zero_i = torch.zeros(1, dtype=torch.int32, device="cuda")
    one_i = torch.ones(1, dtype=torch.int32, device="cuda")
    for gate in (dict(valid=zero_i), dict(valid_mean=torch.tensor([0.875], device="cuda")),
                 dict(valid_mean=torch.tensor([float("nan")], device="cuda")),
                 dict(valid=one_i, valid_mean=torch.tensor([0.5], device="cuda"))):
        model.adamw_step(g, lr_base=1e-3, step=1, **gate)
        torch.cuda.synchronize()
        assert torch.equal(model.weights_tensor(), p0), gate
    # open gate: the first step of an untouched optimizer state must equal an ungated first step on a twin model
    twin = amt.OutputSequenceGenerator(small_conf, key=5, dtype="float32", gamma=1.0)
    twin.adamw_step(g, lr_base=1e-3, step=1)
    model.adamw_step(g, lr_base=1e-3, step=1, valid=one_i, valid_mean=torch.tensor([1.0], device="cuda"))
    torch.cuda.synchronize()
    assert not torch.equal(model.weights_tensor(), p0)
    # closed gates had left m and v untouched (the global-norm sum is an atomic reduction: last-bit differences only)
    assert float((model.weights_tensor() - twin.weights_tensor()).abs().max()) < 1e-6
    with pytest.raises(RuntimeError):
        model.adamw_step(g, lr_base=1e-3, step=0)


def test_device_seed_word_gives_fresh_masks_under_graph_replay(small_conf):
    """Dropout masks are drawn from seed + *amt_seed_word: a captured step whose first node advances the word draws
    fresh masks on every replay, and replay k equals the eager step with seed + k."""
    model = amt.OutputSequenceGenerator(small_conf, key=6, dtype="bfloat16", gamma=1.0)
    audio, labels = synth.make_batch(2, 320 * 64, seed=4)
    a, y = torch.from_numpy(audio).to("cuda", torch.bfloat16), torch.from_numpy(labels).cuda()
    gbuf = torch.zeros(model.n_params, dtype=torch.float32, device="cuda")
    word = model.seed_word()
    assert int(word.item()) == 0

    def eager(seed):
        word.zero_()
        out = model.loss_and_grads(a, y, seed=seed, train=True)
        return out.loss, out.grads_flat.clone()

    refs = [eager(100 + k) for k in range(1, 4)]
    assert abs(refs[0][0] - refs[1][0]) > 0 and float((refs[0][1] - refs[1][1]).norm()) > 0   # masks do differ
    word.zero_()

    def step():
        model.advance_seed(1)
        return model.loss_and_grads(a, y, seed=100, train=True, grads_out=gbuf, sync=False)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    word.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    for k in range(3):
        g.replay()
        torch.cuda.synchronize()
        assert int(word.item()) == k + 1
        assert abs(float(out[0][0]) - refs[k][0]) <= 1e-5 * abs(refs[k][0])
        assert float((gbuf - refs[k][1]).norm() / refs[k][1].norm()) < 1e-4


def test_replanning_between_known_shapes_keeps_a_captured_step_valid(small_conf):
    """an eager call at another shape followed by the replay of a graph captured earlier must not touch
    freed memory.  Shapes planned once keep their plan: alternating predict(B1, T1) and loss_and_grads(B2, T2)
    allocates nothing (amt_plan_generation constant) and the captured step keeps reproducing the eager result; a
    first-time LARGER shape does replace the workspace, and the generation counter says so."""
    model = amt.OutputSequenceGenerator(small_conf, key=8, dtype="bfloat16", gamma=1.0)
    audio, labels = synth.make_batch(3, 320 * 80, seed=7)
    a, y = torch.from_numpy(audio).to("cuda", torch.bfloat16), torch.from_numpy(labels).cuda()
    small_audio = torch.from_numpy(synth.make_batch(2, 320 * 48, seed=1)[0]).to("cuda", torch.bfloat16)
    gbuf = torch.zeros(model.n_params, dtype=torch.float32, device="cuda")

    def step():
        return model.loss_and_grads(a, y, seed=9, train=True, grads_out=gbuf, sync=False)

    model.predict(None, small_audio)           # plan (2, 48 frames, inference)
    step(); torch.cuda.synchronize()           # plan (3, 80 frames, training)
    ref = gbuf.clone()
    gen = model.plan_generation()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    logits_ref = model.predict(None, small_audio)[0].clone()
    for _ in range(3):
        lg, _ = model.predict(None, small_audio)             # eager, other shape, other mode: re-plans (cached)
        assert torch.equal(lg, logits_ref)
        gbuf.fill_(-5.0)
        g.replay()
        torch.cuda.synchronize()
        assert float((gbuf - ref).norm() / ref.norm()) < 1e-5
        assert model.plan_generation() == gen
    big = torch.from_numpy(synth.make_batch(6, 320 * 160, seed=2)[0]).to("cuda", torch.bfloat16)
    big_y = torch.from_numpy(synth.make_batch(6, 320 * 160, seed=2)[1]).cuda()
    model.loss_and_grads(big, big_y, seed=1, train=True)     # needs a larger workspace: the old one is replaced
    assert model.plan_generation() != gen                    # ... and graphs captured before are stale: re-capture


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_team_run_equals_one_launch_per_block(dtype, monkeypatch):
    """XCD-team form of the CNN stages (csrc/team.h, chain.hip: block_run_fwd, opt-in AMT_TEAM=1): the Blocks of a stage
    in ONE launch, hand-offs through the XCD's L2.  Same tile body: logits, loss and every gradient bit-identical to
    the one-launch-per-Block path; clips that do not fill the 8 teams, stages that straddle tiles and several clips
    per team included; no hand-off timed out."""
    conf = dict(np_model.MODEL_CONFIG)
    conf["depths"] = [1, 1, 2, 2, 3, 4, 2]
    conf["num_transformer_layers"] = 1
    outs = {}
    for B, S in ((3, 50), (11, 32)):
        audio, labels = synth.make_batch(B, 320 * S, seed=13)
        for team in ("0", "1"):
            monkeypatch.setenv("AMT_TEAM", team)
            model = amt.OutputSequenceGenerator(conf, key=21, dtype=dtype, gamma=1.0)
            out = model.loss_and_grads(torch.from_numpy(audio), torch.from_numpy(labels), seed=3, train=True,
                                       return_logits=True)
            assert out.grads_valid and model.team_status() == 0
            outs[team] = (out.loss, out.logits.clone(), out.grads_flat.clone())
        assert torch.equal(outs["0"][1], outs["1"][1])                                     # forward: bit for bit
        assert abs(outs["0"][0] - outs["1"][0]) <= 1e-6 * abs(outs["0"][0])                 # loss: an atomic sum
        assert float((outs["0"][2] - outs["1"][2]).norm() / outs["0"][2].norm()) < 1e-5   # backward: fp32 atomics order


@pytest.mark.gpu
def test_adamw_steps_match_the_optimizer_oracle(small_conf):
    """amt_adamw_step against oracle/optim.py (train.py:646-725): three consecutive updates with per-depth learning
    rates (layer_lr_decay 0.7), weight decay and the global-norm clip ACTIVE in the first two (so a norm accumulator
    that was not reset between steps, or a clip applied to the gradients instead of the updates, shows), then the
    device-side bucket words of amt_pack_tail."""
    from oracle import optim
    model = amt.OutputSequenceGenerator(small_conf, key=11, dtype="float32", gamma=1.0)
    model.init_optimizer(layer_lr_decay=0.7)
    segs = aparams.device_layout(small_conf)
    depth = np.concatenate([np.full(s.size, s.lr_depth) for s in segs])
    assert depth.size == model.n_params and (depth >= 0).any() and (depth < 0).any()
    chain = optim.AdamWChain(model.n_params, optim.lr_scale_from_depths(depth, 0.7), weight_decay=0.005, clip_norm=1.0)
    p_ref = model.weights_tensor().cpu().numpy().astype(np.float64)
    rng = np.random.default_rng(5)
    clipped = []
    for step, lr in ((1, 0.05), (2, 0.02), (3, 1e-4)):
        g = (rng.standard_normal(model.n_params) * 0.3).astype(np.float32)
        before = p_ref.copy()
        p_ref = chain.update(p_ref, g.astype(np.float64), lr=lr)
        clipped.append(abs(float(np.linalg.norm(p_ref - before)) - 1.0) < 1e-9)
        model.adamw_step(torch.from_numpy(g).cuda(), lr_base=lr, step=step)
        torch.cuda.synchronize()
        got = model.weights_tensor().cpu().numpy().astype(np.float64)
        err = float(np.abs(got - p_ref).max())
        assert err < 2e-6, (step, err)
        p_ref = got.copy()              # fp32 master weights: continue from what the device holds
    assert clipped == [True, True, False]
    # amt_pack_tail: loss and flag into the control words of a bucket, on the device
    loss = torch.tensor([3.25, 7.0], device="cuda")
    buf = torch.full((8,), -1.0, device="cuda")
    for flag, want in ((1, 1.0), (0, 0.0)):
        model.pack_tail(loss, torch.tensor([flag], dtype=torch.int32, device="cuda"), head_valid=buf[0:1],
                        tail_loss=buf[5:6], tail_valid=buf[6:7])
        torch.cuda.synchronize()
        assert buf.tolist() == [want, -1.0, -1.0, -1.0, -1.0, 3.25, want, -1.0]


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_large_n_chains_equal_the_short_tile_chains(small_conf, dtype, monkeypatch):
    """csrc/chain_big.hip (round 3): from 8192 tokens on the forward chains run 128-token tiles (64 in f32) with the hidden
    dimension in slices.  Every product keeps the unfused K order and every rounding its place, so inference logits,
    training logits, loss and - up to the atomic summation order of the weight gradients - gradients equal the short-tile
    chains' (AMT_BIG=0) bit for bit, ragged last tile included (41 clips x 250 frames = 10 250 tokens = 80 tiles + 10)."""
    conf = dict(small_conf)
    audio, labels = synth.make_batch(41, 80000, seed=12)
    a, y = torch.from_numpy(audio), torch.from_numpy(labels)
    outs = {}
    for big in ("1", "0"):
        monkeypatch.setenv("AMT_BIG", big)
        model = amt.OutputSequenceGenerator(conf, key=31, dtype=dtype, gamma=1.0)
        logits, _ = model.predict(None, a)
        step = model.loss_and_grads(a, y, seed=7, train=True, return_logits=True)
        outs[big] = (logits.clone(), step.logits.clone(), step.loss, step.grads_flat.clone())
        model.close()
    assert torch.equal(outs["1"][0], outs["0"][0])
    assert torch.equal(outs["1"][1], outs["0"][1])
    assert abs(outs["1"][2] - outs["0"][2]) <= 1e-6 * abs(outs["0"][2])
    rel = float((outs["1"][3] - outs["0"][3]).norm() / outs["0"][3].norm())
    assert rel < (1e-5 if dtype == "float32" else 1e-3), rel


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_large_n_weight_gradient_schedule_equals_the_rider_schedule(small_conf, dtype, monkeypatch):
    """From 8192 tokens on the backward pass runs another weight-gradient schedule (csrc/model.hip, round 4): no launch
    carries riders, a transformer scope's problems go out as one grouped launch (XCD-aware item order), and every CNN weight
    gradient waits for a late batch of 16-problem launches whose operands are kept until the end of the pass.  Same products,
    another summation order: loss and logits equal the rider schedule's bit for bit (AMT_RIDER_MAX_TOKENS=0: riders at every
    size), gradients to the atomics' order.  AMT_FLUSH_SCOPES=2 (two scopes per grouped launch) is held to the same."""
    audio, labels = synth.make_batch(41, 80000, seed=12)          # 10 250 tokens: past the switch
    a, y = torch.from_numpy(audio), torch.from_numpy(labels)
    outs = {}
    for name, env in (("riders", {"AMT_RIDER_MAX_TOKENS": "0"}), ("large-n", {}), ("two-scopes", {"AMT_FLUSH_SCOPES": "2"}),
                      ("late-off", {"AMT_CNN_WGRAD_LATE": "0"}), ("block-tiles-16", {"AMT_BLOCK_TRAIN_TT": "16"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        model = amt.OutputSequenceGenerator(small_conf, key=31, dtype=dtype, gamma=1.0)
        for k in env:
            monkeypatch.delenv(k)
        step = model.loss_and_grads(a, y, seed=7, train=True, return_logits=True)
        again = model.loss_and_grads(a, y, seed=7, train=True)              # the second call of a planned shape
        outs[name] = (step.logits.clone(), step.loss, step.grads_flat.clone(), again.grads_flat.clone())
        assert step.grads_valid and again.grads_valid
        model.close()
    ref = outs["riders"]
    # (block-tiles-16: the C = 128 training Block forward on its 16-row tiles instead of the 64-row tiles it takes from
    # 16 000 rows on - 41 clips x 500 rows here; rows are independent, so the logits must not move by a bit)
    for name in ("large-n", "two-scopes", "late-off", "block-tiles-16"):
        o = outs[name]
        assert torch.equal(o[0], ref[0]), name
        assert abs(o[1] - ref[1]) <= 1e-6 * abs(ref[1]), name
        for g in (o[2], o[3]):
            rel = float((g - ref[2]).norm() / ref[2].norm())
            assert rel < (1e-5 if dtype == "float32" else 1e-3), (name, rel)
            worst = float((g - ref[2]).abs().max() / ref[2].abs().max())
            assert worst < (1e-5 if dtype == "float32" else 2e-3), (name, worst)


def test_streaming_size_inference_forms_equal_the_short_tile_chains(small_conf, monkeypatch):
    """From 49 152 tokens on (streaming inference, BASELINE configs[3]) the bf16 forward runs `ffn_fwd_big` as two independent
    workgroups per compute unit on 64-token tiles and the C = 16 / 64 / 128 Blocks as two workgroups per compute unit on
    half-length tiles (csrc/chain_big.hip mode 5, csrc/chain.hip block_fwd_occ2_kernel).  Same products in the same order:
    the logits of 200 windows x 5 s (50 000 tokens, ragged last tiles) equal the short-tile chains' bit for bit."""
    audio, _ = synth.make_batch_fast(200, 80000, seed=21)
    a = torch.from_numpy(audio).to("cuda", torch.bfloat16)
    outs = {}
    for name, big, occ2 in (("new", None, None), ("old", "0", "0")):
        if big is None:
            monkeypatch.delenv("AMT_BIG", raising=False); monkeypatch.delenv("AMT_BLOCK_OCC2", raising=False)
        else:
            monkeypatch.setenv("AMT_BIG", big); monkeypatch.setenv("AMT_BLOCK_OCC2", occ2)
        model = amt.OutputSequenceGenerator(small_conf, key=5, dtype="bfloat16", gamma=1.0)
        logits, _ = model.predict(None, a)
        outs[name] = logits.clone()
        model.close()
    assert torch.isfinite(outs["new"]).all()
    assert torch.equal(outs["new"], outs["old"])


def test_streaming_size_training_forward_equals_the_four_wave_form(small_conf, monkeypatch):
    """The same occupancy-2 FFN form in its TRAINING instantiation (it saves h, hn, u, g for the backward pass): 99 clips x 10 s
    = 49 500 tokens through `loss_and_grads` with AMT_BIG default (5) against AMT_BIG=1 (four waves, one workgroup per compute
    unit): identical logits and loss, gradients equal up to the atomic summation order of the weight gradients."""
    audio, labels = synth.make_batch_fast(99, 160000, seed=23)
    a = torch.from_numpy(audio).to("cuda", torch.bfloat16)
    y = torch.from_numpy(labels).cuda()
    outs = {}
    for big in (None, "1"):
        if big is None:
            monkeypatch.delenv("AMT_BIG", raising=False)
        else:
            monkeypatch.setenv("AMT_BIG", big)
        model = amt.OutputSequenceGenerator(small_conf, key=9, dtype="bfloat16", gamma=1.0)
        o = model.loss_and_grads(a, y, seed=4, train=True, return_logits=True)
        outs[big] = (o.logits.clone(), o.loss, o.grads_flat.clone(), o.grads_valid)
        model.close()
    assert outs[None][3] and outs["1"][3]
    assert torch.equal(outs[None][0], outs["1"][0])
    assert abs(outs[None][1] - outs["1"][1]) <= 1e-6 * abs(outs["1"][1])
    rel = float((outs[None][2] - outs["1"][2]).norm() / outs["1"][2].norm())
    assert rel < 1e-3, rel


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_attention_backward_keep_bits_equal_the_rehashed_masks(small_conf, dtype, monkeypatch):
    """Round 3: the global-attention forward leaves the keep decisions of its weight dropout as one bit per score and
    both backward roles read them back (csrc/attn.hip, AttnArgs::keepbits) instead of hashing the 64 x 64 indices of every
    tile twice more.  Same decisions, so the step equals the re-hashing form (AMT_ATTN_BITS=0) up to the atomic summation
    order of the weight gradients; S = 150: three key tiles, the last one ragged."""
    audio, labels = synth.make_batch(3, 320 * 150, seed=17)
    a, y = torch.from_numpy(audio), torch.from_numpy(labels)
    outs = {}
    for bits in ("1", "0"):
        monkeypatch.setenv("AMT_ATTN_BITS", bits)
        model = amt.OutputSequenceGenerator(small_conf, key=13, dtype=dtype, gamma=1.0)
        o = model.loss_and_grads(a, y, seed=99, train=True, return_logits=True)
        outs[bits] = (o.logits.clone(), o.loss, o.grads_flat.clone())
        model.close()
    assert torch.equal(outs["1"][0], outs["0"][0])
    assert abs(outs["1"][1] - outs["0"][1]) <= 1e-6 * abs(outs["0"][1])          # the loss is an atomic sum: last bits
    rel = float((outs["1"][2] - outs["0"][2]).norm() / outs["0"][2].norm())
    assert rel < (1e-5 if dtype == "float32" else 1e-3), rel


def test_cfg5_share_fp8_attention_step_at_32_clips_per_gpu():
    """BASELINE configs[4] (global batch 256 over 8 GPUs = 32 clips x 10 s per GPU, fp8 MFMA attention forward, bf16
    everything else): the per-GPU share at its workload, through the HIP path.  No oracle finishes this size in a test, so
    the checks are the size-independent ones: the step is finite and valid, the fp8 step stays within fp8 rounding of the bf16
    step on the same batch (loss 1e-3 relative, gradients 2e-2 relative L2), clips stay independent (the first 8 clips' logits
    do not depend on the other 24), and the 16 000-token large-N chains are the ones that ran (AMT_BIG default)."""
    conf = np_model.MODEL_CONFIG
    audio, labels = synth.make_batch_fast(32, 160000, seed=5)
    a = torch.from_numpy(audio).to("cuda", torch.bfloat16)
    y = torch.from_numpy(labels).cuda()
    bf = amt.OutputSequenceGenerator(conf, key=1234, dtype="bfloat16", gamma=1.0)
    r = bf.loss_and_grads(a, y, seed=3, train=True, return_logits=True)
    assert r.grads_valid and np.isfinite(r.loss)
    small = bf.loss_and_grads(a[:8], y[:8], seed=3, train=False, return_logits=True)
    whole = bf.loss_and_grads(a, y, seed=3, train=False, return_logits=True)
    assert torch.equal(whole.logits[:8], small.logits)            # 16 000-token chains vs 4 000-token chains: same bits
    g_ref = r.grads_flat.clone(); l_ref = r.loss
    bf.close()
    f8 = amt.OutputSequenceGenerator(conf, key=1234, dtype="bfloat16", gamma=1.0)
    f8.set_attn_fp8(True)
    for it in range(2):                                             # first call: default scales; second: delayed scaling
        q = f8.loss_and_grads(a, y, seed=3, train=True)
        assert q.grads_valid and abs(q.loss - l_ref) < 1e-3 * abs(l_ref), (it, q.loss, l_ref)
        gd = float((q.grads_flat - g_ref).norm() / g_ref.norm())
        assert 0.0 < gd < 2e-2, (it, gd)
