"""Round 3: the rest of the training-step surface through the C ABI.

  * gradient accumulation over minibatches on the device   (train.py:275-310)  amt_forward_backward_accum
  * train-mode forward through the reference's call        (model.py:740-769)  amt_forward_train
  * optimizer count that skipped steps do not advance      (train.py:369-377)  amt_adamw_step_gated(step = -1)
  * a non-finite loss closes the bucket flags              (train.py:369)      amt_pack_tail
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import audio_to_midi_amd as amt  # noqa: E402
from audio_to_midi_amd import synth, train  # noqa: E402
from oracle import np_model, torch_model  # noqa: E402


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_accumulated_minibatches_equal_one_batch(small_conf, dtype):
    """k minibatches accumulated on the device == one batch of all their clips (mean of means of equal-sized groups), and
    == the float64 oracle on the whole batch; nothing reaches the caller's buffer before the last minibatch."""
    params = np_model.init_params(small_conf, seed=5, gamma=1.0)
    audio, labels = synth.make_batch(6, 320 * 40, seed=8)
    a, y = torch.from_numpy(audio), torch.from_numpy(labels)
    model = amt.OutputSequenceGenerator(small_conf, dtype=dtype)
    model.load_equinox_params(params)
    whole = model.loss_and_grads(a, y, scale=4.0, train=False)
    g = torch.full((model.n_params,), 9.0, dtype=torch.float32, device="cuda")
    loss = torch.full((2,), -1.0, device="cuda")
    valid = torch.zeros(1, dtype=torch.int32, device="cuda")
    k = 3
    for i in range(k):
        out = model.loss_and_grads(a[2 * i:2 * i + 2], y[2 * i:2 * i + 2], scale=4.0, train=False, grads_out=g, accum=(i, k),
                                   loss_out=loss, valid_out=valid)
        if i < k - 1:
            torch.cuda.synchronize()
            assert float((g - 9.0).abs().max()) == 0.0 and float(loss[0]) == -1.0      # untouched until the last one
    assert out.grads_valid
    tol = 1e-5 if dtype == "float32" else 2e-2      # bf16: d logits are scaled by 1 / minibatch size before they are rounded
    assert abs(out.loss - whole.loss) <= 1e-5 * abs(whole.loss) and abs(out.scaled_loss - 4 * whole.loss) <= 1e-4 * abs(whole.loss)
    assert float((out.grads_flat - whole.grads_flat).norm() / whole.grads_flat.norm()) < tol
    if dtype == "float32":
        loss_ref, _, grads_ref = torch_model.loss_and_grads(params, audio, labels, small_conf)
        got = model.grads_to_equinox(out.grads_flat.cpu().numpy())
        worst = max(np.linalg.norm(got[n] - r) / (np.linalg.norm(r) + 1e-30) for n, r in grads_ref.items()
                    if np.linalg.norm(r) > 1e-3 * max(np.linalg.norm(v) for v in grads_ref.values()))
        assert abs(out.loss - loss_ref) <= 1e-5 * abs(loss_ref) and worst < 2e-3, worst
    with pytest.raises(RuntimeError):
        model.loss_and_grads(a[:2], y[:2], accum=(3, 3))


def test_data_parallel_step_accumulates_on_the_device(small_conf):
    """DataParallelStep over a HipTrainBackend with accum_steps = 2 (eager and as a replayed hipGraph, one- and two-phase
    form): the bucket holds the gradients of the whole rank batch; every minibatch of a step draws its masks from the same
    key, as the reference's scan body does (train.py:274, 283-290)."""
    model = amt.OutputSequenceGenerator(small_conf, key=9, dtype="bfloat16", gamma=1.0)
    audio, labels = synth.make_batch(4, 320 * 48, seed=3)
    a, y = torch.from_numpy(audio).to("cuda", torch.bfloat16), torch.from_numpy(labels).cuda()
    ref = []
    for i in range(2):      # what the step must produce with seed s: every minibatch with seed s, summed / 2
        ref.append(model.loss_and_grads(a[2 * i:2 * i + 2], y[2 * i:2 * i + 2], seed=50, train=True))
    g_ref = (ref[0].grads_flat + ref[1].grads_flat) / 2
    l_ref = (ref[0].loss + ref[1].loss) / 2
    for overlap in (False, True):
        bucket = train.GradBucket(model.n_params, "cuda")
        be = train.HipTrainBackend(model, a, y, bucket, use_optimizer=False, accum_steps=2)
        dp = train.DataParallelStep(be, bucket, be.split, dist=None, overlap=overlap, graph=False)
        dp.prepare(seed=50)
        model.seed_word().zero_()
        dp.step(seed=50)
        loss, valid = dp.result()
        assert valid and abs(loss - l_ref) <= 1e-5 * abs(l_ref)
        assert float((bucket.grads - g_ref).norm() / g_ref.norm()) < 2e-2, overlap


def test_train_mode_forward_is_the_forward_of_the_training_step(small_conf):
    """OutputSequenceGenerator.__call__(..., enable_dropout=True) (model.py:740-769; compute_loss calls it at
    train.py:56-58): the logits are bit for bit the ones the training step computes its loss from for the same key,
    they differ from the inference logits, another key gives other masks, and inference ignores the key."""
    model = amt.OutputSequenceGenerator(small_conf, key=2, dtype="bfloat16", gamma=1.0)
    audio, labels = synth.make_batch(2, 320 * 64, seed=1)
    a, y = torch.from_numpy(audio), torch.from_numpy(labels)
    (lt, pt), state = model(a, None, None, key=77, enable_dropout=True)
    step = model.loss_and_grads(a, y, seed=77, train=True, return_logits=True)
    assert torch.equal(lt, step.logits) and state is None
    assert torch.allclose(pt, torch.sigmoid(lt), atol=1e-6)
    (li, _), _ = model(a, None, None, key=77, enable_dropout=False)
    assert not torch.equal(li, lt) and torch.equal(li, model.predict(None, a)[0])
    (lt2, _), _ = model(a, None, None, key=78, enable_dropout=True)
    assert not torch.equal(lt2, lt)
    z = lt.double()
    bce = torch.nn.functional.binary_cross_entropy_with_logits(z, y.cuda().double(), reduction="sum") / 2
    assert abs(float(bce) - step.loss) <= 1e-4 * abs(step.loss)


def test_optimizer_count_lives_on_the_device_and_skipped_steps_do_not_advance_it(small_conf):
    """with step = -1 the bias corrections use the device count of APPLIED updates.  open, skipped, open must
    equal two explicit steps 1, 2 on a twin model (a host count would have used 1, 3)."""
    model = amt.OutputSequenceGenerator(small_conf, key=5, dtype="float32", gamma=1.0)
    twin = amt.OutputSequenceGenerator(small_conf, key=5, dtype="float32", gamma=1.0)
    model.init_optimizer(); twin.init_optimizer()
    rng = np.random.default_rng(0)
    g1 = torch.from_numpy((rng.standard_normal(model.n_params) * 1e-2).astype(np.float32)).cuda()
    g2 = torch.from_numpy((rng.standard_normal(model.n_params) * 1e-2).astype(np.float32)).cuda()
    opened, closed = torch.tensor([1.0], device="cuda"), torch.tensor([0.5], device="cuda")
    assert model.optimizer_count() == 0
    model.adamw_step(g1, lr_base=1e-3, step=-1, valid_mean=opened)
    assert model.optimizer_count() == 1
    p1 = model.weights_tensor().clone()
    model.adamw_step(g2, lr_base=1e-3, step=-1, valid_mean=closed)          # non-finite somewhere: skipped
    assert model.optimizer_count() == 1 and torch.equal(model.weights_tensor(), p1)
    model.adamw_step(g2, lr_base=1e-3, step=-1, valid_mean=opened)
    assert model.optimizer_count() == 2
    twin.adamw_step(g1, lr_base=1e-3, step=1)
    twin.adamw_step(g2, lr_base=1e-3, step=2)
    assert twin.optimizer_count() == 2
    assert float((model.weights_tensor() - twin.weights_tensor()).abs().max()) < 5e-7      # norm^2 is an atomic sum: last bits
    with pytest.raises(RuntimeError):
        model.adamw_step(g1, lr_base=1e-3, step=-2)


def test_non_finite_loss_closes_the_bucket_flags(small_conf):
    model = amt.OutputSequenceGenerator(small_conf, key=5, dtype="float32", gamma=1.0)
    one = torch.ones(1, dtype=torch.int32, device="cuda")
    buf = torch.full((4,), -1.0, device="cuda")
    for lv, want in ((2.5, 1.0), (float("nan"), 0.0), (float("inf"), 0.0)):
        model.pack_tail(torch.tensor([lv, 0.0], device="cuda"), one, head_valid=buf[0:1], tail_loss=buf[1:2], tail_valid=buf[2:3])
        torch.cuda.synchronize()
        assert float(buf[0]) == want and float(buf[2]) == want


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_fused_optimizer_protocol_equals_the_three_pass_form(small_conf, dtype):
    """include/amt.h, "the fused single-rank protocol": loss_and_grads(fused_optimizer=True) leaves the raw gradient sums in
    the context and adamw_step(None) consumes them (norm pass + commit pass, accumulator left zeroed).  Three steps with the
    clip active equal grads_finalize + adamw_compute + adamw_apply on a twin model; a non-finite step (NaN audio) is
    skipped with flag 0, leaves parameters, m, v and the update count alone and does not leak into the next step; a
    discarded backward pass (reset_grad_accumulator) does not either; mixing in a plain loss_and_grads keeps the invariant."""
    audio, labels = synth.make_batch(2, 320 * 48, seed=21)
    a, y = torch.from_numpy(audio), torch.from_numpy(labels)
    fused = amt.OutputSequenceGenerator(small_conf, key=3, dtype=dtype, gamma=1.0)
    plain = amt.OutputSequenceGenerator(small_conf, key=3, dtype=dtype, gamma=1.0)
    fused.init_optimizer(); plain.init_optimizer()
    flag = torch.full((1,), -7, dtype=torch.int32, device="cuda")
    lr = 2e-2          # large enough for the global-norm clip (1.0) to bind in the first steps
    for k in range(3):
        if k == 1:      # a discarded pass and a plain call in between must not change what the next fused step applies
            fused.loss_and_grads(a, y, seed=500, train=True, fused_optimizer=True)
            fused.reset_grad_accumulator()
            fused.loss_and_grads(a, y, seed=501, train=True)
        fused.loss_and_grads(a, y, scale=2.0, seed=10 + k, train=True, fused_optimizer=True)
        fused.adamw_step(None, lr_base=lr, step=-1, valid=flag)
        out = plain.loss_and_grads(a, y, scale=2.0, seed=10 + k, train=True)
        plain.adamw_step(out.grads_flat, lr_base=lr, step=k + 1)
        torch.cuda.synchronize()
        assert int(flag.item()) == 1 and fused.optimizer_count() == k + 1
        d = float((fused.weights_tensor() - plain.weights_tensor()).abs().max())
        # float32 pins the arithmetic (the only difference is the summation order of the weight-gradient atomics).  bfloat16:
        # a last-bit difference of an fp32 master weight can flip the bf16 rounding of its compute copy (2^-9 relative), the
        # next step's gradients then differ visibly and Adam's early steps (|update| ~ lr whatever |g|) turn that into a few
        # percent of one step: measured 4.7e-4 at lr = 2e-2
        assert d < (2e-6 if dtype == "float32" else 2e-3), (k, d)
    p_before = fused.weights_tensor().clone()
    bad = a.clone(); bad[0, 0, 100] = float("nan")
    fused.loss_and_grads(bad, y, seed=77, train=True, fused_optimizer=True)
    fused.adamw_step(None, lr_base=lr, step=-1, valid=flag)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0 and fused.optimizer_count() == 3 and torch.equal(fused.weights_tensor(), p_before)
    fused.loss_and_grads(a, y, scale=2.0, seed=13, train=True, fused_optimizer=True)       # the poisoned sums are gone
    fused.adamw_step(None, lr_base=lr, step=-1, valid=flag)
    out = plain.loss_and_grads(a, y, scale=2.0, seed=13, train=True)
    plain.adamw_step(out.grads_flat, lr_base=lr, step=4)
    torch.cuda.synchronize()
    assert int(flag.item()) == 1 and bool(torch.isfinite(fused.weights_tensor()).all())
    assert float((fused.weights_tensor() - plain.weights_tensor()).abs().max()) < (2e-6 if dtype == "float32" else 4e-3)
    fresh = amt.OutputSequenceGenerator(small_conf, key=3, dtype=dtype, gamma=1.0)
    with pytest.raises(RuntimeError):
        fresh.adamw_step(None, lr_base=lr, step=-1)            # nothing has been left in the context


def test_one_launch_optimizer_times_out_safely_and_falls_back(small_conf, monkeypatch):
    """head.hip: adamw_persist_kernel runs its two passes around a grid barrier that only opens when every workgroup is
    resident.  With more workgroups than the device can hold (AMT_OPT_PERSIST_EXTRA) the barrier cannot open: the launch must
    end by its spin limit (not hang) and STILL APPLY the step - the resident workgroups leave untouched, the rest run, and the
    arrival that completes the count commits the whole update alone, with the bits of the two-launch form (round 4; round 3
    skipped such a step, which would let one rank of a data-parallel job part from the others).  The time-out is counted and
    the next call takes the two launches."""
    monkeypatch.setenv("AMT_OPT_PERSIST_EXTRA", "64")
    audio, labels = synth.make_batch(2, 320 * 48, seed=21)
    a, y = torch.from_numpy(audio), torch.from_numpy(labels)
    fused = amt.OutputSequenceGenerator(small_conf, key=3, dtype="float32", gamma=1.0)
    monkeypatch.delenv("AMT_OPT_PERSIST_EXTRA")
    plain = amt.OutputSequenceGenerator(small_conf, key=3, dtype="float32", gamma=1.0)
    fused.init_optimizer(); plain.init_optimizer()
    flag = torch.full((1,), -7, dtype=torch.int32, device="cuda")
    for step in (1, 2, 3):          # step 1: time-out + lone commit; step 2 on: two launches
        fused.loss_and_grads(a, y, scale=2.0, seed=10 + step, train=True, fused_optimizer=True)
        fused.adamw_step(None, lr_base=2e-2, step=-1, valid=flag)
        out = plain.loss_and_grads(a, y, scale=2.0, seed=10 + step, train=True)
        plain.adamw_step(out.grads_flat, lr_base=2e-2, step=step)
        torch.cuda.synchronize()
        assert int(flag.item()) == 1 and fused.optimizer_count() == step      # (also where the host notices the sticky word)
        assert fused.optimizer_status() == 1                                   # one time-out, counted once
        assert float((fused.weights_tensor() - plain.weights_tensor()).abs().max()) < 2e-6 * step, step
    assert float(train.GradBucket.of_model(fused).grads.abs().max()) == 0.0   # every commit left a clean accumulator


@pytest.mark.parametrize("late", ["inside", "outside"])
def test_one_launch_optimizer_verdict_is_unanimous_with_a_late_workgroup(small_conf, monkeypatch, late):
    """ADVICE r3 / VERDICT r3 weak 7: one RESIDENT workgroup of the one-launch optimizer is held back in front of its arrival
    (AMT_OPT_PERSIST_DELAY=<wg>:<polls>) while the others spin with a short limit (AMT_OPT_PERSIST_SPINS).
      inside  - it arrives before the limit: the barrier opens, the update is applied in full and equals the three-pass form;
      outside - the others' limit expires first: the launch's decision word says "failed", and the late workgroup - the arrival
                that completes the count - must take THAT verdict, not commit its own slice from its registers (round 3's
                per-workgroup verdict did: a partially applied step): it commits the WHOLE update alone.  The step equals the
                three-pass form's, amt_optimizer_status counts one time-out, and the next call takes the two launches."""
    monkeypatch.setenv("AMT_OPT_PERSIST_SPINS", "4000")
    monkeypatch.setenv("AMT_OPT_PERSIST_DELAY", "7:%d" % (400 if late == "inside" else 40000))
    audio, labels = synth.make_batch(2, 320 * 48, seed=21)
    a, y = torch.from_numpy(audio), torch.from_numpy(labels)
    fused = amt.OutputSequenceGenerator(small_conf, key=3, dtype="float32", gamma=1.0)
    monkeypatch.delenv("AMT_OPT_PERSIST_SPINS"); monkeypatch.delenv("AMT_OPT_PERSIST_DELAY")
    plain = amt.OutputSequenceGenerator(small_conf, key=3, dtype="float32", gamma=1.0)
    fused.init_optimizer(); plain.init_optimizer()
    flag = torch.full((1,), -7, dtype=torch.int32, device="cuda")
    fused.loss_and_grads(a, y, scale=2.0, seed=10, train=True, fused_optimizer=True)
    fused.adamw_step(None, lr_base=2e-2, step=-1, valid=flag)
    out = plain.loss_and_grads(a, y, scale=2.0, seed=10, train=True)
    plain.adamw_step(out.grads_flat, lr_base=2e-2, step=1)
    torch.cuda.synchronize()
    assert int(flag.item()) == 1 and fused.optimizer_count() == 1
    assert float((fused.weights_tensor() - plain.weights_tensor()).abs().max()) < 2e-6
    assert fused.optimizer_status() == (0 if late == "inside" else 1)
    assert fused.optimizer_status() == (0 if late == "inside" else 1)                  # counted once
    # the next step (two launches after a time-out) starts from whole m / v and a clean accumulator
    fused.loss_and_grads(a, y, scale=2.0, seed=11, train=True, fused_optimizer=True)
    fused.adamw_step(None, lr_base=2e-2, step=-1, valid=flag)
    out = plain.loss_and_grads(a, y, scale=2.0, seed=11, train=True)
    plain.adamw_step(out.grads_flat, lr_base=2e-2, step=2)
    torch.cuda.synchronize()
    assert int(flag.item()) == 1 and fused.optimizer_count() == 2
    assert float((fused.weights_tensor() - plain.weights_tensor()).abs().max()) < 4e-6


@pytest.mark.parametrize("clip_active", [True, False])
def test_optimizer_forms_agree_bit_for_bit_on_the_same_gradients(small_conf, monkeypatch, clip_active):
    """The global norm of the updates is summed in a fixed order (head.hip: sum_partials; per-workgroup slots instead of a
    float atomicAdd in arrival order), and the two-launch form partitions the parameters as the one-launch form does: from
    the SAME accumulator contents the one-launch form, the two-launch form and a one-launch step whose barrier timed out (one
    workgroup commits) write the same bits - which is what keeps the replicas of a data-parallel job identical when the clip
    is active, also when one rank has fallen back to two launches (tests/test_dp_ranks_gpu.py has the ranks)."""
    audio, labels = synth.make_batch(2, 320 * 48, seed=21)
    a, y = torch.from_numpy(audio), torch.from_numpy(labels)
    gen = torch.Generator(device="cpu").manual_seed(5)
    models = {}
    for form in ("one-launch", "two-launch", "timed-out"):
        if form == "two-launch":
            monkeypatch.setenv("AMT_OPT_PERSIST", "0")
        if form == "timed-out":
            monkeypatch.setenv("AMT_OPT_PERSIST_SPINS", "4000")
            monkeypatch.setenv("AMT_OPT_PERSIST_DELAY", "7:40000")
        m = amt.OutputSequenceGenerator(small_conf, key=3, dtype="float32", gamma=1.0)
        for k in ("AMT_OPT_PERSIST", "AMT_OPT_PERSIST_SPINS", "AMT_OPT_PERSIST_DELAY"):
            monkeypatch.delenv(k, raising=False)
        m.init_optimizer()
        models[form] = (m, train.GradBucket.of_model(m))
    n = models["one-launch"][0].n_params
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    for step in (1, 2, 3):
        # the same gradient sums in every context (a backward pass of its own would differ in its last bits: fp32 atomics)
        g = (torch.randn(n, generator=gen) * (1.0 if clip_active else 1e-3)).cuda()
        for form, (m, bucket) in models.items():
            m.loss_and_grads(a, y, seed=step, train=True, fused_optimizer=True)        # marks the accumulator as holding a step
            bucket.grads.copy_(g)
            m.adamw_step(None, lr_base=(1e-2 if clip_active else 1e-6), step=-1, valid=flag)
            torch.cuda.synchronize()
            assert int(flag.item()) == 1 and m.optimizer_count() == step, form
        w = {form: m.weights_tensor() for form, (m, _) in models.items()}
        assert torch.equal(w["one-launch"], w["two-launch"]), step
        assert torch.equal(w["one-launch"], w["timed-out"]), step
    assert models["one-launch"][0].optimizer_status() == 0 and models["timed-out"][0].optimizer_status() == 1


def test_in_place_protocol_on_the_models_own_bucket(small_conf):
    """VERDICT r3 item 2 on one GPU: DataParallelStep over HipTrainBackend(fused=True) and GradBucket.of_model - the bucket IS
    the context's accumulator (include/amt.h: amt_grad_bucket), the step's collectives would run on it in place, and
    amt_adamw_step_reduced consumes it.  One-collective and two-phase (overlap) form, graph replay, against the round-3
    protocol (separate bucket, unscale pass, three-pass optimizer) on a twin model; accumulation over two minibatches."""
    audio, labels = synth.make_batch(4, 320 * 48, seed=4)
    outs = {}
    for mode in ("legacy", "inplace", "inplace-overlap", "inplace-signal-overlap", "inplace-signal-accum2-overlap", "inplace-accum2"):
        model = amt.OutputSequenceGenerator(small_conf, key=8, dtype="float32", gamma=1.0)
        a, y = torch.from_numpy(audio).cuda(), torch.from_numpy(labels).cuda()
        inplace = mode != "legacy"
        bucket = train.GradBucket.of_model(model) if inplace else train.GradBucket(model.n_params, "cuda")
        be = train.HipTrainBackend(model, a, y, bucket, fused=inplace, accum_steps=2 if "accum2" in mode else 1)
        assert be.in_place == inplace
        # "signal": ONE launch sequence + the device-side first-half signal (amt_first_half_wait); else the two-call form
        dp = train.DataParallelStep(be, bucket, be.split, dist=None, overlap=mode.endswith("overlap"), graph=True, lr_base=1e-3,
                                    signal=("signal" in mode) if mode.endswith("overlap") else None)
        assert dp.in_place == inplace and dp.signal == ("signal" in mode)
        dp.prepare(seed=5)
        assert float(bucket.grads.abs().max()) == 0.0 or not inplace      # rehearsal sums were discarded
        model.seed_word().zero_()
        for _ in range(3):
            dp.step()
        loss, valid = dp.result()
        assert valid and dp.applied == 3 and dp.optimizer_timeouts == 0 and model.first_half_status() == 0
        if inplace:
            assert float(bucket.grads.abs().max()) == 0.0                  # the commit pass leaves the accumulator zeroed
        outs[mode] = (loss, model.weights_tensor().clone())
    for mode in ("inplace", "inplace-overlap", "inplace-signal-overlap"):
        assert abs(outs[mode][0] - outs["legacy"][0]) <= 1e-5 * abs(outs["legacy"][0]), mode
        assert float((outs[mode][1] - outs["legacy"][1]).abs().max()) < 5e-6, mode
    # two minibatches of two clips with the same key: each clip sees the masks of its position in the MINIBATCH, so the
    # step is a different (equally valid) draw - finite, applied, and close to the one-batch step
    assert float((outs["inplace-accum2"][1] - outs["legacy"][1]).abs().max()) < 5e-2
    assert float((outs["inplace-signal-accum2-overlap"][1] - outs["inplace-accum2"][1]).abs().max()) < 5e-6
    with pytest.raises(ValueError):       # more than one rank on a foreign bucket cannot use the fused optimizer
        class _D:
            ReduceOp = None
            @staticmethod
            def is_initialized(): return True
            @staticmethod
            def get_world_size(): return 2
            @staticmethod
            def get_backend(): return "gloo"
        m2 = amt.OutputSequenceGenerator(small_conf, key=8, dtype="float32", gamma=1.0)
        b2 = train.GradBucket(m2.n_params, "cuda")
        train.DataParallelStep(train.HipTrainBackend(m2, a, y, b2, fused=True), b2, 10, dist=_D())


def test_data_parallel_step_uses_the_fused_protocol_under_graph_replay(small_conf):
    """DataParallelStep over HipTrainBackend(fused=True), captured and replayed: the rehearsal runs of prepare() are
    discarded, every replay + apply is one optimizer update, and the result equals the unfused backend's."""
    audio, labels = synth.make_batch(2, 320 * 48, seed=4)
    outs = {}
    for fused in (True, False):
        # float32: the two forms then differ only by the summation order of the weight-gradient atomics (a bf16 context
        # amplifies such last-bit differences through the rounding of its compute copies: 3e-4 after three steps)
        model = amt.OutputSequenceGenerator(small_conf, key=8, dtype="float32", gamma=1.0)
        a, y = torch.from_numpy(audio).cuda(), torch.from_numpy(labels).cuda()
        bucket = train.GradBucket(model.n_params, "cuda")
        be = train.HipTrainBackend(model, a, y, bucket, fused=fused)
        dp = train.DataParallelStep(be, bucket, be.split, dist=None, overlap=False, graph=True, lr_base=1e-3)
        dp.prepare(seed=5)
        model.seed_word().zero_()
        for _ in range(3):
            dp.step()
        loss, valid = dp.result()
        assert valid and dp.applied == 3 and model.optimizer_count() == 3
        outs[fused] = (loss, model.weights_tensor().clone())
    assert abs(outs[True][0] - outs[False][0]) <= 1e-5 * abs(outs[False][0])
    assert float((outs[True][1] - outs[False][1]).abs().max()) < 5e-6


@pytest.mark.parametrize("form", ["two input sets", "staging copies"])
def test_input_feeder_hands_every_step_its_own_batch(small_conf, form):
    """train.InputFeeder (train.py:280: a fresh batch per step) in both forms - uploads straight into the input set the next
    captured step reads (the step is captured once per set), and staging pairs copied on the device in front of the step.
    With a zero learning rate the parameters stay put, so the loss of step k is a function of batch k alone: six steps over
    three distinct batches must report, in order, the losses of those batches evaluated one by one."""
    conf = dict(small_conf); conf["transformer_dropout_rate"] = 0.0
    model = amt.OutputSequenceGenerator(conf, key=8, dtype="float32", gamma=1.0)
    batches = [synth.make_batch(2, 320 * 48, seed=50 + j) for j in range(3)]
    want = []
    for au, ev in batches:
        out = model.loss_and_grads(torch.from_numpy(au), torch.from_numpy(ev), seed=1, train=True)
        want.append(float(out.loss))
    assert len({round(w, 3) for w in want}) == 3
    a0, y0 = torch.from_numpy(batches[0][0]).cuda(), torch.from_numpy(batches[0][1]).cuda()
    bucket = train.GradBucket.of_model(model)
    be = train.HipTrainBackend(model, a0, y0, bucket, fused=True)
    if form == "two input sets":
        be.add_input_set()
    dp = train.DataParallelStep(be, bucket, be.split, dist=None, graph=True, lr_base=0.0)
    dp.prepare(seed=5)
    feeder = train.InputFeeder(backend=be) if form == "two input sets" else train.InputFeeder(be.audio, be.labels)
    pinned = [train.InputFeeder.pin(au, ev, model.audio_dtype) for au, ev in batches]
    feeder.prefetch(*pinned[0])
    got = []
    for k in range(6):
        feeder.take()
        dp.step()
        if k + 1 < 6:
            feeder.prefetch(*pinned[(k + 1) % 3])
        loss, valid = dp.result()
        assert valid
        got.append(loss)
    for k in range(6):
        assert abs(got[k] - want[k % 3]) <= 2e-5 * abs(want[k % 3]), (form, k, got, want)
    if form == "two input sets":
        assert len(dp._graph_sets) == 2 and be.active in (0, 1)


def test_train_entry_loss_falls_and_a_skipped_step_rolls_nothing_forward(small_conf, tmp_path):
    """audio_to_midi_amd.train.train(): the reference's loop (train.py:211-452) wired from the schedule, the loss-scale
    controller, DataParallelStep on the model's own bucket, the double-buffered input feeder and the checkpoint cadence.
    50 steps on a small model over a fixed set of four batches: the loss falls; one batch carries a NaN sample - that step is
    skipped ON THE DEVICE (nothing to roll back: parameters stay finite, the update count is steps - skipped) and the loop
    goes on; checkpoints appear every 20 steps and only the newest `checkpoints_to_keep` stay."""
    model = amt.OutputSequenceGenerator(small_conf, key=11, dtype="bfloat16", gamma=1.0)
    batches = []
    for j in range(4):
        audio, labels = synth.make_batch(4, 320 * 48, seed=30 + j)
        batches.append((labels, audio))                     # the reference's order: (events, audio)

    def loader():
        for k in range(50):
            ev, au = batches[k % 4]
            if k == 17:
                au = au.copy(); au[1, 0, 1000] = np.nan     # one poisoned sample
            yield ev, au

    logs = []
    out = train.train(model, loader(), num_steps=50, base_learning_rate=2e-3, warmup_steps=5, checkpoint_dir=tmp_path,
                      checkpoint_every=20, checkpoints_to_keep=1, print_every=10, check_every=1, log=logs.append)
    assert out["steps"] == 50 and out["skipped"] == 1 and out["applied"] == 49 == model.optimizer_count()
    assert bool(torch.isfinite(model.weights_tensor()).all())
    hist = dict(out["history"])
    assert 18 not in hist and len(hist) == 49                                  # step 18 consumed the poisoned batch
    first, last = np.mean([hist[k] for k in range(1, 9)]), np.mean([hist[k] for k in range(43, 51)])
    assert last < 0.8 * first, (first, last)
    assert out["checkpoints"] == [40] and sorted(p.name for p in tmp_path.iterdir()) == ["metadata_40.json", "opt_40.f32", "params_40.f32"]
    assert any("NAN/inf" in ln for ln in logs) and any(ln.startswith("Step 50/50") for ln in logs)
    from audio_to_midi_amd import infer
    flat, conf, step = infer.load_checkpoint(tmp_path)
    assert step == 40 and flat.size == model.n_params


def test_train_entry_recaptures_on_a_new_loss_scale_and_keeps_its_batches_in_order(small_conf):
    """train(loss_scaling=True): the scale doubles while scaled_loss < 1e4 (train.py:379-382) and is halved by a non-finite
    step (train.py:369-377); the scale is an argument of the captured launches, so every change re-captures the step - on the
    SAME two input sets, with the set in use carried over (InputFeeder keeps feeding).  The poisoned batch must still be the
    one step 8 consumes, after three re-captures."""
    model = amt.OutputSequenceGenerator(small_conf, key=11, dtype="bfloat16", gamma=1.0)
    batches = [synth.make_batch(4, 320 * 48, seed=70 + j) for j in range(3)]

    def loader():
        for k in range(14):
            au, ev = batches[k % 3]
            if k == 7:
                au = au.copy(); au[2, 1, 500] = np.inf
            yield ev, au

    logs = []
    out = train.train(model, loader(), num_steps=14, base_learning_rate=1e-3, warmup_steps=2, loss_scaling=True, grad_scale=1.0,
                      print_every=0, log=logs.append)
    hist = dict(out["history"])
    assert out["steps"] == 14 and out["skipped"] == 1 and 8 not in hist and len(hist) == 13, (out["skipped"], sorted(hist))
    assert out["applied"] == 13 == model.optimizer_count()
    assert sum("Grad scale" in ln for ln in logs) >= 2 and any("NAN/inf" in ln for ln in logs)
    assert out["grad_scale"] >= 2.0
    assert bool(torch.isfinite(model.weights_tensor()).all())
    # the losses are unscaled whatever the scale was, and they fall over the same three batches
    assert np.mean([hist[k] for k in (12, 13, 14)]) < np.mean([hist[k] for k in (1, 2, 3)])


def test_captured_optimizer_takes_the_two_launch_form(small_conf, monkeypatch):
    """a one-launch optimizer captured into a hipGraph would replay with no host in between to notice a barrier
    time-out.  Under stream capture the fused protocol therefore takes the norm + commit launches: with a context whose
    one-launch form WOULD time out (64 workgroups more than can be resident, a short spin limit), a captured step replays
    three times, applies three updates and reports no time-out - while the same context's EAGER call does time out (and
    says so through amt_optimizer_status; the step is applied all the same)."""
    audio, labels = synth.make_batch(2, 320 * 48, seed=21)
    a, y = torch.from_numpy(audio).cuda().to(torch.float16), torch.from_numpy(labels).cuda()
    warm = amt.OutputSequenceGenerator(small_conf, key=3, dtype="bfloat16", gamma=1.0)      # loads every kernel once, eagerly
    warm.init_optimizer()
    warm.loss_and_grads(a, y, seed=1, train=True, fused_optimizer=True)
    warm.adamw_step(None, lr_base=1e-3, step=-1)
    monkeypatch.setenv("AMT_OPT_PERSIST", "0")
    warm2 = amt.OutputSequenceGenerator(small_conf, key=3, dtype="bfloat16", gamma=1.0)     # ... the two-launch kernels too
    warm2.init_optimizer()
    warm2.loss_and_grads(a, y, seed=1, train=True, fused_optimizer=True)
    warm2.adamw_step(None, lr_base=1e-3, step=-1)
    monkeypatch.delenv("AMT_OPT_PERSIST")
    monkeypatch.setenv("AMT_OPT_PERSIST_EXTRA", "64")
    monkeypatch.setenv("AMT_OPT_PERSIST_SPINS", "2000")
    model = amt.OutputSequenceGenerator(small_conf, key=3, dtype="bfloat16", gamma=1.0)
    monkeypatch.delenv("AMT_OPT_PERSIST_EXTRA"); monkeypatch.delenv("AMT_OPT_PERSIST_SPINS")
    model.init_optimizer()
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    model.loss_and_grads(a, y, seed=1, train=True, fused_optimizer=True)        # plans the workspace outside the capture
    model.reset_grad_accumulator()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        model.advance_seed(1)
        model.loss_and_grads(a, y, seed=5, train=True, fused_optimizer=True, sync=False)
        model.adamw_step(None, lr_base=1e-3, step=-1, valid=flag)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert model.optimizer_count() == 3 and int(flag.item()) == 1 and model.optimizer_status() == 0
    # the same context, eagerly: the one-launch form cannot get its 64 extra workgroups resident - it times out, says so,
    # and the step is applied all the same (one workgroup commits it)
    p = model.weights_tensor().clone()
    model.loss_and_grads(a, y, seed=9, train=True, fused_optimizer=True)
    model.adamw_step(None, lr_base=1e-3, step=-1, valid=flag)
    torch.cuda.synchronize()
    assert int(flag.item()) == 1 and not torch.equal(model.weights_tensor(), p)
    assert model.optimizer_status() == 1 and model.optimizer_count() == 4


@pytest.mark.gpu
def test_float16_labels_go_in_as_they_are(small_conf):
    """The reference's batches carry float16 labels (`midi_events.astype(np.float16)`, /root/reference/grain_loader.py:88):
    amt_set_label_dtype(AMT_F16) makes amt_forward_backward* read them without a conversion.  On labels that are float16
    values the two forms are the same computation: same loss, same gradients (up to the weight-gradient atomics)."""
    import numpy as np
    import torch
    import audio_to_midi_amd as amt
    from audio_to_midi_amd import synth
    model = amt.OutputSequenceGenerator(small_conf, key=5, dtype="bfloat16", gamma=1.0)
    audio, labels = synth.make_batch(2, 320 * 100, seed=8)
    y16 = torch.from_numpy(labels).to(torch.float16)
    a = model.loss_and_grads(torch.from_numpy(audio), y16, seed=3, train=True)
    b = model.loss_and_grads(torch.from_numpy(audio), y16.to(torch.float32), seed=3, train=True)
    assert abs(a.loss - b.loss) <= 1e-6 * abs(b.loss)
    assert float((a.grads_flat - b.grads_flat).norm() / b.grads_flat.norm()) < 1e-5
    # and the float16 rounding of the labels itself moves the loss by less than 1e-3 relative
    c = model.loss_and_grads(torch.from_numpy(audio), torch.from_numpy(labels), seed=3, train=True)
    assert abs(a.loss - c.loss) <= 1e-3 * abs(c.loss)


@pytest.mark.gpu
def test_input_feeder_uploads_one_block_per_batch(small_conf):
    """HipTrainBackend keeps audio and labels of an input set in ONE device block and InputFeeder.pin builds the matching
    pinned block, so a fresh batch is one host-to-device copy (train.py:280 uploads every minibatch)."""
    import torch
    import audio_to_midi_amd as amt
    from audio_to_midi_amd import synth, train as T
    model = amt.OutputSequenceGenerator(small_conf, key=5, dtype="bfloat16", gamma=1.0)
    audio, labels = synth.make_batch(2, 320 * 100, seed=8)
    bucket = T.GradBucket.of_model(model)
    be = T.HipTrainBackend(model, torch.from_numpy(audio).cuda(), torch.from_numpy(labels).to("cuda", torch.float16), bucket, fused=True)
    feeder = T.InputFeeder(backend=be)
    assert isinstance(be._sets[0], T.InputBlock) and len(be._sets) == 2
    a2, y2 = synth.make_batch(2, 320 * 100, seed=9)
    ha, hy = T.InputFeeder.pin(a2, y2, model.audio_dtype, torch.float16)
    assert ha._amt_block is hy._amt_block and ha.is_pinned()
    feeder.prefetch(ha, hy)
    feeder.take()
    torch.cuda.synchronize()
    assert torch.equal(be.audio.cpu(), ha) and torch.equal(be.labels.cpu(), hy)
    assert be.labels.dtype == torch.float16 and be.audio.data_ptr() == be._sets[be.active].buf.data_ptr()


@pytest.mark.gpu
def test_train_resumes_where_a_checkpoint_left_off(small_conf, tmp_path):
    """train(resume=True) restores parameters, AdamW moments and the count of applied updates from the newest checkpoint and
    continues at its step + 1 (the reference: checkpoint manager + `start_step`, /root/reference/train.py:211, 799-831): 6 steps
    in one go and 4 + 2 resumed steps on the same batches end with the same parameters (dropout off: p = 0 in this conf)."""
    import torch
    import audio_to_midi_amd as amt
    from audio_to_midi_amd import synth, train as T
    conf = dict(small_conf)
    conf["transformer_dropout_rate"] = 0.0
    batches = [synth.make_batch(2, 320 * 60, seed=40 + i) for i in range(6)]

    def loader(lo, hi):
        for a, y in batches[lo:hi]:
            yield y, a

    kw = dict(base_learning_rate=1e-3, warmup_steps=2, checkpoint_every=2, print_every=0, log=lambda *_: None)
    m1 = amt.OutputSequenceGenerator(conf, key=11, dtype="float32", gamma=1.0)
    r1 = T.train(m1, loader(0, 6), num_steps=6, checkpoint_dir=tmp_path / "a", **kw)
    w1 = m1.weights_tensor().clone()
    m2 = amt.OutputSequenceGenerator(conf, key=11, dtype="float32", gamma=1.0)
    T.train(m2, loader(0, 4), num_steps=6, checkpoint_dir=tmp_path / "b", **kw)      # the loader ends after 4 batches: "interrupted" at step 4
    m3 = amt.OutputSequenceGenerator(conf, key=99, dtype="float32", gamma=1.0)      # other weights: everything comes from the checkpoint
    r3 = T.train(m3, loader(4, 6), num_steps=6, checkpoint_dir=tmp_path / "b", resume=True, **kw)
    assert r3["start_step"] == 4 and r3["applied"] == 6 == r1["applied"]
    rel = float((m3.weights_tensor() - w1).norm() / w1.norm())
    assert rel < 1e-5, rel
