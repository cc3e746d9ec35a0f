"""Multi-process (world_size 2 and 4, gloo, CPU tensors) test of the data-parallel step protocol:
N ranks on shards of the global batch == 1 rank on the whole batch, the finite
flag is an AND over ranks, and the loss-scale controller follows train.py:369-382."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from audio_to_midi_amd import params as P
from audio_to_midi_amd import synth, train
from oracle import np_model, torch_model


def _conf():
    conf = dict(np_model.MODEL_CONFIG)
    conf["depths"] = [1, 1, 1, 1, 1, 1, 1]
    conf["num_transformer_layers"] = 1
    return conf


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_loss_scale_controller():
    c = train.LossScaleController(scale=8.0)
    assert c.update(True, 100.0, 800.0) is False and c.scale == 16.0         # scaled_loss < 1e4: double
    assert c.update(True, 100.0, 20000.0) is False and c.scale == 16.0
    assert c.update(False, 100.0, 800.0) is True and c.scale == 8.0          # non-finite grads: halve + skip
    assert c.update(True, float("nan"), 1.0) is True and c.scale == 4.0
    off = train.LossScaleController(enabled=False)
    assert off.update(True, 1.0, 1.0) is False and off.scale == 1.0


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` typed without a launcher: the process starts the ranks as children
    through torch.distributed.run BEFORE anything touches a GPU, relays rank 0's JSON line and exits with their code.
    --dry-run-launcher stops after the rendezvous and one all-reduce on CPU tensors (gloo), so this runs without a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run-launcher"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["ranks_seen"] == 2 and out["n_gpus"] == 2 and out["allreduce_ok"] is True and out["backend"] == "gloo"


# ------------------------------------------------------------------ DataParallelStep (the protocol bench.py runs)
class _OracleBackend:
    """CPU stand-in for HipTrainBackend: the oracle computes the rank's gradients, the bucket / flag / collective /
    gate logic around it is the production code (audio_to_midi_amd.train.DataParallelStep)."""

    def __init__(self, conf, params, audio, labels, bucket, split, poison=False, accum_steps=1):
        self.conf, self.params, self.audio, self.labels = conf, params, audio, labels
        self.bucket, self.split, self.poison = bucket, split, poison
        self.accum_steps = accum_steps
        self.gates = []
        self._cache = None

    def _compute(self):
        # train.py:275-310: the rank's batch in `accum_steps` minibatches; gradients summed in fp32, / steps at the end
        k = self.accum_steps
        mb = self.audio.shape[0] // k
        loss, flat = 0.0, None
        for i in range(k):
            li, _, g = torch_model.loss_and_grads(self.params, self.audio[i * mb:(i + 1) * mb], self.labels[i * mb:(i + 1) * mb],
                                                  self.conf, dtype=torch.float64)
            fi = torch.from_numpy(P.equinox_to_flat(g, self.conf)).to(torch.float32)
            flat = fi if flat is None else flat + fi
            loss += li
        flat, loss = flat / k, loss / k
        if self.poison:
            flat[self.split + 5] = float("nan")      # a non-finite gradient in the first half's range
        return float(loss), flat

    def full(self, seed):
        loss, flat = self._compute()
        b = self.bucket
        b.grads.copy_(flat)
        valid = 0.0 if self.poison else 1.0
        b.tail[0] = loss; b.tail[1] = valid; b.head[0] = valid

    def half(self, phase, seed):
        b = self.bucket
        if phase == 1:
            self._cache = self._compute()
            loss, flat = self._cache
            b.grads[self.split:].copy_(flat[self.split:])
            b.tail[0] = loss
            b.tail[1] = 0.0 if self.poison else 1.0
        else:
            _, flat = self._cache
            b.grads[:self.split].copy_(flat[:self.split])
            b.head[0] = 0.0 if self.poison else 1.0   # the whole step's flag (AND of both halves)

    def advance_seed(self):
        pass

    def apply(self, step, lr_base):
        # what amt_adamw_step_gated evaluates on the device: *valid_f32 >= 0.999
        self.gates.append(bool(self.bucket.valid_all[0] >= 0.999))


def _split_of(conf):
    return next(s.offset for s in P.device_layout(conf) if s.name == "norm.weight")


def _dp_worker(rank, world, port, q, poison, overlap, accum=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    conf = _conf()
    params = np_model.init_params(conf, seed=3, gamma=1.0)
    audio, labels = synth.make_batch(4, 320 * 20, seed=9)
    per = 4 // world
    shard = slice(rank * per, rank * per + per)
    bucket = train.GradBucket(P.param_count(conf), "cpu")
    split = _split_of(conf)
    be = _OracleBackend(conf, params, audio[shard], labels[shard], bucket, split, poison=poison and rank == world - 1, accum_steps=accum)
    dp = train.DataParallelStep(be, bucket, split, dist=dist, overlap=overlap, graph=True)   # graph is ignored on CPU
    dp.prepare(seed=0)
    dp.step()
    loss, valid = dp.result()
    q.put((rank, loss, valid, be.gates[-1], bucket.grads.numpy().copy() if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap,accum", [(False, 1), (True, 1), (False, 2)])
@pytest.mark.parametrize("poison", [False, True])
def test_data_parallel_step_two_ranks_equal_one_rank(poison, overlap, accum):
    """N-rank gradients == 1-rank gradients on the global batch, for the one-collective and the two-collective
    (overlap) form and with the rank's shard accumulated over two minibatches (train.py:275-310); the finite flags AND
    across ranks and close the optimizer gate on EVERY rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, poison, overlap, accum)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in range(2)], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    conf = _conf()
    params = np_model.init_params(conf, seed=3, gamma=1.0)
    audio, labels = synth.make_batch(4, 320 * 20, seed=9)
    ref_loss, _, g = torch_model.loss_and_grads(params, audio, labels, conf)
    ref = P.equinox_to_flat(g, conf)
    for rank, loss, valid, gate, _ in outs:
        assert valid == (not poison), (rank, valid)
        assert gate == (not poison), (rank, gate)          # the poisoned rank AND the healthy one skip the update
        assert abs(loss - ref_loss) < 1e-5 * abs(ref_loss)
    if not poison:
        grads = outs[0][4]
        assert np.linalg.norm(grads - ref) < 1e-5 * np.linalg.norm(ref)


@pytest.mark.parametrize("poison", [False, True])
def test_data_parallel_step_four_ranks_equal_one_rank(poison):
    """The same protocol at world size 4 (one clip per rank, two-collective form): the mean all-reduce over four shards equals
    the one-rank gradient of the global batch, and one poisoned rank closes the optimizer gate on all four."""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q, poison, True, 1)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in range(world)], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    conf = _conf()
    params = np_model.init_params(conf, seed=3, gamma=1.0)
    audio, labels = synth.make_batch(4, 320 * 20, seed=9)
    ref_loss, _, g = torch_model.loss_and_grads(params, audio, labels, conf)
    ref = P.equinox_to_flat(g, conf)
    for rank, loss, valid, gate, _ in outs:
        assert valid == (not poison) and gate == (not poison), (rank, valid, gate)
        assert abs(loss - ref_loss) < 1e-5 * abs(ref_loss)
    if not poison:
        assert np.linalg.norm(outs[0][4] - ref) < 1e-5 * np.linalg.norm(ref)


# ------------------------------------------------------------------ the in-place protocol (round 4)
class _OracleInPlaceBackend:
    """CPU stand-in for HipTrainBackend(fused=True) over GradBucket.of_model: the backward pass leaves RAW gradient sums
    (scale * accum_steps * the rank's mean gradient) and the rank's loss in the bucket, DataParallelStep SUM-reduces them in
    place, and apply() is what amt_adamw_step_reduced evaluates on the device: g = sums * (1 / world) / (scale * steps), the gate
    isfinite(|update|^2) && isfinite(loss sum), oracle/optim.py's chain behind it, the accumulator left zeroed, the flags
    published.  The ranges, the reduction and the order of the calls are the production code (DataParallelStep)."""
    fused = True
    in_place = True
    SCALE = 4.0

    def __init__(self, conf, params, audio, labels, bucket, split, poison=False, accum_steps=1):
        from oracle import optim
        self.conf, self.audio, self.labels = conf, audio, labels
        self.bucket, self.split, self.poison, self.accum_steps = bucket, split, poison, accum_steps
        self.world = 1                       # set by DataParallelStep
        self.flat = P.equinox_to_flat(params, conf).astype(np.float64)
        self.chain = optim.AdamWChain(self.flat.size)
        self.params_np = params
        self.applied = []
        self._cache = None

    def _compute(self):
        k = self.accum_steps
        mb = self.audio.shape[0] // k
        loss, raw = 0.0, None
        for i in range(k):
            li, _, g = torch_model.loss_and_grads(self.params_np, self.audio[i * mb:(i + 1) * mb], self.labels[i * mb:(i + 1) * mb],
                                                  self.conf, dtype=torch.float64)
            fi = torch.from_numpy(P.equinox_to_flat(g, self.conf)).to(torch.float32) * self.SCALE      # the scaled loss's gradient
            raw = fi if raw is None else raw + fi
            loss += li
        if self.poison:
            raw[self.split + 5] = float("inf")      # a non-finite gradient in the first half's range
        return float(loss / k), raw

    def full(self, seed):
        loss, raw = self._compute()
        self.bucket.grads.add_(raw)                  # the backward pass ADDS into the (clean) accumulator
        self.bucket.tail[0] = loss

    def half(self, phase, seed):
        b = self.bucket
        if phase == 1:
            self._cache = self._compute()
            loss, raw = self._cache
            b.grads[self.split:].add_(raw[self.split:])
            b.tail[0] = loss
        else:
            b.grads[:self.split].add_(self._cache[1][:self.split])

    def discard(self):
        self.bucket.grads.zero_()

    def advance_seed(self):
        pass

    def apply(self, step, lr_base):
        b = self.bucket
        b.loss_div = float(self.world)
        g = b.grads.numpy().astype(np.float64) * (1.0 / self.world) / (self.SCALE * self.accum_steps)
        trial = self.chain.__class__(self.flat.size)
        trial.mu, trial.nu, trial.count = self.chain.mu.copy(), self.chain.nu.copy(), self.chain.count
        with np.errstate(all="ignore"):
            new = trial.update(self.flat, g, lr_base)
            norm2 = float(np.sum((new - self.flat) ** 2))
        open_ = bool(np.isfinite(norm2) and np.isfinite(float(b.tail[0])))
        if open_:
            self.flat, self.chain = new, trial
        b.grads.zero_()                              # the commit pass leaves the accumulator zeroed in every case
        b.head[0] = b.tail[1] = 1.0 if open_ else 0.0
        self.applied.append(open_)


def _inplace_worker(rank, world, port, q, poison, overlap, accum, wire="fp32", inject=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    conf = _conf()
    params = np_model.init_params(conf, seed=3, gamma=1.0)
    audio, labels = synth.make_batch(4, 320 * 20, seed=9)
    per = 4 // world
    shard = slice(rank * per, rank * per + per)
    bucket = train.GradBucket(P.param_count(conf), "cpu")
    split = _split_of(conf)
    be = _OracleInPlaceBackend(conf, params, audio[shard], labels[shard], bucket, split, poison=poison and rank == world - 1,
                               accum_steps=accum)
    dp = train.DataParallelStep(be, bucket, split, dist=dist, overlap=overlap, graph=True, lr_base=1e-2, wire=wire)
    assert dp.in_place and be.world == world
    if inject is not None and rank == world - 1:
        dp._inject_failure = inject                        # this rank's trial step of the overlapped form "fails"
    dp.prepare(seed=0)
    assert float(bucket.grads.abs().max()) == 0.0          # the rehearsal's gradients were discarded
    dp.step()
    loss, valid = dp.result()
    if inject is not None:
        import bench
        q.put((rank, loss, valid, be.applied[-1], be.flat.copy(), float(bucket.grads.abs().max()), dp.overlap, dp.fallback,
               bench.dp_report_fields(dp)))
    else:
        q.put((rank, loss, valid, be.applied[-1], be.flat.copy(), float(bucket.grads.abs().max())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,overlap,accum", [(2, False, 1), (2, True, 1), (2, False, 2), (4, True, 1)])
@pytest.mark.parametrize("poison", [False, True])
def test_in_place_step_n_ranks_equal_one_rank(world, poison, overlap, accum):
    """The in-place protocol (include/amt.h: amt_grad_bucket / amt_adamw_step_reduced): SUM all-reduce of
    the raw gradient sums and the loss word on the accumulator itself, 1 / world inside the optimizer, no flag collective.
    N ranks end with the PARAMETERS one rank reaches on the global batch; an infinite gradient on one rank closes the gate
    on every rank through the reduced sum alone; the accumulator is left zeroed either way."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_inplace_worker, args=(r, world, port, q, poison, overlap, accum)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in range(world)], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle import optim
    conf = _conf()
    params = np_model.init_params(conf, seed=3, gamma=1.0)
    audio, labels = synth.make_batch(4, 320 * 20, seed=9)
    ref_loss, _, g = torch_model.loss_and_grads(params, audio, labels, conf)
    flat0 = P.equinox_to_flat(params, conf).astype(np.float64)
    ref_new = optim.AdamWChain(flat0.size).update(flat0, P.equinox_to_flat(g, conf).astype(np.float64), 1e-2)
    for rank, loss, valid, applied, flat, gmax in outs:
        assert valid == (not poison) and applied == (not poison), (rank, valid, applied)
        assert abs(loss - ref_loss) < 1e-5 * abs(ref_loss)
        assert gmax == 0.0
        if poison:
            assert np.array_equal(flat, flat0)                       # skipped on EVERY rank
        else:
            assert np.linalg.norm(flat - ref_new) < 1e-5 * np.linalg.norm(ref_new - flat0), rank
            assert np.array_equal(flat, outs[0][4])                  # replicas stay bit-identical


def test_a_failed_trial_step_makes_every_rank_fall_back_to_one_collective():
    """DataParallelStep.prepare() runs ONE trial step of the overlapped form when there is more than one
    rank.  A failure seen by ONE rank (here injected on the last rank) becomes the decision of EVERY rank through a MAX
    all-reduce of one word: all fall back, in the same process, to the one-collective in-place form - the one `north_star`
    describes - and say so in the bench line (`grad_allreduce_fallback`).  The step that follows is the plain step."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_inplace_worker, args=(r, world, port, q, False, True, 1, "fp32", "collective")) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in range(world)], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle import optim
    conf = _conf()
    params = np_model.init_params(conf, seed=3, gamma=1.0)
    audio, labels = synth.make_batch(4, 320 * 20, seed=9)
    _, _, g = torch_model.loss_and_grads(params, audio, labels, conf)
    flat0 = P.equinox_to_flat(params, conf).astype(np.float64)
    ref_new = optim.AdamWChain(flat0.size).update(flat0, P.equinox_to_flat(g, conf).astype(np.float64), 1e-2)
    for rank, loss, valid, applied, flat, gmax, overlap, fallback, fields in outs:
        assert overlap is False and fallback is not None, (rank, overlap, fallback)
        assert ("injected" in fallback) == (rank == world - 1)       # the other rank learnt it from the agreement word
        assert fields["grad_allreduce_fallback"] == fallback and "one bucket" in fields["grad_allreduce"]
        assert valid and applied and gmax == 0.0
        assert np.linalg.norm(flat - ref_new) < 1e-5 * np.linalg.norm(ref_new - flat0), rank
        assert np.array_equal(flat, outs[0][4])


@pytest.mark.parametrize("world,overlap", [(2, False), (4, True)])
@pytest.mark.parametrize("poison", [False, True])
def test_in_place_step_with_the_bf16_wire(world, overlap, poison):
    """wire="bf16" (VERDICT r3 item 2c): the gradient sums cross the ranks as bfloat16 - the reference's own message size,
    /root/reference/train.py:292-297 - through an all-to-all of shards, an fp32 sum on arrival and an all-gather of the reduced
    shards; the loss word stays fp32.  Tolerance: the parameters after one AdamW step agree with the one-rank step to 2 % of the
    step (each gradient element carries two bf16 roundings, 2^-9 relative each; Adam normalises the scale away); every rank
    holds bit-identical parameters; an infinite gradient on one rank still closes every rank's gate (inf survives bf16)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_inplace_worker, args=(r, world, port, q, poison, overlap, 1, "bf16")) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in range(world)], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle import optim
    conf = _conf()
    params = np_model.init_params(conf, seed=3, gamma=1.0)
    audio, labels = synth.make_batch(4, 320 * 20, seed=9)
    ref_loss, _, g = torch_model.loss_and_grads(params, audio, labels, conf)
    flat0 = P.equinox_to_flat(params, conf).astype(np.float64)
    ref_new = optim.AdamWChain(flat0.size).update(flat0, P.equinox_to_flat(g, conf).astype(np.float64), 1e-2)
    for rank, loss, valid, applied, flat, gmax in outs:
        assert valid == (not poison) and applied == (not poison), (rank, valid, applied)
        assert abs(loss - ref_loss) < 1e-5 * abs(ref_loss)                 # the loss word is not rounded
        assert gmax == 0.0
        if poison:
            assert np.array_equal(flat, flat0)
        else:
            assert np.linalg.norm(flat - ref_new) < 2e-2 * np.linalg.norm(ref_new - flat0), rank
            assert np.array_equal(flat, outs[0][4])


def test_grad_bucket_layout_and_single_rank_step():
    b = train.GradBucket(10, "cpu")
    assert b.buf.numel() == 16 and b.grads.data_ptr() == b.buf.data_ptr() + 16
    assert b.first_range(6).numel() == 4 + 2 and b.second_range(6).numel() == 4 + 6
    assert b.first_range(6).data_ptr() == b.second_range(6).data_ptr() + 4 * (4 + 6)   # the two ranges tile the bucket

    class _B:
        def __init__(self, bucket): self.bucket, self.applied = bucket, []
        def full(self, seed):
            self.bucket.grads.fill_(float(seed)); self.bucket.tail[0] = 7.0; self.bucket.tail[1] = 1.0; self.bucket.head[0] = 1.0
        def advance_seed(self): pass
        def apply(self, step, lr): self.applied.append(step)
    be = _B(b)
    dp = train.DataParallelStep(be, b, 6, dist=None, overlap=False)
    dp.prepare(seed=3)
    dp.step(); dp.step()
    assert be.applied == [1, 2] and dp.result() == (7.0, True)      # optimizer steps count from 1 (optax)
    assert b.grads.tolist() == [2.0] * 10                            # eager steps use the step number as the seed
