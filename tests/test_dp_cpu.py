"""Multi-process (world_size 2, gloo, CPU tensors) test of the data-parallel step protocol:
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


def _oracle_step(conf, params):
    def step(a, y, scale, seed):
        loss, _, g = torch_model.loss_and_grads(params, a.numpy(), y.numpy(), conf, scale=scale, dtype=torch.float64)
        flat = torch.from_numpy(P.equinox_to_flat({k: v / scale for k, v in g.items()}, conf))
        return loss / scale, loss, flat, True
    return step


def _worker(rank, world, port, q, poison):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    conf = _conf()
    params = np_model.init_params(conf, seed=3, gamma=1.0)
    audio, labels = synth.make_batch(4, 320 * 20, seed=9)
    shard = slice(rank * 2, rank * 2 + 2)
    bucket = train.new_bucket(P.param_count(conf), "cpu")
    step = _oracle_step(conf, params)
    if poison and rank == 1:
        inner = step
        step = lambda a, y, s, sd: (lambda o: (o[0], o[1], o[2], False))(inner(a, y, s, sd))
    loss, scaled, valid = train.compute_training_step(
        None, torch.from_numpy(audio[shard]), torch.from_numpy(labels[shard]), grad_scale=4.0, seed=1,
        bucket=bucket, lr_base=0.0, step=1, dist=dist, accum_steps=2, apply_update=False, local_step=step)
    if rank == 0:
        q.put((loss, scaled, valid, bucket[:-train.BUCKET_TAIL].numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("poison", [False, True])
def test_two_ranks_equal_one_rank(poison):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, poison)) for r in range(2)]
    for p in procs:
        p.start()
    loss, scaled, valid, grads = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    conf = _conf()
    params = np_model.init_params(conf, seed=3, gamma=1.0)
    audio, labels = synth.make_batch(4, 320 * 20, seed=9)
    ref_loss, _, g = torch_model.loss_and_grads(params, audio, labels, conf)
    ref = P.equinox_to_flat(g, conf)
    assert valid == (not poison)
    assert abs(loss - ref_loss) < 1e-6 * abs(ref_loss) and abs(scaled - 4 * ref_loss) < 1e-5 * abs(ref_loss)
    assert np.linalg.norm(grads - ref) < 1e-5 * np.linalg.norm(ref)


def test_single_process_bucket_and_loss_scale_controller():
    b = train.new_bucket(5, "cpu")
    b[:5] = torch.arange(5.0)
    train.pack_tail(b, 2.5, 10.0, 1.0)
    g, loss, scaled, valid = train.allreduce_bucket(b, None)
    assert g.tolist() == [0, 1, 2, 3, 4] and (loss, scaled, valid) == (2.5, 10.0, True)
    c = train.LossScaleController(scale=8.0)
    assert c.update(True, 100.0, 800.0) is False and c.scale == 16.0         # scaled_loss < 1e4: double
    assert c.update(True, 100.0, 20000.0) is False and c.scale == 16.0
    assert c.update(False, 100.0, 800.0) is True and c.scale == 8.0          # non-finite grads: halve + skip
    assert c.update(True, float("nan"), 1.0) is True and c.scale == 4.0
    off = train.LossScaleController(enabled=False)
    assert off.update(True, 1.0, 1.0) is False and off.scale == 1.0


# ------------------------------------------------------------------ DataParallelStep (the protocol bench.py runs)
class _OracleBackend:
    """CPU stand-in for HipTrainBackend: the oracle computes the rank's gradients, the bucket / flag / collective /
    gate logic around it is the production code (audio_to_midi_amd.train.DataParallelStep)."""

    def __init__(self, conf, params, audio, labels, bucket, split, poison=False):
        self.conf, self.params, self.audio, self.labels = conf, params, audio, labels
        self.bucket, self.split, self.poison = bucket, split, poison
        self.gates = []
        self._cache = None

    def _compute(self):
        loss, _, g = torch_model.loss_and_grads(self.params, self.audio, self.labels, self.conf, dtype=torch.float64)
        flat = torch.from_numpy(P.equinox_to_flat(g, self.conf)).to(torch.float32)
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


def _dp_worker(rank, world, port, q, poison, overlap):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    conf = _conf()
    params = np_model.init_params(conf, seed=3, gamma=1.0)
    audio, labels = synth.make_batch(4, 320 * 20, seed=9)
    shard = slice(rank * 2, rank * 2 + 2)
    bucket = train.GradBucket(P.param_count(conf), "cpu")
    split = _split_of(conf)
    be = _OracleBackend(conf, params, audio[shard], labels[shard], bucket, split, poison=poison and rank == 1)
    dp = train.DataParallelStep(be, bucket, split, dist=dist, overlap=overlap, graph=True)   # graph is ignored on CPU
    dp.prepare(seed=0)
    dp.step()
    loss, valid = dp.result()
    q.put((rank, loss, valid, be.gates[-1], bucket.grads.numpy().copy() if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("poison", [False, True])
def test_data_parallel_step_two_ranks_equal_one_rank(poison, overlap):
    """N-rank gradients == 1-rank gradients on the global batch, for the one-collective and the two-collective
    (overlap) form; the finite flags AND across ranks and close the optimizer gate on EVERY rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, poison, overlap)) for r in range(2)]
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
