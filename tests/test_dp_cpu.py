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
