"""The transformer stack's forward as ONE XCD-team launch (csrc/tlayer.hip) against the launch-per-operator chains
(AMT_TLAYER=0): the same saved tensors bit for bit, so logits are equal and the gradients - computed by the same backward
kernels from those tensors - differ only by the summation order of the weight-gradient atomics.
Reference path: /root/reference/model.py:529-556 (TransformerLayer), :559-670 (stack)."""
import numpy as np
import pytest
import torch

import audio_to_midi_amd as amt
from audio_to_midi_amd import synth
from oracle import np_model

pytestmark = pytest.mark.gpu


def _run(conf, params, audio, labels, dtype, tlayer, monkeypatch, train=True):
    monkeypatch.setenv("AMT_TLAYER", "1" if tlayer else "0")
    model = amt.OutputSequenceGenerator(conf, dtype=dtype)
    model.load_equinox_params(params)
    assert model.tlayer_active() == bool(tlayer)
    out = model.loss_and_grads(torch.from_numpy(audio), torch.from_numpy(labels), seed=5, train=train, return_logits=True)
    torch.cuda.synchronize()
    assert model.team_status() == 0
    res = (out.logits.clone(), float(out.loss), out.grads_flat.clone())
    model.close()
    return res


# (dtype, clips, frames): float32 runs 32-row token groups (231 frames: 8 groups x 4 heads = 32 roles, the last group 7 rows;
# 500 frames: 16 groups = 64 roles on 32 workgroups, two roles per phase); bfloat16 runs 64-row groups (500 frames: the
# benchmark's clip, 8 groups, 52 rows in the last; 333: six groups, 13 rows in the last).  9 clips: one team takes two clips.
@pytest.mark.parametrize("dtype,B,S", [("float32", 3, 231), ("float32", 2, 500), ("bfloat16", 3, 500), ("bfloat16", 9, 333)])
@pytest.mark.parametrize("train", [True, False])
def test_team_layer_forward_equals_the_chain_launches(small_conf, dtype, B, S, train, monkeypatch):
    conf = dict(small_conf)
    params = np_model.init_params(conf, seed=21, gamma=1.0)
    audio, labels = synth.make_batch(B, 320 * S, seed=9)
    a = _run(conf, params, audio, labels, dtype, True, monkeypatch, train)
    b = _run(conf, params, audio, labels, dtype, False, monkeypatch, train)
    assert torch.isfinite(a[0]).all()
    assert torch.equal(a[0], b[0]), float((a[0] - b[0]).abs().max())
    assert abs(a[1] - b[1]) <= 1e-6 * abs(b[1])
    rel = float((a[2] - b[2]).norm() / b[2].norm())
    assert rel < (1e-5 if dtype == "float32" else 1e-3), rel


def test_team_layer_forward_full_depth_bench_shape(monkeypatch):
    """Full depth (16 layers), 8 clips x 10 s, bf16, train mode: the benchmark's launch."""
    conf = dict(np_model.MODEL_CONFIG)
    conf["depths"] = [1, 1, 1, 1, 1, 1, 1]          # (the CNN is not what this test is about)
    params = np_model.init_params(conf, seed=3, gamma=1.0)
    audio, labels = synth.make_batch(8, 160000, seed=4)
    a = _run(conf, params, audio, labels, "bfloat16", True, monkeypatch)
    b = _run(conf, params, audio, labels, "bfloat16", False, monkeypatch)
    assert torch.equal(a[0], b[0]), float((a[0] - b[0]).abs().max())
    rel = float((a[2] - b[2]).norm() / b[2].norm())
    assert rel < 1e-3, rel
