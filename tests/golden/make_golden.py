"""Generates tests/golden/*.npz.  Run from the repository root:  python tests/golden/make_golden.py

PARITY UNPINNED: the reference (JAX / Equinox) cannot run offline, so these vectors
come from this repository's float64 oracle (oracle/np_model.py for logits and loss,
oracle/torch_model.py autograd for gradients), not from the reference.  They freeze the oracle:
a later edit that changes its results fails tests/test_golden.py, and the HIP path is compared
against the same numbers on the GPU box.

Inputs are regenerated from seeds (audio_to_midi_amd.synth.make_batch, np_model.init_params);
the fixture stores checksums of them so a silent change of a generator is detected as well.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from audio_to_midi_amd import synth  # noqa: E402
from oracle import np_model, post, torch_model  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (depths, layer pairs, frames, batch, weight seed, data seed)
    "reduced_s50": ([1, 1, 1, 1, 1, 2, 1], 2, 50, 2, 7, 3),      # padded local windows (6 pad rows)
    "reduced_s32": ([1, 1, 1, 1, 1, 2, 1], 2, 32, 2, 7, 3),      # pad-free local windows
    "reduced_s250": ([1, 1, 1, 1, 1, 1, 1], 1, 250, 1, 11, 5),   # the reference's 5 s window geometry (31 windows)
    # round 3: the benchmark shape per clip - full depth (39 Blocks, 16 layers), 10 s = 500 frames (62 local windows,
    # RoPE rows beyond the reference's 300-row table), one clip; of the stacked attention leaf only layer pair 0 is stored
    "full_s500": ([3, 3, 3, 3, 3, 21, 3], 8, 500, 1, 1234, 42),
}


def case_conf(depths, pairs):
    conf = dict(np_model.MODEL_CONFIG)
    conf["depths"] = list(depths)
    conf["num_transformer_layers"] = pairs
    return conf


def build(name):
    depths, pairs, S, B, wseed, dseed = CASES[name]
    conf = case_conf(depths, pairs)
    params = np_model.init_params(conf, seed=wseed, gamma=1.0)
    audio, labels = synth.make_batch(B, 320 * S, seed=dseed)
    logits = np.stack([np_model.forward(params, audio[b].astype(np.float64), conf)[0] for b in range(B)])
    loss = np_model.batch_loss(params, audio.astype(np.float64), labels.astype(np.float64), conf)
    loss_t, logits_t, grads = torch_model.loss_and_grads(params, audio, labels, conf)
    assert np.abs(logits - logits_t).max() < 1e-9 and abs(loss - loss_t) < 1e-9 * abs(loss)
    keys = list(grads)
    out = {
        "logits": logits.astype(np.float32), "loss": np.float64(loss),
        "grad_names": np.array(keys),
        "grad_norms": np.array([np.linalg.norm(grads[k]) for k in keys]),
        "audio_sum": np.float64(audio.astype(np.float64).sum()), "labels_sum": np.float64(labels.astype(np.float64).sum()),
        "params_sum": np.float64(sum(float(v.sum()) for v in params.values())),
        # three whole gradient leaves: first conv, one attention projection, the head
        "grad.layers.0.0.conv.weight": grads["layers.0.0.conv.weight"].astype(np.float32),
        "grad.decoder.decoder_pooling.weight": grads["decoder.decoder_pooling.weight"].astype(np.float32),
    }
    k_attn = [k for k in keys if k.endswith("kv_down_proj.weight")][0]
    out["grad_attn_name"] = np.array(k_attn)
    out["grad_attn"] = (grads[k_attn][:1] if name.startswith("full") else grads[k_attn]).astype(np.float32)
    # round 4: the same step with the GEMM weights rounded to bf16 (what a bf16 context multiplies with): the comparison
    # target of the bf16 HIP path, so that its tolerance covers the arithmetic and not the weight rounding (which alone
    # moves the full-depth logits by ~1e-1)
    loss_w, logits_w, grads_w = torch_model.loss_and_grads(torch_model.round_gemm_weights(params), audio, labels, conf)
    out["bf16w.logits"] = logits_w.astype(np.float32)
    out["bf16w.loss"] = np.float64(loss_w)
    out["bf16w.grad.layers.0.0.conv.weight"] = grads_w["layers.0.0.conv.weight"].astype(np.float32)
    out["bf16w.grad.decoder.decoder_pooling.weight"] = grads_w["decoder.decoder_pooling.weight"].astype(np.float32)
    out["bf16w.grad_attn"] = (grads_w[k_attn][:1] if name.startswith("full") else grads_w[k_attn]).astype(np.float32)
    return out


def build_post():
    """Post-processing vectors (Python restatement of rust-plugins/src/common.rs:13-144)."""
    rng = np.random.default_rng(2024)
    windows = rng.uniform(0, 1, (4, 250, 90)).astype(np.float32)
    stitched = post.stitch_probs(windows, 0.5, 0.02)
    track = rng.uniform(0.0, 0.08, (400, 90)).astype(np.float32)
    for _ in range(30):
        k = rng.integers(0, 90); s = rng.integers(0, 380); d = rng.integers(2, 90)
        e = min(400, s + d)
        track[s:e, k] = np.maximum(np.exp(-0.05 * np.arange(e - s)), 0.6) * rng.uniform(0.7, 1.0)
    events = np.array(post.extract_events(track), dtype=np.int64).reshape(-1, 4)
    return {"windows": windows, "stitched": stitched, "track": track, "events": events}


if __name__ == "__main__":
    for name in (sys.argv[1:] or CASES):
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **build(name))
        print("wrote", name)
    if len(sys.argv) == 1:
        np.savez_compressed(os.path.join(HERE, "post.npz"), **build_post())
        print("wrote post")
