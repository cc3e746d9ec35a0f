"""Oracle #1: literal numpy-float64 restatement of the reference forward pass + loss.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the
reference has no tests/fixtures for this path and cannot run here.

Layout follows the reference exactly: channel-first ``(C, L)`` in the CNN,
``(S, H)`` in the transformer, one clip at a time (the reference vmaps over the
batch, train.py:56-58).  Parameters are a flat ``dict`` keyed by the Equinox
pytree paths ("Table P" of SURVEY.md §8a), in Equinox shapes.

Third-party semantics restated here (equinox / jax / optax are not vendored in
the reference; requirements.txt:1,3,6 leaves them unpinned):
  * eqx.nn.Conv1d   = cross-correlation, weight (out, in/groups, k), bias (out,1)
  * eqx.nn.LayerNorm= biased variance, eps 1e-5, elementwise affine
  * eqx.nn.Linear   = W @ x + b, weight (out, in)
  * jax.nn.gelu     = tanh approximation (approximate=True is the JAX default)
  * x.at[idx].add   = out-of-bounds updates are dropped
  * optax.sigmoid_binary_cross_entropy(z, y) = -y*log_sigmoid(z) - (1-y)*log_sigmoid(-z)
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

# model.py:20-34
MODEL_CONFIG = {
    "dims": [4 * (2 ** i) for i in range(7)],
    "depths": [3, 3, 3, 3, 3, 21, 3],
    "cnn_hidden_expansion": 2.0,
    "num_transformer_layers": 8,
    "num_transformer_heads": 4,
    "attention_size": 64,
    "compressed_attention_q_size": 64,
    "compressed_attention_kv_size": 64,
    "transformer_dropout_rate": 0.1,
    "transformer_hidden_expansion": 2.0,
    "sdd_rate": 0.1,
}
MIDI_EVENT_VOCCAB_SIZE = 90  # audio_to_midi_dataset.py:26
LOCAL_CONTEXT_WINDOW = 16    # model.py:635
LN_EPS = 1e-5                # eqx.nn.LayerNorm default


# --------------------------------------------------------------------------- params
def param_shapes(conf=MODEL_CONFIG) -> "OrderedDict[str, tuple]":
    """Equinox leaf shapes, in pytree order (model.py:673-738, Table P)."""
    dims, depths = conf["dims"], conf["depths"]
    hid = [int(d * conf["cnn_hidden_expansion"]) for d in dims]
    H = dims[-1]
    nl = conf["num_transformer_layers"]
    nh, hd = conf["num_transformer_heads"], conf["attention_size"]
    ckv = conf["compressed_attention_kv_size"]
    inter = int(H * conf["transformer_hidden_expansion"])
    sh: "OrderedDict[str, tuple]" = OrderedDict()
    for s in range(len(dims)):
        p = f"layers.{s}.0."
        if s == 0:  # Stem, model.py:84-100
            sh[p + "conv.weight"] = (dims[0], 2, 5)
            sh[p + "conv.bias"] = (dims[0], 1)
            sh[p + "norm.weight"] = (dims[0],)
            sh[p + "norm.bias"] = (dims[0],)
        else:       # Downsample, model.py:102-118
            sh[p + "conv.weight"] = (dims[s], dims[s - 1], 2)
            sh[p + "conv.bias"] = (dims[s], 1)
            sh[p + "norm.weight"] = (dims[s - 1],)
            sh[p + "norm.bias"] = (dims[s - 1],)
        for j in range(1, depths[s] + 1):  # Block, model.py:120-167
            q = f"layers.{s}.{j}."
            C, Hc = dims[s], hid[s]
            sh[q + "depth_conv.weight"] = (C, 1, 7)
            sh[q + "depth_conv.bias"] = (C, 1)
            sh[q + "norm.weight"] = (C,)
            sh[q + "norm.bias"] = (C,)
            sh[q + "point_conv_1.weight"] = (Hc, C, 1)
            sh[q + "point_conv_1.bias"] = (Hc, 1)
            sh[q + "point_conv_2.weight"] = (C, Hc, 1)
            sh[q + "point_conv_2.bias"] = (C, 1)
            sh[q + "gamma"] = (C,)
    sh["norm.weight"] = (H,)
    sh["norm.bias"] = (H,)
    for kind in ("local_attention", "global_attention"):
        t = f"transformer.layers.{kind}."
        a = t + ("attention_block.self_attention." if kind == "local_attention" else "attention_block.")
        sh[t + "attention_norm.weight"] = (nl, H)
        sh[t + "attention_norm.bias"] = (nl, H)
        sh[a + "query_up_proj.weight"] = (nl, nh * hd, H)
        sh[a + "kv_down_proj.weight"] = (nl, ckv, H)
        sh[a + "key_up_proj.weight"] = (nl, nh * hd, ckv)
        sh[a + "value_up_proj.weight"] = (nl, nh * hd, ckv)
        sh[a + "output_proj.weight"] = (nl, H, nh * hd)
        sh[t + "feed_forward_norm.weight"] = (nl, H)
        sh[t + "feed_forward_norm.bias"] = (nl, H)
        f = t + "feed_forward_block."
        sh[f + "attention_to_intermediate_proj.weight"] = (nl, 2 * inter, H)
        sh[f + "attention_to_intermediate_proj.bias"] = (nl, 2 * inter)
        sh[f + "intermediate_to_attention_proj.weight"] = (nl, H, inter)
        sh[f + "intermediate_to_attention_proj.bias"] = (nl, H)
    sh["decoder.norm.weight"] = (H,)
    sh["decoder.norm.bias"] = (H,)
    sh["decoder.decoder_pooling.weight"] = (MIDI_EVENT_VOCCAB_SIZE, H)
    sh["decoder.decoder_pooling.bias"] = (MIDI_EVENT_VOCCAB_SIZE,)
    return sh


def init_params(conf=MODEL_CONFIG, seed: int = 1234, gamma: float = 1e-6) -> "OrderedDict[str, np.ndarray]":
    """Equinox-distribution seeded init: conv/linear weight & bias ~ U(+-1/sqrt(fan_in)),
    fan_in = (in/groups)*k; LayerNorm weight 1 / bias 0; layer-scale gamma (model.py:157-158).
    The random stream is numpy's, not JAX's (threefry streams cannot be reproduced here)."""
    rng = np.random.default_rng(seed)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    shapes = param_shapes(conf)
    for name, shp in shapes.items():
        leaf = name.rsplit(".", 1)[-1]
        if name.endswith("norm.weight"):
            out[name] = np.ones(shp)
        elif name.endswith("norm.bias"):
            out[name] = np.zeros(shp)
        elif leaf == "gamma":
            out[name] = np.full(shp, gamma)
        elif leaf == "weight":
            stacked = name.startswith("transformer.")
            core = shp[1:] if stacked else shp
            fan_in = int(np.prod(core[1:]))
            lim = 1.0 / math.sqrt(fan_in)
            out[name] = rng.uniform(-lim, lim, size=shp)
        elif leaf == "bias":
            wshape = shapes[name[: -len("bias")] + "weight"]
            stacked = name.startswith("transformer.")
            core = wshape[1:] if stacked else wshape
            fan_in = int(np.prod(core[1:]))
            lim = 1.0 / math.sqrt(fan_in)
            out[name] = rng.uniform(-lim, lim, size=shp)
        else:
            raise AssertionError(name)
    return out


def num_params(conf=MODEL_CONFIG) -> int:
    return sum(int(np.prod(s)) for s in param_shapes(conf).values())


# --------------------------------------------------------------------------- rope
def precompute_frequencies(dim: int, max_pos: int, theta: float = 10000.0):
    """rope.py:12-22 — float32 arithmetic, like the reference (jnp float32)."""
    inv_freq = (np.float32(1.0) /
                (np.float32(theta) ** (np.arange(0, dim, 2, dtype=np.float32)[: dim // 2] / np.float32(dim)))
                ).astype(np.float32)
    t = np.arange(0, max_pos, dtype=np.float32)
    freqs = np.outer(t, inv_freq).astype(np.float32)
    return np.cos(freqs).astype(np.float32), np.sin(freqs).astype(np.float32)


def calculate_rope(x, cos, sin):
    """rope.py:25-53.  x: (S, heads, hd); interleaved (even, odd) pairs."""
    S = x.shape[0]
    c = cos[:S][:, None, :].astype(np.float64)
    s = sin[:S][:, None, :].astype(np.float64)
    x1 = x[..., 0::2]
    x2 = x[..., 1::2]
    out = np.stack([x1 * c - x2 * s, x1 * s + x2 * c], axis=-1)
    return out.reshape(x.shape)


# --------------------------------------------------------------------------- layers
def gelu_tanh(x):
    return 0.5 * x * (1.0 + np.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def layer_norm(x, w, b, axis):
    mu = x.mean(axis=axis, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=axis, keepdims=True)
    xh = (x - mu) / np.sqrt(var + LN_EPS)
    shp = [1] * x.ndim
    shp[axis] = -1
    return xh * w.reshape(shp) + b.reshape(shp)


def conv1d(x, w, b, stride=1, pad=(0, 0), groups=1):
    """Cross-correlation.  x (Cin, L), w (Cout, Cin/groups, k), b (Cout, 1)."""
    cin, L = x.shape
    cout, cing, k = w.shape
    xp = np.pad(x, ((0, 0), pad))
    Lo = (xp.shape[1] - k) // stride + 1
    out = np.zeros((cout, Lo))
    if groups == 1:
        for t in range(k):
            out += w[:, :, t] @ xp[:, t: t + stride * (Lo - 1) + 1: stride]
    else:
        assert groups == cin == cout and cing == 1
        for t in range(k):
            out += w[:, 0, t][:, None] * xp[:, t: t + stride * (Lo - 1) + 1: stride]
    return out + b


def stem(p, pre, x):
    """model.py:98-100"""
    out = conv1d(x, p[pre + "conv.weight"], p[pre + "conv.bias"], stride=5)
    return layer_norm(out, p[pre + "norm.weight"], p[pre + "norm.bias"], axis=0)


def downsample(p, pre, x):
    """model.py:116-118"""
    out = layer_norm(x, p[pre + "norm.weight"], p[pre + "norm.bias"], axis=0)
    return conv1d(out, p[pre + "conv.weight"], p[pre + "conv.bias"], stride=2)


def block(p, pre, x):
    """model.py:160-167 (stochastic depth never active: Sequential forwards only `key`)."""
    C = x.shape[0]
    out = conv1d(x, p[pre + "depth_conv.weight"], p[pre + "depth_conv.bias"], pad=(3, 3), groups=C)
    out = layer_norm(out, p[pre + "norm.weight"], p[pre + "norm.bias"], axis=0)
    out = conv1d(out, p[pre + "point_conv_1.weight"], p[pre + "point_conv_1.bias"])
    out = gelu_tanh(out)
    out = conv1d(out, p[pre + "point_conv_2.weight"], p[pre + "point_conv_2.bias"])
    out = p[pre + "gamma"][:, None] * out
    return out + x


def dot_product_attention(q, k, v, keep_mask=None, keep_prob=1.0):
    """model.py:241-257.  q,k,v: (S, hd).  keep_mask: optional (Sq, Sk) 0/1 dropout mask."""
    q = q / math.sqrt(q.shape[-1])
    logits = q @ k.T
    logits = logits - logits.max(axis=-1, keepdims=True)
    w = np.exp(logits)
    w = w / w.sum(axis=-1, keepdims=True)
    if keep_mask is not None:
        w = w * keep_mask / keep_prob
    return w @ v


def self_attention(lp, x, cos, sin, nh, keep_masks=None, keep_prob=1.0):
    """model.py:340-369.  lp: dict with Wq, Wkvd, Wku, Wvu, Wo for this layer."""
    S = x.shape[0]
    q = calculate_rope((x @ lp["Wq"].T).reshape(S, nh, -1), cos, sin)
    c = x @ lp["Wkvd"].T
    k = calculate_rope((c @ lp["Wku"].T).reshape(S, nh, -1), cos, sin)
    v = (c @ lp["Wvu"].T).reshape(S, nh, -1)
    heads = []
    for h in range(nh):
        km = None if keep_masks is None else keep_masks[h]
        heads.append(dot_product_attention(q[:, h], k[:, h], v[:, h], km, keep_prob))
    attn = np.stack(heads, axis=1).reshape(S, -1)
    return attn @ lp["Wo"].T


def local_window_geometry(seq_len, window=LOCAL_CONTEXT_WINDOW):
    """model.py:417-433: (pad_left, pad_right, num_windows, stride)."""
    stride = window // 2
    required = stride - (seq_len - window) % stride
    pl = pr = 0
    if required != stride:
        pl = required // 2
        pr = required // 2 + (required % 2)
    nwin = (seq_len + pl + pr - window) // stride + 1
    return pl, pr, nwin, stride


def local_self_attention(lp, x, cos, sin, nh, keep_masks=None, keep_prob=1.0):
    """model.py:409-471 including the padded-coordinate scatter into an unpadded buffer."""
    S, H = x.shape
    W = LOCAL_CONTEXT_WINDOW
    pl, pr, nwin, stride = local_window_geometry(S, W)
    xp = np.pad(x, ((pl, pr), (0, 0)))
    output = np.zeros((S, H))
    count = np.zeros((S,))
    for wi in range(nwin):
        st = wi * stride
        km = None if keep_masks is None else keep_masks[wi]
        ow = self_attention(lp, xp[st: st + W], cos, sin, nh, km, keep_prob)
        for off in range(W):
            idx = st + off  # padded coordinate, scattered into the UNPADDED buffer
            if idx < S:     # JAX scatter drops out-of-bounds updates
                output[idx] += ow[off]
                count[idx] += 1
    return output / count[:, None]


def feed_forward(lp, x):
    """model.py:226-238 (dropout handled by caller)."""
    y = x @ lp["W1"].T + lp["b1"]
    half = y.shape[-1] // 2
    h = gelu_tanh(y[..., :half]) * y[..., half:]
    return h @ lp["W2"].T + lp["b2"]


def _layer_params(p, kind, i):
    t = f"transformer.layers.{kind}."
    a = t + ("attention_block.self_attention." if kind == "local_attention" else "attention_block.")
    f = t + "feed_forward_block."
    return {
        "ln1w": p[t + "attention_norm.weight"][i], "ln1b": p[t + "attention_norm.bias"][i],
        "Wq": p[a + "query_up_proj.weight"][i], "Wkvd": p[a + "kv_down_proj.weight"][i],
        "Wku": p[a + "key_up_proj.weight"][i], "Wvu": p[a + "value_up_proj.weight"][i],
        "Wo": p[a + "output_proj.weight"][i],
        "ln2w": p[t + "feed_forward_norm.weight"][i], "ln2b": p[t + "feed_forward_norm.bias"][i],
        "W1": p[f + "attention_to_intermediate_proj.weight"][i],
        "b1": p[f + "attention_to_intermediate_proj.bias"][i],
        "W2": p[f + "intermediate_to_attention_proj.weight"][i],
        "b2": p[f + "intermediate_to_attention_proj.bias"][i],
    }


def transformer_layer(lp, x, cos, sin, nh, local, drop=None):
    """model.py:529-556.  drop: None or dict(attn=masks, ffn=(S,H) mask, keep=prob)."""
    xn = layer_norm(x, lp["ln1w"], lp["ln1b"], axis=1)
    am = None if drop is None else drop["attn"]
    kp = 1.0 if drop is None else drop["keep"]
    if local:
        r = local_self_attention(lp, xn, cos, sin, nh, am, kp)
    else:
        r = self_attention(lp, xn, cos, sin, nh, am, kp)
    h = x + r
    hn = layer_norm(h, lp["ln2w"], lp["ln2b"], axis=1)
    r = feed_forward(lp, hn)
    if drop is not None:
        r = r * drop["ffn"] / kp
    return h + r


def cnn_frontend(p, conf, samples):
    """model.py:755-762: 7 stages, final LayerNorm, transpose -> (S, H)."""
    h = samples
    for s in range(len(conf["dims"])):
        h = stem(p, f"layers.{s}.0.", h) if s == 0 else downsample(p, f"layers.{s}.0.", h)
        for j in range(1, conf["depths"][s] + 1):
            h = block(p, f"layers.{s}.{j}.", h)
    h = layer_norm(h, p["norm.weight"], p["norm.bias"], axis=0)
    return h.T


def forward(p, samples, conf=MODEL_CONFIG, rope=None, dropout_masks=None):
    """OutputSequenceGenerator.__call__ for ONE clip (model.py:740-769).

    samples: (2, T) float.  Returns (logits (S,90), probs (S,90)) in float64.
    dropout_masks: None (inference / p=0) or a list, per transformer layer
    (local0, global0, local1, ...), of dict(attn=..., ffn=..., keep=...).
    """
    p = {k: np.asarray(v, dtype=np.float64) for k, v in p.items()}
    x = np.asarray(samples, dtype=np.float64)
    h = cnn_frontend(p, conf, x)
    S = h.shape[0]
    nh, hd = conf["num_transformer_heads"], conf["attention_size"]
    if rope is None:
        rope = precompute_frequencies(hd, max(S, 300))  # train.py:786 uses 300
    cos, sin = rope
    li = 0
    for i in range(conf["num_transformer_layers"]):
        for kind in ("local_attention", "global_attention"):  # model.py:609-611
            lp = _layer_params(p, kind, i)
            dm = None if dropout_masks is None else dropout_masks[li]
            h = transformer_layer(lp, h, cos, sin, nh, kind == "local_attention", dm)
            li += 1
    hn = layer_norm(h, p["decoder.norm.weight"], p["decoder.norm.bias"], axis=1)  # model.py:190
    logits = hn @ p["decoder.decoder_pooling.weight"].T + p["decoder.decoder_pooling.bias"]
    probs = 1.0 / (1.0 + np.exp(-logits))
    return logits, probs


def log_sigmoid(z):
    return np.minimum(z, 0.0) - np.log1p(np.exp(-np.abs(z)))


def loss_from_output(logits, labels, scale=1.0):
    """train.py:39-47: sigmoid BCE, elementwise, * scale, SUM over (frames, keys)."""
    l = -labels * log_sigmoid(logits) - (1.0 - labels) * log_sigmoid(-logits)
    return float(np.sum(l * scale))


def batch_loss(p, audio, labels, conf=MODEL_CONFIG, scale=1.0, rope=None):
    """train.py:49-62: per-clip loss, MEAN over the batch."""
    losses = []
    for b in range(audio.shape[0]):
        logits, _ = forward(p, audio[b], conf, rope)
        losses.append(loss_from_output(logits, np.asarray(labels[b], dtype=np.float64), scale))
    return float(np.mean(losses))
