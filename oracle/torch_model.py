"""Oracle #2: independent torch-CPU restatement from stock ops, batched, with autograd.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED.

Differs on purpose from oracle/np_model.py in how it is written (F.conv1d,
F.layer_norm, F.gelu(approximate="tanh"), unfold + index_add_ for the local
windows, autograd for gradients) so that the two restatements check each other.
Used (a) in float64 as the gradient oracle, (b) in float32 with all host cores
as bench.py's reported ``cpu_baseline`` ("torch-CPU restatement of the
reference; JAX is unavailable offline").
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .np_model import (LOCAL_CONTEXT_WINDOW, LN_EPS, MODEL_CONFIG,
                       local_window_geometry, precompute_frequencies)


def to_torch(params, dtype=torch.float64, requires_grad=False):
    out = {}
    for k, v in params.items():
        t = torch.as_tensor(v).to(dtype).clone()
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


def round_gemm_weights(params, dtype=torch.bfloat16):
    """The parameters a reduced-precision context computes with: the operands of its GEMMs (every leaf with >= 2 dims except
    norms, biases, gamma, the depthwise taps and the stem conv, which the kernels read in fp32) rounded to `dtype`.  For the
    oracle side of a bf16 comparison, so that it measures the arithmetic and not the rounding of the weights (the reference
    rounds its weights to fp16 the same way: train.py:54, 273)."""
    return {k: torch.tensor(v).to(dtype).to(torch.float64).numpy() if v.ndim >= 2 and "norm" not in k
            and "bias" not in k and "gamma" not in k and "depth_conv" not in k and k != "layers.0.0.conv.weight"
            else v for k, v in params.items()}


def _chan_ln(x, w, b):
    """LayerNorm over channels of (B, C, L)  (model.py:100,117,162,759: vmap over positions)."""
    return F.layer_norm(x.transpose(1, 2), (x.shape[1],), w, b, LN_EPS).transpose(1, 2)


def _rope(x, cos, sin):
    """rope.py:25-53.  x (B, S, nh, hd), cos/sin (S, hd/2)."""
    x1, x2 = x[..., 0::2], x[..., 1::2]
    c, s = cos[None, :, None, :], sin[None, :, None, :]
    return torch.stack([x1 * c - x2 * s, x1 * s + x2 * c], dim=-1).flatten(-2)


def _no_tap(name, t):
    return t


def _sdpa(q, k, v, keep=None, keep_prob=1.0, tap=_no_tap, tag=""):
    """model.py:241-257.  q,k,v (..., S, hd)."""
    w = torch.softmax((q / math.sqrt(q.shape[-1])) @ k.transpose(-1, -2), dim=-1)
    if keep is not None:
        w = w * keep / keep_prob
    return tap(tag + "p", w) @ v


def _self_attention(lp, x, cos, sin, nh, keep=None, keep_prob=1.0, tap=_no_tap, tag=""):
    """model.py:340-369.  x (B, S, H) -> (B, S, H).  `tap(name, tensor) -> tensor` (default: identity) sees every tensor
    an implementation would store between operators (tests: intermediates, rounding-point studies)."""
    B, S, _ = x.shape
    q = tap(tag + "q", _rope(tap(tag + "q_raw", F.linear(x, lp["Wq"])).view(B, S, nh, -1), cos[:S], sin[:S]))
    c = tap(tag + "c", F.linear(x, lp["Wkvd"]))
    k = tap(tag + "k", _rope(tap(tag + "k_raw", F.linear(c, lp["Wku"])).view(B, S, nh, -1), cos[:S], sin[:S]))
    v = tap(tag + "v", F.linear(c, lp["Wvu"])).view(B, S, nh, -1)
    o = _sdpa(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), keep, keep_prob, tap, tag)
    o = tap(tag + "ao", o.transpose(1, 2).reshape(B, S, -1))
    return F.linear(o, lp["Wo"])


def _local_attention(lp, x, cos, sin, nh, keep=None, keep_prob=1.0, tap=_no_tap, tag=""):
    """model.py:409-471.  keep: optional (B, nwin, nh, W, W)."""
    B, S, H = x.shape
    W = LOCAL_CONTEXT_WINDOW
    pl, pr, nwin, stride = local_window_geometry(S, W)
    xp = F.pad(x, (0, 0, pl, pr))
    win = xp.unfold(1, W, stride).permute(0, 1, 3, 2)          # (B, nwin, W, H)
    kp = None if keep is None else keep.reshape(B * nwin, nh, W, W)
    ow = _self_attention(lp, win.reshape(B * nwin, W, H), cos, sin, nh, kp, keep_prob, tap, tag)
    ow = ow.view(B, nwin, W, H)
    idx = (torch.arange(nwin)[:, None] * stride + torch.arange(W)[None, :]).reshape(-1)
    valid = idx < S                                           # OOB scatter updates dropped
    out = torch.zeros(B, S, H, dtype=x.dtype)
    out.index_add_(1, idx[valid], ow.reshape(B, nwin * W, H)[:, valid])
    cnt = torch.zeros(S, dtype=x.dtype)
    cnt.index_add_(0, idx[valid], torch.ones(int(valid.sum()), dtype=x.dtype))
    return out / cnt[None, :, None]


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


def cnn_frontend(p, x, conf=MODEL_CONFIG, tap=_no_tap):
    """model.py:755-762.  x (B, 2, T) -> (B, S, H).  Tap names: "cnn.{s}.0.y" (Stem / Downsample output), "cnn.{s}.0.ln"
    (the LayerNorm a Downsample opens with), "cnn.{s}.{j}.{ln,u,h,z,y}" per Block, "cnn.norm" (final LayerNorm)."""
    dims, depths = conf["dims"], conf["depths"]
    h = x
    for s in range(len(dims)):
        pre = f"layers.{s}.0."
        if s == 0:
            h = F.conv1d(h, p[pre + "conv.weight"], p[pre + "conv.bias"][:, 0], stride=5)
            h = _chan_ln(h, p[pre + "norm.weight"], p[pre + "norm.bias"])
        else:
            h = tap(f"cnn.{s}.0.ln", _chan_ln(h, p[pre + "norm.weight"], p[pre + "norm.bias"]))
            h = F.conv1d(h, p[pre + "conv.weight"], p[pre + "conv.bias"][:, 0], stride=2)
        h = tap(f"cnn.{s}.0.y", h)
        C = dims[s]
        for j in range(1, depths[s] + 1):
            q = f"layers.{s}.{j}."
            t = f"cnn.{s}.{j}."
            o = F.conv1d(h, p[q + "depth_conv.weight"], p[q + "depth_conv.bias"][:, 0], padding=3, groups=C)
            o = tap(t + "ln", _chan_ln(o, p[q + "norm.weight"], p[q + "norm.bias"]))
            o = tap(t + "u", F.conv1d(o, p[q + "point_conv_1.weight"], p[q + "point_conv_1.bias"][:, 0]))
            o = tap(t + "h", F.gelu(o, approximate="tanh"))
            o = tap(t + "z", F.conv1d(o, p[q + "point_conv_2.weight"], p[q + "point_conv_2.bias"][:, 0]))
            h = tap(t + "y", p[q + "gamma"][None, :, None] * o + h)
    h = tap("cnn.norm", _chan_ln(h, p["norm.weight"], p["norm.bias"]))
    return h.transpose(1, 2)


def forward(p, audio, conf=MODEL_CONFIG, rope=None, dropout_masks=None, return_intermediates=False, tap=_no_tap):
    """Batched OutputSequenceGenerator (model.py:740-769 under vmap).  audio (B,2,T).  Tap names of layer l: "tl.{l}.{xn,
    q_raw,q,c,k_raw,k,v,p,ao,h,hn,u,g,y}" (p: attention weights after dropout; local layers: q .. ao are per window); "head.hn"."""
    dtype = audio.dtype
    h = cnn_frontend(p, audio, conf, tap)
    inter = {"cnn": h}
    B, S, H = h.shape
    nh, hd = conf["num_transformer_heads"], conf["attention_size"]
    if rope is None:
        rope = precompute_frequencies(hd, max(S, 300))
    cos = torch.as_tensor(rope[0]).to(dtype)
    sin = torch.as_tensor(rope[1]).to(dtype)
    li = 0
    for i in range(conf["num_transformer_layers"]):
        for kind in ("local_attention", "global_attention"):
            lp = _layer_params(p, kind, i)
            dm = None if dropout_masks is None else dropout_masks[li]
            kp = 1.0 if dm is None else dm["keep"]
            am = None if dm is None else dm["attn"]
            tg = f"tl.{li}."
            xn = tap(tg + "xn", F.layer_norm(h, (H,), lp["ln1w"], lp["ln1b"], LN_EPS))
            if kind == "local_attention":
                r = _local_attention(lp, xn, cos, sin, nh, am, kp, tap, tg)
            else:
                r = _self_attention(lp, xn, cos, sin, nh, am, kp, tap, tg)
            h = tap(tg + "h", h + r)
            hn = tap(tg + "hn", F.layer_norm(h, (H,), lp["ln2w"], lp["ln2b"], LN_EPS))
            y = tap(tg + "u", F.linear(hn, lp["W1"], lp["b1"]))
            y1, y2 = y.chunk(2, dim=-1)
            r = F.linear(tap(tg + "g", F.gelu(y1, approximate="tanh") * y2), lp["W2"], lp["b2"])
            if dm is not None:
                r = r * dm["ffn"] / kp
            h = tap(tg + "y", h + r)
            inter[f"layer{li}"] = h
            li += 1
    hn = tap("head.hn", F.layer_norm(h, (H,), p["decoder.norm.weight"], p["decoder.norm.bias"], LN_EPS))
    logits = F.linear(hn, p["decoder.decoder_pooling.weight"], p["decoder.decoder_pooling.bias"])
    probs = torch.sigmoid(logits)
    if return_intermediates:
        return logits, probs, inter
    return logits, probs


def loss_fn(logits, labels, scale=1.0):
    """train.py:39-62: BCE-with-logits summed over (frames, keys) * scale, mean over batch."""
    B = logits.shape[0]
    per = F.binary_cross_entropy_with_logits(logits.float() if logits.dtype == torch.float16 else logits,
                                             labels.to(logits.dtype), reduction="none")
    return (per * scale).reshape(B, -1).sum(dim=1).mean()


def loss_and_grads(params_np, audio, labels, conf=MODEL_CONFIG, scale=1.0, rope=None,
                   dtype=torch.float64, dropout_masks=None):
    """value_and_grad of compute_loss (train.py:49-62) w.r.t. every parameter leaf.

    Returns (loss float, logits ndarray (B,S,90), grads dict name -> ndarray)."""
    p = to_torch(params_np, dtype, requires_grad=True)
    a = torch.as_tensor(audio).to(dtype)
    y = torch.as_tensor(labels).to(dtype)
    logits, _ = forward(p, a, conf, rope, dropout_masks)
    loss = loss_fn(logits, y, scale)
    loss.backward()
    grads = {k: (v.grad.detach().numpy().copy() if v.grad is not None else None) for k, v in p.items()}
    return float(loss.detach()), logits.detach().numpy().copy(), grads
