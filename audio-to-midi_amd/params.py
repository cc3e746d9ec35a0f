"""Parameter inventory and the flat device layout.

Two views of the same 11 606 230 parameters:

* **Equinox view** — a dict keyed by the reference's pytree paths with Equinox shapes
  (model.py:84-198, 260-335, 474-527, 615-647, 673-738): what a checkpoint of the reference
  contains, and what the oracle consumes.
* **Device view** — ONE flat fp32 vector laid out for channels-last kernels (the C library's
  ``amt_layout_segment`` is authoritative; ``device_layout`` below restates it in Python and
  tests/ check that the two agree).  Differences from the Equinox view, all of them bijections:
    - ``depth_conv.weight``  (C,1,7)      -> [7][C]
    - downsample ``conv.weight`` (Co,Ci,2)-> [Co][k*Ci + ci]   (a k2/s2 conv over channels-last
                                             is a Linear over two concatenated positions)
    - conv biases (C,1) -> [C];  1x1 conv weights (O,I,1) -> [O][I]
    - per transformer layer: ``wqc`` = [query_up_proj ; kv_down_proj] rows, ``wkv`` =
      [key_up_proj ; value_up_proj] rows, and the GLU projection ``w1``/``b1`` with rows
      reordered in blocks of (16 x u1 | 16 x u2) so the GLU pairs up inside one MFMA lane.
    - stacked transformer leaves (leading dim 8, local/global) -> per-layer segments ``tl.<i>``
      with i = 2*pair + (0 local | 1 global)  (model.py:609-611 applies local first).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, NamedTuple

import numpy as np

N_EVENTS = 90  # MIDI_EVENT_VOCCAB_SIZE, audio_to_midi_dataset.py:26


class Segment(NamedTuple):
    name: str
    offset: int
    rows: int
    cols: int
    lr_depth: int

    @property
    def size(self):
        return self.rows * self.cols


def equinox_shapes(conf) -> "OrderedDict[str, tuple]":
    """Leaf name -> Equinox shape, in pytree order (Table P of SURVEY.md)."""
    dims, depths = conf["dims"], conf["depths"]
    H = dims[-1]
    nl = conf["num_transformer_layers"]
    hd_all = conf["num_transformer_heads"] * conf["attention_size"]
    ckv = conf["compressed_attention_kv_size"]
    inter = int(H * conf["transformer_hidden_expansion"])
    sh: "OrderedDict[str, tuple]" = OrderedDict()
    for s, C in enumerate(dims):
        p = f"layers.{s}.0."
        cin = 2 if s == 0 else dims[s - 1]
        sh[p + "conv.weight"] = (C, cin, 5 if s == 0 else 2)
        sh[p + "conv.bias"] = (C, 1)
        sh[p + "norm.weight"] = (C if s == 0 else cin,)
        sh[p + "norm.bias"] = (C if s == 0 else cin,)
        for j in range(1, depths[s] + 1):
            q = f"layers.{s}.{j}."
            sh[q + "depth_conv.weight"] = (C, 1, 7)
            sh[q + "depth_conv.bias"] = (C, 1)
            sh[q + "norm.weight"] = (C,)
            sh[q + "norm.bias"] = (C,)
            sh[q + "point_conv_1.weight"] = (2 * C, C, 1)
            sh[q + "point_conv_1.bias"] = (2 * C, 1)
            sh[q + "point_conv_2.weight"] = (C, 2 * C, 1)
            sh[q + "point_conv_2.bias"] = (C, 1)
            sh[q + "gamma"] = (C,)
    sh["norm.weight"] = (H,)
    sh["norm.bias"] = (H,)
    for kind in ("local_attention", "global_attention"):
        t, a, f = _tnames(kind)
        sh[t + "attention_norm.weight"] = (nl, H)
        sh[t + "attention_norm.bias"] = (nl, H)
        sh[a + "query_up_proj.weight"] = (nl, hd_all, H)
        sh[a + "kv_down_proj.weight"] = (nl, ckv, H)
        sh[a + "key_up_proj.weight"] = (nl, hd_all, ckv)
        sh[a + "value_up_proj.weight"] = (nl, hd_all, ckv)
        sh[a + "output_proj.weight"] = (nl, H, hd_all)
        sh[t + "feed_forward_norm.weight"] = (nl, H)
        sh[t + "feed_forward_norm.bias"] = (nl, H)
        sh[f + "attention_to_intermediate_proj.weight"] = (nl, 2 * inter, H)
        sh[f + "attention_to_intermediate_proj.bias"] = (nl, 2 * inter)
        sh[f + "intermediate_to_attention_proj.weight"] = (nl, H, inter)
        sh[f + "intermediate_to_attention_proj.bias"] = (nl, H)
    sh["decoder.norm.weight"] = (H,)
    sh["decoder.norm.bias"] = (H,)
    sh["decoder.decoder_pooling.weight"] = (conf.get("n_events", N_EVENTS), H)
    sh["decoder.decoder_pooling.bias"] = (conf.get("n_events", N_EVENTS),)
    return sh


def _tnames(kind):
    t = f"transformer.layers.{kind}."
    a = t + ("attention_block.self_attention." if kind == "local_attention" else "attention_block.")
    f = t + "feed_forward_block."
    return t, a, f


def glu_row_perm(inter: int) -> np.ndarray:
    """device row r of w1/b1 -> Equinox row.  Blocks of 32 = 16 'gelu' rows then their 16 gates."""
    r = np.arange(2 * inter)
    blk, w = r // 32, r % 32
    return np.where(w < 16, 16 * blk + w, inter + 16 * blk + (w - 16))


def device_layout(conf) -> List[Segment]:
    """Python restatement of the library's flat layout (amt_layout_segment)."""
    dims, depths = conf["dims"], conf["depths"]
    H = dims[-1]
    hd_all = conf["num_transformer_heads"] * conf["attention_size"]
    ckv = conf["compressed_attention_kv_size"]
    inter = int(H * conf["transformer_hidden_expansion"])
    V = conf.get("n_events", N_EVENTS)
    segs: List[Segment] = []
    off = 0

    def seg(name, rows, cols, depth):
        nonlocal off
        segs.append(Segment(name, off, rows, cols, depth))
        off += rows * cols

    base = 0
    for s, C in enumerate(dims):
        p = f"layers.{s}.0."
        if s == 0:
            seg(p + "conv.weight", C, 10, base)
            seg(p + "conv.bias", C, 1, base)
            seg(p + "norm.weight", C, 1, base)
            seg(p + "norm.bias", C, 1, base)
        else:
            cin = dims[s - 1]
            seg(p + "norm.weight", cin, 1, base)
            seg(p + "norm.bias", cin, 1, base)
            seg(p + "conv.weight", C, 2 * cin, base)
            seg(p + "conv.bias", C, 1, base)
        for j in range(1, depths[s] + 1):
            q = f"layers.{s}.{j}."
            d = base + j
            seg(q + "depth_conv.weight", 7, C, d)
            seg(q + "depth_conv.bias", C, 1, d)
            seg(q + "norm.weight", C, 1, d)
            seg(q + "norm.bias", C, 1, d)
            seg(q + "point_conv_1.weight", 2 * C, C, d)
            seg(q + "point_conv_1.bias", 2 * C, 1, d)
            seg(q + "point_conv_2.weight", C, 2 * C, d)
            seg(q + "point_conv_2.bias", C, 1, d)
            seg(q + "gamma", C, 1, d)
        base += depths[s]
    seg("norm.weight", H, 1, -1)
    seg("norm.bias", H, 1, -1)
    for li in range(2 * conf["num_transformer_layers"]):
        t = f"tl.{li}."
        seg(t + "ln1.weight", H, 1, -1)
        seg(t + "ln1.bias", H, 1, -1)
        seg(t + "wqc", hd_all + ckv, H, -1)
        seg(t + "wkv", 2 * hd_all, ckv, -1)
        seg(t + "wo", H, hd_all, -1)
        seg(t + "ln2.weight", H, 1, -1)
        seg(t + "ln2.bias", H, 1, -1)
        seg(t + "w1", 2 * inter, H, -1)
        seg(t + "b1", 2 * inter, 1, -1)
        seg(t + "w2", H, inter, -1)
        seg(t + "b2", H, 1, -1)
    seg("decoder.norm.weight", H, 1, -1)
    seg("decoder.norm.bias", H, 1, -1)
    seg("decoder.decoder_pooling.weight", V, H, -1)
    seg("decoder.decoder_pooling.bias", V, 1, -1)
    return segs


def library_layout(conf) -> List[Segment]:
    """The authoritative layout, queried from libamt_hip.so (no GPU needed)."""
    from . import lib as _lib
    L = _lib.load()
    cfg = _lib.make_config(conf, "float32")
    n = L.amt_layout_count(cfg)
    if n < 0:
        raise ValueError("configuration rejected by the library")
    out = []
    sg = _lib.AmtSegment()
    for i in range(n):
        if L.amt_layout_segment(cfg, i, sg) != 0:
            raise RuntimeError("amt_layout_segment failed")
        out.append(Segment(sg.name.decode(), int(sg.offset), int(sg.rows), int(sg.cols), int(sg.lr_depth)))
    return out


def param_count(conf) -> int:
    return sum(s.size for s in device_layout(conf))


def _layer_source(conf, li):
    kind = "local_attention" if li % 2 == 0 else "global_attention"
    return li // 2, _tnames(kind)


def validate_equinox(p: Dict[str, np.ndarray], conf, ensemble_select=None) -> "OrderedDict[str, np.ndarray]":
    """Check a dict of reference leaves against ``equinox_shapes(conf)`` and return it with exactly those shapes.

    The reference trains an ENSEMBLE: every array leaf of its checkpoints carries a leading ensemble axis
    (train.py:259-264 vmaps the step over it; infer.py:213-224 selects ``x[ensemble_select]`` after restoring).
    ``ensemble_select=None`` expects leaves without that axis; an integer selects that member from leaves that have
    it (leaves that already have the plain shape are accepted as they are).  Anything else raises: a missing or
    unexpected key, a leaf whose shape is neither the expected one nor (E,) + expected - so a transposed weight of
    equal size, which a bare ``reshape(-1)`` would take silently, is an error here."""
    shapes = equinox_shapes(conf)
    missing = [k for k in shapes if k not in p]
    extra = [k for k in p if k not in shapes]
    if missing or extra:
        raise KeyError(f"parameter dict does not match the model: missing {missing[:4]}{'...' if len(missing) > 4 else ''}, "
                       f"unexpected {extra[:4]}{'...' if len(extra) > 4 else ''}")
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for k, shp in shapes.items():
        v = np.asarray(p[k])
        if tuple(v.shape) == tuple(shp):
            out[k] = v
        elif ensemble_select is not None and v.ndim == len(shp) + 1 and tuple(v.shape[1:]) == tuple(shp):
            if not (-v.shape[0] <= ensemble_select < v.shape[0]):
                raise IndexError(f"{k}: ensemble_select={ensemble_select} but the leaf holds {v.shape[0]} members")
            out[k] = v[ensemble_select]
        else:
            hint = " (leading ensemble axis? pass ensemble_select)" if v.ndim == len(shp) + 1 and tuple(v.shape[1:]) == tuple(shp) else ""
            raise ValueError(f"{k}: shape {tuple(v.shape)}, expected {tuple(shp)}{hint}")
    return out


def equinox_to_flat(p: Dict[str, np.ndarray], conf, ensemble_select=None) -> np.ndarray:
    """Equinox-view dict -> flat fp32 vector in device layout.  Every leaf is shape-checked first
    (``validate_equinox``); ``ensemble_select`` strips the reference checkpoints' leading ensemble axis."""
    p = validate_equinox(p, conf, ensemble_select)
    inter = int(conf["dims"][-1] * conf["transformer_hidden_expansion"])
    perm = glu_row_perm(inter)
    segs = device_layout(conf)
    flat = np.zeros(sum(s.size for s in segs), dtype=np.float32)
    for s in segs:
        n = s.name
        if n.startswith("tl."):
            li = int(n.split(".")[1])
            leaf = n.split(".", 2)[2]
            i, (t, a, f) = _layer_source(conf, li)
            if leaf == "ln1.weight": v = p[t + "attention_norm.weight"][i]
            elif leaf == "ln1.bias": v = p[t + "attention_norm.bias"][i]
            elif leaf == "wqc": v = np.concatenate([p[a + "query_up_proj.weight"][i], p[a + "kv_down_proj.weight"][i]], 0)
            elif leaf == "wkv": v = np.concatenate([p[a + "key_up_proj.weight"][i], p[a + "value_up_proj.weight"][i]], 0)
            elif leaf == "wo": v = p[a + "output_proj.weight"][i]
            elif leaf == "ln2.weight": v = p[t + "feed_forward_norm.weight"][i]
            elif leaf == "ln2.bias": v = p[t + "feed_forward_norm.bias"][i]
            elif leaf == "w1": v = p[f + "attention_to_intermediate_proj.weight"][i][perm]
            elif leaf == "b1": v = p[f + "attention_to_intermediate_proj.bias"][i][perm]
            elif leaf == "w2": v = p[f + "intermediate_to_attention_proj.weight"][i]
            elif leaf == "b2": v = p[f + "intermediate_to_attention_proj.bias"][i]
            else: raise KeyError(n)
        else:
            v = np.asarray(p[n])
            if n.endswith("depth_conv.weight"):
                v = v[:, 0, :].T                                   # (C,1,7) -> [7][C]
            elif n.endswith(".0.conv.weight") and not n.startswith("layers.0."):
                v = np.transpose(v, (0, 2, 1)).reshape(v.shape[0], -1)  # (Co,Ci,2) -> [Co][k*Ci+ci]
        flat[s.offset: s.offset + s.size] = np.asarray(v, dtype=np.float32).reshape(-1)
    return flat


def flat_to_equinox(flat: np.ndarray, conf) -> "OrderedDict[str, np.ndarray]":
    """Inverse of equinox_to_flat (also used to present gradients in the reference's pytree)."""
    flat = np.asarray(flat)
    shapes = equinox_shapes(conf)
    out = OrderedDict((k, np.zeros(v, dtype=flat.dtype)) for k, v in shapes.items())
    inter = int(conf["dims"][-1] * conf["transformer_hidden_expansion"])
    perm = glu_row_perm(inter)
    hd_all = conf["num_transformer_heads"] * conf["attention_size"]
    for s in device_layout(conf):
        n = s.name
        v = flat[s.offset: s.offset + s.size].reshape(s.rows, s.cols)
        if n.startswith("tl."):
            li = int(n.split(".")[1])
            leaf = n.split(".", 2)[2]
            i, (t, a, f) = _layer_source(conf, li)
            if leaf == "ln1.weight": out[t + "attention_norm.weight"][i] = v[:, 0]
            elif leaf == "ln1.bias": out[t + "attention_norm.bias"][i] = v[:, 0]
            elif leaf == "wqc":
                out[a + "query_up_proj.weight"][i] = v[:hd_all]
                out[a + "kv_down_proj.weight"][i] = v[hd_all:]
            elif leaf == "wkv":
                out[a + "key_up_proj.weight"][i] = v[:hd_all]
                out[a + "value_up_proj.weight"][i] = v[hd_all:]
            elif leaf == "wo": out[a + "output_proj.weight"][i] = v
            elif leaf == "ln2.weight": out[t + "feed_forward_norm.weight"][i] = v[:, 0]
            elif leaf == "ln2.bias": out[t + "feed_forward_norm.bias"][i] = v[:, 0]
            elif leaf == "w1": out[f + "attention_to_intermediate_proj.weight"][i][perm] = v
            elif leaf == "b1": out[f + "attention_to_intermediate_proj.bias"][i][perm] = v[:, 0]
            elif leaf == "w2": out[f + "intermediate_to_attention_proj.weight"][i] = v
            elif leaf == "b2": out[f + "intermediate_to_attention_proj.bias"][i] = v[:, 0]
            else: raise KeyError(n)
        else:
            shp = shapes[n]
            if n.endswith("depth_conv.weight"):
                out[n][:, 0, :] = v.T
            elif n.endswith(".0.conv.weight") and not n.startswith("layers.0."):
                co, ci, k = shp
                out[n][...] = np.transpose(v.reshape(co, k, ci), (0, 2, 1))
            else:
                out[n][...] = v.reshape(shp)
    return out


def init_equinox(conf, seed: int = 1234, gamma: float = 1e-6) -> "OrderedDict[str, np.ndarray]":
    """Equinox default init: Conv/Linear weight and bias ~ U(+-1/sqrt(fan_in)) with
    fan_in = (in/groups)*k; LayerNorm weight 1, bias 0; layer scale gamma (model.py:157-158).
    (numpy RNG; JAX's threefry stream is not reproducible offline.)"""
    rng = np.random.default_rng(seed)
    shapes = equinox_shapes(conf)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shp in shapes.items():
        leaf = name.rsplit(".", 1)[-1]
        stacked = name.startswith("transformer.")
        if name.endswith("norm.weight"):
            out[name] = np.ones(shp, dtype=np.float32)
        elif name.endswith("norm.bias"):
            out[name] = np.zeros(shp, dtype=np.float32)
        elif leaf == "gamma":
            out[name] = np.full(shp, gamma, dtype=np.float32)
        else:
            wshape = shp if leaf == "weight" else shapes[name[: -len("bias")] + "weight"]
            core = wshape[1:] if stacked else wshape
            lim = 1.0 / math.sqrt(int(np.prod(core[1:])))
            out[name] = rng.uniform(-lim, lim, size=shp).astype(np.float32)
    return out
