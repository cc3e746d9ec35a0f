"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, the
flat layout restated in Python equals the library's, and the Equinox <-> device mapping is a
bijection.  (No compute calls: there is no GPU here.)"""
import os
import re

import numpy as np
import pytest

import audio_to_midi_amd as amt
from audio_to_midi_amd import lib, params
from oracle import np_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = lib.load()
    declared = set()
    for h in ("amt.h", "amt_ops.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        declared |= set(re.findall(r"\b(amt_[a-z0-9_]+)\s*\(", src))
    assert declared, "no declarations found"
    for name in sorted(declared):
        assert hasattr(L, name), f"libamt_hip.so does not export {name}"
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)


def test_layout_python_equals_library(small_conf):
    for conf in (amt.model_config, small_conf):
        a, b = params.device_layout(conf), params.library_layout(conf)
        assert a == b
        cfg = lib.make_config(conf, "bfloat16")
        assert lib.load().amt_param_count(cfg) == params.param_count(conf)
    assert params.param_count(amt.model_config) == 11_606_230


def test_lr_depth_labels_follow_reference():
    """train.py:646-672: depth = sum(depths[:stage]) + index in the Sequential; 40 conv depths"""
    segs = params.device_layout(amt.model_config)
    depth = {s.name: s.lr_depth for s in segs}
    assert depth["layers.0.0.conv.weight"] == 0 and depth["layers.0.3.gamma"] == 3
    assert depth["layers.1.0.conv.weight"] == 3 and depth["layers.5.21.gamma"] == 36
    assert depth["layers.6.0.norm.weight"] == 36 and depth["layers.6.3.gamma"] == 39
    assert depth["tl.0.wqc"] == -1 and depth["decoder.norm.bias"] == -1
    assert max(depth.values()) == 39


def test_equinox_device_mapping_is_a_bijection(small_conf):
    conf = small_conf
    p = np_model.init_params(conf, 9, gamma=0.5)
    flat = params.equinox_to_flat(p, conf)
    assert flat.dtype == np.float32 and flat.size == params.param_count(conf)
    back = params.flat_to_equinox(flat, conf)
    assert list(back) == list(params.equinox_shapes(conf)) == list(np_model.param_shapes(conf))
    for k, v in p.items():
        assert back[k].shape == v.shape and np.array_equal(back[k], v.astype(np.float32)), k
    # every flat element is hit exactly once
    marks = params.flat_to_equinox(np.arange(flat.size, dtype=np.float64), conf)
    allv = np.concatenate([v.reshape(-1) for v in marks.values()])
    assert np.array_equal(np.sort(allv), np.arange(flat.size))


def test_glu_permutation_pairs_gate_rows():
    perm = params.glu_row_perm(512)
    assert sorted(perm) == list(range(1024))
    for r in range(0, 1024, 32):
        assert np.array_equal(perm[r + 16: r + 32], perm[r: r + 16] + 512)


def test_rope_table_matches_oracle():
    a = amt.precompute_frequencies(64, 512)
    c, s = np_model.precompute_frequencies(64, 512)
    assert np.array_equal(a.cos_freq, c) and np.array_equal(a.sin_freq, s)


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        amt.OutputSequenceGenerator(amt.model_config)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "audio-to-midi_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("the oracle", "").replace("oracle consumes", ""), (f, "mentions oracle")


def test_synthetic_inputs_follow_the_data_contract():
    a, y = amt.synth.make_batch(2, 16000, seed=1)
    assert a.shape == (2, 2, 16000) and y.shape == (2, 50, 90)
    assert np.allclose((a.astype(np.float64) ** 2).mean(axis=(1, 2)), 1.0, atol=2e-3)
    assert np.array_equal(a, a.astype(np.float16).astype(np.float32))
    assert y.min() >= 0.005 and y.max() <= 0.995 and np.all(y[:, :, 88:] == np.float32(0.005))


def test_checkpoint_leaves_are_validated_and_the_ensemble_axis_is_handled(tmp_path):
    """reference checkpoints carry a leading ensemble axis on every leaf (infer.py:213-224) and a
    wrong-layout leaf of equal size must not be accepted silently."""
    import numpy as np
    from audio_to_midi_amd import convert_orbax, infer
    from audio_to_midi_amd import params as P
    from oracle import np_model
    conf = dict(np_model.MODEL_CONFIG)
    conf["depths"] = [1, 1, 1, 1, 1, 2, 1]
    conf["num_transformer_layers"] = 1
    p = P.init_equinox(conf, seed=5)
    flat = P.equinox_to_flat(p, conf)
    # ensemble axis: member k is selected, plain leaves pass through, a missing selection is an error
    ens = {k: np.stack([v, 3 * v]) for k, v in p.items()}
    assert np.array_equal(P.equinox_to_flat(ens, conf, ensemble_select=0), flat)
    assert np.array_equal(P.equinox_to_flat(ens, conf, ensemble_select=1), 3 * flat)
    with pytest.raises(ValueError, match="ensemble"):
        P.equinox_to_flat(ens, conf)
    with pytest.raises(IndexError):
        P.equinox_to_flat(ens, conf, ensemble_select=2)
    # a transposed weight has the right SIZE and the wrong layout
    bad = dict(p)
    k = "layers.5.1.point_conv_1.weight"
    bad[k] = np.transpose(bad[k], (1, 0, 2))
    with pytest.raises(ValueError, match="point_conv_1"):
        P.equinox_to_flat(bad, conf)
    missing = dict(p); missing.pop("norm.bias")
    with pytest.raises(KeyError):
        P.equinox_to_flat(missing, conf)
    extra = dict(p); extra["layers.9.0.conv.weight"] = np.zeros(3)
    with pytest.raises(KeyError):
        P.equinox_to_flat(extra, conf)
    # the converter's pytree flattening: nested dicts / sequences as Orbax restores them, None leaves dropped
    tree = {}
    for name, v in ens.items():
        node = tree
        parts = name.split(".")
        for q in parts[:-1]:
            node = node.setdefault(q, {})
        node[parts[-1]] = v
    tree["layers"]["0"]["0"]["norm"]["eps"] = None
    tree["layers"] = [tree["layers"][str(i)] for i in range(len(tree["layers"]))]      # a sequence node
    leaves = convert_orbax.flatten_tree(tree)
    assert set(leaves) == set(p)
    assert np.array_equal(convert_orbax.leaves_to_flat(leaves, conf, ensemble_select=1), 3 * flat)
    back = convert_orbax.flat_to_leaves(flat, conf)
    assert all(back[k].shape == (1,) + p[k].shape and np.array_equal(back[k][0], p[k]) for k in p)
    # the reference's REAL path structure.  OutputSequenceGenerator.layers is a list of eqx.nn.Sequential
    # (model.py:674, 704-710) whose modules sit in an inner `.layers` tuple: layers[s].layers[j].point_conv_1.weight
    # (train.py:650-653 reads path[1].idx and path[3].idx).  A tree built with that level must convert to the same
    # flat vector, the name mapping must be a bijection, and an unknown model leaf must raise instead of being dropped.
    ref_tree = {"layers": []}
    for s in range(len(conf["dims"])):
        mods = []
        for j in range(conf["depths"][s] + 1):
            node = {}
            for name, v in ens.items():
                pre = f"layers.{s}.{j}."
                if name.startswith(pre):
                    cur = node
                    parts = name[len(pre):].split(".")
                    for q in parts[:-1]:
                        cur = cur.setdefault(q, {})
                    cur[parts[-1]] = v
            mods.append(node)
        ref_tree["layers"].append({"layers": tuple(mods)})
    for name, v in ens.items():
        if not name.startswith("layers."):
            cur = ref_tree
            parts = name.split(".")
            for q in parts[:-1]:
                cur = cur.setdefault(q, {})
            cur[parts[-1]] = v
    ref_leaves = convert_orbax.flatten_tree(ref_tree)
    assert "layers.5.layers.1.point_conv_1.weight" in ref_leaves and "layers.5.1.point_conv_1.weight" not in ref_leaves
    assert {convert_orbax.canonical_name(k) for k in ref_leaves} == set(p)
    assert all(convert_orbax.reference_name(convert_orbax.canonical_name(k)) == k for k in ref_leaves)
    assert np.array_equal(convert_orbax.leaves_to_flat(ref_leaves, conf, ensemble_select=1), 3 * flat)
    stray = dict(ref_leaves); stray["layers.5.layers.1.point_conv_3.weight"] = np.zeros((2, 1))
    with pytest.raises(KeyError, match="parameter inventory"):
        convert_orbax.leaves_to_flat(stray, conf, ensemble_select=0)
    ok_extra = dict(ref_leaves); ok_extra["state.unused"] = np.zeros(1)          # outside the model: ignored
    assert np.array_equal(convert_orbax.leaves_to_flat(ok_extra, conf, ensemble_select=0), flat)
    # flat file round trip
    infer.save_checkpoint(str(tmp_path), flat, conf, step=7)
    f2, c2, s2 = infer.load_checkpoint(str(tmp_path))
    assert np.array_equal(f2, flat) and s2 == 7 and c2["depths"] == conf["depths"]
