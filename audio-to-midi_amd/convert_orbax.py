"""Checkpoint converter between the reference's Orbax checkpoints and this library's flat weight files.

Runs WHERE THE REFERENCE RUNS (needs ``orbax-checkpoint``, and for the reverse direction ``jax`` + ``equinox`` and
the reference checkout on PYTHONPATH); none of that exists in the offline build container, so the Orbax calls below
are written against the public API (``CheckpointManager(path, item_names=('params', 'state'))``, ``StandardRestore`` /
``StandardSave``; reference: train.py:384-394, 799-831 and infer.py:172-224) and are NOT exercised by the test suite.
What IS tested offline (tests/test_host_cpu.py) is everything between the restored pytree and the weight file: the
flattening of the nested pytree into ``'layers.5.3.point_conv_1.weight'``-style names, the shape validation against
Table P, the ensemble axis, and the bijection to the device layout.

The reference trains an ensemble: every array leaf has a leading ensemble axis of size E (infer.py:213-224 picks
``x[ensemble_select]``).  ``to-amt`` strips it (``--ensemble-select``, default 0); ``from-amt`` adds it back (E = 1).

  python -m audio_to_midi_amd.convert_orbax to-amt   <orbax_dir> <amt_dir> [--step N] [--ensemble-select K]
  python -m audio_to_midi_amd.convert_orbax from-amt <amt_dir> <orbax_dir> --reference /path/to/audio-to-midi [--step N]
  python -m audio_to_midi_amd.convert_orbax npz-to-amt <leaves.npz> <amt_dir> [--ensemble-select K]   (no Orbax needed)
"""
from __future__ import annotations

import argparse
import os
import re
import sys
from collections import OrderedDict
from typing import Any, Dict

import numpy as np

from . import params as _params
from .infer import load_checkpoint, save_checkpoint
from .model import model_config


def flatten_tree(tree: Any, prefix: str = "") -> "OrderedDict[str, np.ndarray]":
    """Nested pytree as Orbax restores it without a target (dicts keyed by attribute name, lists / tuples or dicts
    keyed by '0', '1', ... for sequences, None for non-array leaves) -> {'a.b.0.c': array}.  None leaves are dropped
    (eqx.filter(model, eqx.is_inexact_array) saves them as None, train.py:387)."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    if tree is None:
        return out
    if isinstance(tree, dict):
        items = tree.items()
    elif isinstance(tree, (list, tuple)):
        items = ((str(i), v) for i, v in enumerate(tree))
    else:
        out[prefix[:-1]] = np.asarray(tree)
        return out
    for k, v in items:
        out.update(flatten_tree(v, f"{prefix}{k}."))
    return out


_SEQ_LEVEL = re.compile(r"^layers\.(\d+)\.layers\.(\d+)\.")


def canonical_name(name: str) -> str:
    """Reference pytree path -> the leaf name used by ``params.equinox_shapes``.

    ``OutputSequenceGenerator.layers`` is a list of ``eqx.nn.Sequential`` (model.py:674, 704-710) and a Sequential
    keeps its modules in an inner ``.layers`` tuple, so the real path of a CNN leaf is
    ``layers.<s>.layers.<j>.point_conv_1.weight`` (train.py:650-653 reads ``path[1].idx`` and ``path[3].idx`` for
    exactly that reason); this package names it ``layers.<s>.<j>.point_conv_1.weight``."""
    return _SEQ_LEVEL.sub(r"layers.\1.\2.", name)


def reference_name(name: str) -> str:
    """Inverse of ``canonical_name``: put the Sequential's ``.layers`` level back."""
    m = re.match(r"^layers\.(\d+)\.(\d+)\.", name)
    return name if m is None else f"layers.{m.group(1)}.layers.{m.group(2)}." + name[m.end():]


def leaves_to_flat(leaves: Dict[str, np.ndarray], conf: dict, ensemble_select=0) -> np.ndarray:
    """Reference leaves (with or without the ensemble axis, reference or canonical names) -> flat fp32 vector in
    device layout, fully validated.  A leaf under ``layers.`` / ``transformer.`` / ``decoder.`` / ``norm.`` that the
    parameter inventory does not know is an error (a silent drop would hide a naming mismatch); leaves outside the
    model (e.g. the empty ``state``) are ignored."""
    shapes = _params.equinox_shapes(conf)
    known: "OrderedDict[str, np.ndarray]" = OrderedDict()
    unknown = []
    for k, v in leaves.items():
        c = canonical_name(k)
        if c in shapes:
            if c in known:
                raise KeyError(f"leaf {c} given twice ({k})")
            known[c] = v
        elif c.split(".")[0] in ("layers", "transformer", "decoder", "norm"):
            unknown.append(k)
    if unknown:
        raise KeyError(f"{len(unknown)} model leaves are not in the parameter inventory, e.g. {unknown[:4]}")
    return _params.equinox_to_flat(known, conf, ensemble_select=ensemble_select)


def flat_to_leaves(flat: np.ndarray, conf: dict, ensemble: bool = True) -> "OrderedDict[str, np.ndarray]":
    """Inverse: Equinox-shaped leaves, each with a leading ensemble axis of size 1 when ``ensemble``."""
    eq = _params.flat_to_equinox(np.asarray(flat, dtype=np.float32), conf)
    return OrderedDict((k, v[None] if ensemble else v) for k, v in eq.items())


def _restore_orbax(path: str, step=None):
    import orbax.checkpoint as ocp   # noqa: WPS433 (only where the reference runs)
    mgr = ocp.CheckpointManager(os.path.abspath(path), item_names=("params", "state"))
    step = mgr.latest_step() if step is None else step
    if step is None:
        raise FileNotFoundError(f"no Orbax checkpoint in {path}")
    restored = mgr.restore(step, args=ocp.args.Composite(params=ocp.args.StandardRestore()))
    return restored["params"], int(step), mgr.metadata()


def cmd_to_amt(a):
    tree, step, meta = _restore_orbax(a.src, a.step)
    conf = dict(model_config)
    if isinstance(meta, dict) and isinstance(meta.get("model"), dict):
        conf.update({k: v for k, v in meta["model"].items() if k in conf})
    flat = leaves_to_flat(flatten_tree(tree), conf, ensemble_select=a.ensemble_select)
    save_checkpoint(a.dst, flat, conf, step=step)
    print(f"wrote {a.dst}/params_{step}.f32 ({flat.size} parameters, ensemble member {a.ensemble_select})")


def cmd_npz_to_amt(a):
    with np.load(a.src) as z:
        leaves = {k: z[k] for k in z.files}
    conf = dict(model_config)
    flat = leaves_to_flat(leaves, conf, ensemble_select=a.ensemble_select)
    save_checkpoint(a.dst, flat, conf, step=a.step or 0)
    print(f"wrote {a.dst}/params_{a.step or 0}.f32 ({flat.size} parameters)")


def cmd_from_amt(a):
    flat, conf, step = load_checkpoint(a.src, a.step)
    leaves = flat_to_leaves(flat, conf, ensemble=True)
    if a.npz:
        np.savez(a.npz, **leaves)
        print(f"wrote {a.npz} ({len(leaves)} leaves, ensemble axis 1)")
    if a.reference is None:
        if not a.npz:
            raise SystemExit("from-amt needs --reference (to rebuild the Equinox pytree) and/or --npz")
        return
    sys.path.insert(0, os.path.abspath(a.reference))
    import equinox as eqx   # noqa: WPS433
    import jax
    import orbax.checkpoint as ocp
    from model import OutputSequenceGenerator, get_model_metadata   # the reference's own model.py
    from model import model_config as ref_config

    @eqx.filter_vmap(out_axes=(eqx.if_array(0), eqx.if_array(0)))
    def make_ensemble(key):   # infer.py:179-183
        return eqx.nn.make_with_state(OutputSequenceGenerator)(ref_config, key)

    model, state = make_ensemble(jax.random.split(jax.random.PRNGKey(0), 1))
    params = eqx.filter(model, eqx.is_inexact_array)
    paths, treedef = jax.tree_util.tree_flatten_with_path(params)

    def name(path):   # the reference's own path (with the Sequential level) -> this package's leaf name
        return canonical_name(".".join(str(getattr(k, "name", getattr(k, "idx", getattr(k, "key", k)))) for k in path))

    new_leaves = []
    for path, leaf in paths:
        k = name(path)
        if k not in leaves:
            raise KeyError(f"reference leaf {k} has no counterpart in the flat layout")
        if tuple(leaf.shape) != tuple(leaves[k].shape):
            raise ValueError(f"{k}: reference shape {tuple(leaf.shape)} vs {tuple(leaves[k].shape)}")
        new_leaves.append(jax.numpy.asarray(leaves[k], dtype=leaf.dtype))
    params = jax.tree_util.tree_unflatten(treedef, new_leaves)
    mgr = ocp.CheckpointManager(os.path.abspath(a.dst), item_names=("params", "state"),
                                metadata=get_model_metadata(), options=ocp.CheckpointManagerOptions(create=True))
    mgr.save(step, args=ocp.args.Composite(params=ocp.args.StandardSave(params), state=ocp.args.StandardSave(state)))
    mgr.wait_until_finished()
    print(f"wrote Orbax checkpoint step {step} to {a.dst}")


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("to-amt"); p.add_argument("src"); p.add_argument("dst"); p.add_argument("--step", type=int)
    p.add_argument("--ensemble-select", type=int, default=0); p.set_defaults(fn=cmd_to_amt)
    p = sub.add_parser("npz-to-amt"); p.add_argument("src"); p.add_argument("dst"); p.add_argument("--step", type=int)
    p.add_argument("--ensemble-select", type=int, default=0); p.set_defaults(fn=cmd_npz_to_amt)
    p = sub.add_parser("from-amt"); p.add_argument("src"); p.add_argument("dst"); p.add_argument("--step", type=int)
    p.add_argument("--reference"); p.add_argument("--npz"); p.set_defaults(fn=cmd_from_amt)
    a = ap.parse_args(argv)
    a.fn(a)


if __name__ == "__main__":
    main()
