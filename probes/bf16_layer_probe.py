"""Per-layer error of the DEVICE's bf16 forward against the float64 oracle (needs a GPU).

After a training-mode forward at the benchmark shape per clip (full depth, 10 s = 500 frames, B = 2) the residual-stream
checkpoints of the context (include/amt_ops.h: amt_op_stream_checkpoint - the input of each of the 39 Blocks, of the final
CNN norm, of each of the 16 layers and of the decoder) are read back and compared with the oracle's tensors at the same
points (oracle/torch_model.py taps).  Prints, per checkpoint, max |d| and rms(d) / rms(reference), and the logit error.

    python probes/bf16_layer_probe.py [--gamma 1.0] [--dtype bfloat16]  >  profiles/r04_bf16_layer_probe.txt
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audio_to_midi_amd as amt  # noqa: E402
from audio_to_midi_amd import synth  # noqa: E402
from audio_to_midi_amd.model import _DevView  # noqa: E402
from oracle import np_model, torch_model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gamma", type=float, default=1.0)
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--frames", type=int, default=500)
    ap.add_argument("--batch", type=int, default=2)
    args = ap.parse_args()
    torch.set_num_threads(16)
    conf = np_model.MODEL_CONFIG
    B, S = args.batch, args.frames
    params = np_model.init_params(conf, seed=1234, gamma=args.gamma)
    audio, labels = synth.make_batch(B, 320 * S, seed=42)
    model = amt.OutputSequenceGenerator(conf, dtype=args.dtype)
    model.load_equinox_params(params)
    out = model.loss_and_grads(torch.from_numpy(audio), torch.from_numpy(labels), train=False, return_logits=True)
    L = model._L
    n = L.amt_op_stream_count(model._h)
    dev = {}
    res_dt = "<f2" if args.dtype == "bfloat16" else "<f4"
    for i in range(n):
        name = C.create_string_buffer(64)
        ptr, rows, cols = C.c_void_p(), C.c_int64(), C.c_int32()
        assert L.amt_op_stream_checkpoint(model._h, i, name, C.byref(ptr), C.byref(rows), C.byref(cols)) == 0
        t = _DevView(ptr.value, rows.value * cols.value, model.device, typestr=res_dt).tensor()
        dev[name.value.decode()] = t.to(torch.float64).cpu().numpy().reshape(B, rows.value // B, cols.value)
    # oracle with the weights the context multiplies with, float64, every stored tensor tapped
    p_or = params if args.dtype == "float32" else torch_model.round_gemm_weights(params)
    taps = {}
    def tap(name, x):
        taps[name] = x.detach()
        return x
    with torch.no_grad():
        logits_ref, _ = torch_model.forward(torch_model.to_torch(p_or, torch.float64), torch.from_numpy(audio).double(), conf, tap=tap)
    depths = conf["depths"]
    ref = {}
    for s in range(len(depths)):
        for j in range(1, depths[s] + 1):
            ref[f"cnn.{s}.{j}.x"] = taps[f"cnn.{s}.{j - 1}.y"].transpose(1, 2).numpy()
    ref["cnn.norm.x"] = taps[f"cnn.{len(depths) - 1}.{depths[-1]}.y"].transpose(1, 2).numpy()
    nl = 2 * conf["num_transformer_layers"]
    ref["tl.0.x"] = taps["cnn.norm"].transpose(1, 2).numpy()
    for l in range(1, nl):
        ref[f"tl.{l}.x"] = taps[f"tl.{l - 1}.y"].numpy()
    ref["head.x"] = taps[f"tl.{nl - 1}.y"].numpy()
    print(f"# {args.dtype}, gamma {args.gamma}, B {B}, S {S}: device residual-stream checkpoints against the float64 oracle")
    print(f"{'checkpoint':16s} {'max|d|':>10s} {'rms(d)/rms(ref)':>16s} {'rms(ref)':>9s}")
    for k in dev:
        d = dev[k] - ref[k]
        print(f"{k:16s} {np.abs(d).max():10.3e} {np.sqrt((d ** 2).mean()) / np.sqrt((ref[k] ** 2).mean()):16.3e} {np.sqrt((ref[k] ** 2).mean()):9.3f}")
    dl = out.logits.cpu().numpy() - logits_ref.numpy()
    print(f"{'logits':16s} {np.abs(dl).max():10.3e} {np.sqrt((dl ** 2).mean()) / np.sqrt((logits_ref.numpy() ** 2).mean()):16.3e} {np.sqrt((logits_ref.numpy() ** 2).mean()):9.3f}")


if __name__ == "__main__":
    main()
