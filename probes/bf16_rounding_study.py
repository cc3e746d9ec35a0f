"""Which STORED tensor carries the bf16 logit error?  (CPU only, no GPU needed.)

The float64 oracle (oracle/torch_model.py) is run at the benchmark shape per clip (full depth, 10 s = 500 frames, B = 2)
with a `tap` that rounds chosen classes of stored tensors to bfloat16 - the roundings the device path performs when it
writes a tensor in its activation dtype - and the maximum |dlogit| against the unrounded run is reported for
  * every class alone,
  * all classes together (the device's bf16 path, up to MFMA summation order and the P rounding inside attention),
  * all classes but one.
GEMM weights are bf16-rounded in both runs (util.bf16_round_gemm_weights), as in the parity tests.

    python probes/bf16_rounding_study.py [--gamma 1.0] [--train]  >  profiles/r04_bf16_rounding_study.txt
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import np_model, torch_model  # noqa: E402


bf16_round_gemm_weights = torch_model.round_gemm_weights


CLASSES = {
    "tr_resid": lambda n: n.startswith("tl.") and n.split(".")[2] in ("h", "y") or n == "cnn.norm",
    "tr_ln": lambda n: (n.startswith("tl.") and n.split(".")[2] in ("xn", "hn")) or n == "head.hn",
    "tr_qkv": lambda n: n.startswith("tl.") and n.split(".")[2] in ("q_raw", "q", "c", "k_raw", "k", "v"),
    "tr_ao": lambda n: n.startswith("tl.") and n.split(".")[2] == "ao",
    "tr_p": lambda n: n.startswith("tl.") and n.split(".")[2] == "p",
    "tr_ffn": lambda n: n.startswith("tl.") and n.split(".")[2] in ("u", "g"),
    "cnn_resid": lambda n: n.startswith("cnn.") and n.endswith(".y"),
    "cnn_branch": lambda n: n.startswith("cnn.") and n.split(".")[-1] in ("ln", "u", "h", "z"),
}


def rb(t):
    return t.to(torch.bfloat16).to(t.dtype)


def rh(t):
    return t.to(torch.float16).to(t.dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gamma", type=float, default=1.0)
    ap.add_argument("--frames", type=int, default=500)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    conf = np_model.MODEL_CONFIG
    S, B = args.frames, args.batch
    from audio_to_midi_amd import synth
    audio, _ = synth.make_batch(B, 320 * S, seed=42)
    params = bf16_round_gemm_weights(np_model.init_params(conf, seed=1234, gamma=args.gamma))
    pt = torch_model.to_torch(params, torch.float64)
    a = torch.from_numpy(audio).double()
    a = rb(a)                                    # the device reads bf16 audio
    with torch.no_grad():
        t = time.time()
        ref, _ = torch_model.forward(pt, a, conf)
        print(f"# gamma {args.gamma}, B {B}, S {S}: reference forward {time.time() - t:.1f} s, rms(logit) {float(ref.pow(2).mean().sqrt()):.3f}")

        def run(active, fp16=()):
            def tap(name, x):
                for c in fp16:
                    if CLASSES[c](name):
                        return rh(x)
                for c in active:
                    if CLASSES[c](name):
                        return rb(x)
                return x
            lg, _ = torch_model.forward(pt, a, conf, tap=tap)
            d = (lg - ref).abs()
            return float(d.max()), float(d.pow(2).mean().sqrt())

        allc = list(CLASSES)
        print(f"{'rounded classes':42s} max|dlogit|   rms")
        for c in allc:
            m, r = run([c])
            print(f"only {c:37s} {m:.3e}   {r:.3e}", flush=True)
        m, r = run(allc)
        print(f"{'ALL (device bf16 path)':42s} {m:.3e}   {r:.3e}", flush=True)
        for c in allc:
            m, r = run([x for x in allc if x != c])
            print(f"all but {c:34s} {m:.3e}   {r:.3e}", flush=True)
        m, r = run([x for x in allc if x not in ("tr_resid", "cnn_resid")])
        print(f"{'all but tr_resid + cnn_resid':42s} {m:.3e}   {r:.3e}", flush=True)
        m, r = run(allc, fp16=("tr_resid", "cnn_resid"))
        print(f"{'ALL, tr_resid + cnn_resid stored as fp16':42s} {m:.3e}   {r:.3e}", flush=True)
        m, r = run(allc, fp16=("tr_resid",))
        print(f"{'ALL, tr_resid stored as fp16':42s} {m:.3e}   {r:.3e}", flush=True)
        m, r = run([], fp16=allc)
        print(f"{'every class stored as fp16 (the reference)':42s} {m:.3e}   {r:.3e}", flush=True)


if __name__ == "__main__":
    main()
