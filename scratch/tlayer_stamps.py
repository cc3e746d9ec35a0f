"""Per-phase timing of the XCD-team layer launch (csrc/tlayer.hip) from its in-kernel wall-clock stamps.
AMT_TLAYER=1 AMT_TLAYER_STAMPS=1 python scratch/tlayer_stamps.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("AMT_TLAYER", "1")
os.environ.setdefault("AMT_TLAYER_STAMPS", "1")
import audio_to_midi_amd as amt
from audio_to_midi_amd import synth
from oracle import np_model

conf = dict(np_model.MODEL_CONFIG)
model = amt.OutputSequenceGenerator(conf, key=1, dtype="bfloat16", gamma=1.0)
audio, labels = synth.make_batch(8, 160000, seed=4)
a = torch.from_numpy(audio).to("cuda", torch.float16)
y = torch.from_numpy(labels).cuda()
for _ in range(5):
    model.loss_and_grads(a, y, seed=5, train=True, sync=False)
torch.cuda.synchronize()
st = model.tlayer_stamps(16)
assert st is not None and model.team_status() == 0
st = st.astype(np.float64) / 100.0          # us
t0 = st[:, 0, 0].min()
print("launch span (first stamp -> last stamp): %.1f us" % (st[:, 15, 4].max() - t0))
names = ["A ln1+qkv", "B attn", "D wo+ln2+w1", "E w2"]
for kind, sel in (("local", range(0, 16, 2)), ("global", range(1, 16, 2))):
    print(kind, "layers: per-phase span of a workgroup (stamp k+1 - stamp k), us: median / p90 / max over workgroups and layers")
    for k in range(4):
        d = np.concatenate([st[:, l, k + 1] - st[:, l, k] for l in sel])
        print("  %-12s %6.2f %6.2f %6.2f" % (names[k], np.median(d), np.percentile(d, 90), d.max()))
    lay = np.concatenate([st[:, l, 4] - st[:, l, 0] for l in sel])
    print("  %-12s %6.2f" % ("layer", np.median(lay)))


def sub(name, l_sel, a, b):
    d = np.concatenate([st[:, l, b] - st[:, l, a] for l in l_sel])
    print("    %-28s %6.2f %6.2f %6.2f" % (name, np.median(d), np.percentile(d, 90), d.max()))


L = range(16)
print("inside D (all layers): median / p90 / max")
sub("entry -> wait entered", L, 2, 20); sub("wait", L, 20, 21); sub("copy-in ao, x", L, 21, 22); sub("Wo stage + residual", L, 22, 23)
sub("h rows out + LN2", L, 23, 24); sub("hn out + W1 stage", L, 24, 25); sub("GLU epilogue + u,g out", L, 25, 26); sub("return + publish", L, 26, 3)
print("inside E (all layers)")
sub("entry -> wait entered", L, 3, 27); sub("wait", L, 27, 28); sub("copy-in g,h", L, 28, 29); sub("W2 stage", L, 29, 30)
sub("epilogue + y out", L, 30, 31); sub("return + publish", L, 31, 4)
for k in range(5):
    d = st[:, 8, k] - st[:, 8, k].min()
    print("layer 8 stamp %d: skew median %.2f max %.2f us" % (k, np.median(d), d.max()))
