import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import audio_to_midi_amd as amt
from audio_to_midi_amd import synth, train as T
from oracle import np_model
conf = dict(np_model.MODEL_CONFIG)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
t0 = time.time()
batches = [synth.make_batch(4, 80000, seed=500 + i) for i in range(64)]
print("gen", time.time() - t0)
def loader():
    for k in range(steps):
        a, y = batches[k % len(batches)]
        yield y, a
model = amt.OutputSequenceGenerator(conf, key=7, dtype="bfloat16")
t0 = time.time()
out = T.train(model, loader(), num_steps=steps, base_learning_rate=lr, warmup_steps=20, print_every=100, log=print, check_every=50)
torch.cuda.synchronize()
print("train", time.time() - t0, out["applied"])
a, y = synth.make_batch(4, 80000, seed=9000)
logits, probs = model.predict(None, torch.from_numpy(a))
p = probs.cpu().numpy()
print("probs: max", p.max(), "cells > 0.5:", (p > 0.5).sum(), "> 0.4:", (p > 0.4).sum(), "> 0.1:", (p > 0.1).sum(), "labels > 0.5:", (y > 0.5).sum())
