"""Overfit one synthetic batch for a few hundred graph-replayed steps (DataParallelStep, warm-up + cosine schedule):
the loss must fall monotonically-ish and stay finite; both step forms (one call / two-phase) must agree at the end."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_to_midi_amd as amt
from audio_to_midi_amd import train as T
dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
a, y = amt.synth.make_batch_fast(8, 160000, seed=100)
audio = torch.from_numpy(a).to(dev, torch.bfloat16); labels = torch.from_numpy(y).to(dev)
out = {}
for overlap in (False, True):
    model = amt.OutputSequenceGenerator(amt.model_config, key=1234, dtype="bfloat16", device=dev)
    model.init_optimizer()
    bucket = T.GradBucket(model.n_params, dev)
    be = T.HipTrainBackend(model, audio, labels, bucket)
    dp = T.DataParallelStep(be, bucket, be.split, dist=None, overlap=overlap, graph=True)
    sched = T.create_learning_rate_schedule(2e-4, 20, steps)
    dp.prepare(seed=0)
    losses = []
    for i in range(steps):
        dp.lr_base = sched(i)
        dp.step()
        if i % 25 == 0 or i == steps - 1:
            l, ok = dp.result()
            losses.append(l)
            print(f"overlap={overlap} step {i:4d} lr {dp.lr_base:.2e} loss {l:.1f} valid {ok}", flush=True)
            assert ok and np.isfinite(l)
    assert losses[-1] < 0.5 * losses[0], losses
    out[overlap] = losses
print("final losses:", out[False][-1], out[True][-1])
