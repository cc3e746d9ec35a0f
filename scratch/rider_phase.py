"""Cycle stamps inside rider workgroups (wgrad.h: tn_body) of the LAST rider-carrying launch of the step.
Timing build only (AMT_EXTRA_HIPFLAGS=-DAMT_CHAIN_TIMING); slot 127 of amt_stamps."""
import ctypes as C, os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_to_midi_amd as amt
from audio_to_midi_amd import lib as amtlib
dev = torch.device("cuda", 0)
model = amt.OutputSequenceGenerator(amt.model_config, key=1234, dtype="bfloat16", device=dev)
a, y = amt.synth.make_batch_fast(8, 160000, seed=100)
audio = torch.from_numpy(a).to(dev, torch.bfloat16); labels = torch.from_numpy(y).to(dev)
grads = torch.zeros(model.n_params, dtype=torch.float32, device=dev)
def fwd_bwd(seed):
    model.loss_and_grads(audio, labels, scale=1.0, seed=seed, train=True, grads_out=grads, sync=False)
fwd_bwd(0); torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    fwd_bwd(0)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fwd_bwd(1)
for i in range(10):
    g.replay()
torch.cuda.synchronize()
L = amtlib.load()
buf = np.zeros((128, 256, 12), dtype=np.uint64)
L.amt_debug_read_stamps(C.c_void_p(buf.ctypes.data))
s = buf[127].astype(np.int64)
live = s[:, 0] > 0
s = s[live]
print("stamped", len(s))
groups = collections.defaultdict(list)
for r in s:
    groups[(int(r[7]) - 1, int(r[6]) - 1)].append(r)
for (shape, nsteps), rows in sorted(groups.items()):
    rows = np.array(rows)
    d = np.diff(rows[:, :5], axis=1)
    print(f"M*1000+Kp={shape} stages={nsteps} wgs={len(rows)}: cycles first-stage-ready {d[:,0].mean():.0f}  rest-of-stages {d[:,1].mean():.0f}  epilogue-issue {d[:,2].mean():.0f}  atomics-return {d[:,3].mean():.0f}  total {(rows[:,4]-rows[:,0]).mean():.0f} (max {(rows[:,4]-rows[:,0]).max()})")
