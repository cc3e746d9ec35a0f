"""In-situ phase timestamps of the chain kernels during hipGraph replay of the training step.
Needs a library built with AMT_EXTRA_HIPFLAGS=-DAMT_CHAIN_TIMING (python __graft_entry__.py --force)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_to_midi_amd as amt
from audio_to_midi_amd import lib as amtlib

dev = torch.device("cuda", 0)
model = amt.OutputSequenceGenerator(amt.model_config, key=1234, dtype="bfloat16", device=dev)
a, y = amt.synth.make_batch_fast(8, 160000, seed=100)
audio = torch.from_numpy(a).to(dev, model.audio_dtype); labels = torch.from_numpy(y).to(dev)
grads = torch.zeros(model.n_params, dtype=torch.float32, device=dev)
model.init_optimizer()
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
EAGER = os.environ.get("EAGER") == "1"
for i in range(30):
    if EAGER:
        fwd_bwd(i)
    else:
        g.replay()
    model.adamw_step(grads, lr_base=1e-4, step=1)
if os.environ.get("PREDICT") == "1":
    for i in range(5):
        model.predict(None, audio)
torch.cuda.synchronize()
L = amtlib.load()
buf = np.zeros((128, 256, 12), dtype=np.uint64)
rc = L.amt_debug_read_stamps(C.c_void_p(buf.ctypes.data))
print("rc", rc)
prev_end = None
for slot in range(128):
    s = buf[slot]
    live = s[:, 0] > 0
    if not live.any():
        continue
    s = s[live].astype(np.int64)
    d = np.diff(s[:, :6], axis=1).mean(axis=0)
    rt = (s[:, 7].max() - s[:, 6].min()) / 100.0      # us, 100 MHz
    cyc = s[:, 5].max() - s[:, 0].min()
    t0, t1 = s[:, 6].min() / 100.0, s[:, 7].max() / 100.0
    gap = (t0 - prev_end) if prev_end is not None else float("nan")
    prev_end = t1
    print(f"slot {slot:2d} wgs {live.sum():3d} phases(cyc) " + " ".join(f"{x:7.0f}" for x in d) +
          f" | exec {rt:.2f} us, gap since previous chain kernel ended {gap:.2f} us; first-WG start spread {(s[:, 6].max() - s[:, 6].min()) / 100.0:.2f} us")
