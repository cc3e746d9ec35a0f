"""Rider workgroup start / end (wall clock) of the last C = 128 Block-backward launch; timing build only."""
import ctypes as C, os, sys
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
buf = np.zeros((128, 256, 8), dtype=np.uint64)
L.amt_debug_read_stamps(C.c_void_p(buf.ctypes.data))
s = buf[126]; live = s[:, 0] > 0; s = s[live].astype(np.int64)
print("rider wgs stamped", int(live.sum()))
t0 = s[:, 0].min()
st, en = (s[:, 0] - t0) / 100.0, (s[:, 1] - t0) / 100.0
print("rider start  us: min %.2f median %.2f max %.2f" % (st.min(), np.median(st), st.max()))
print("rider length us: min %.2f median %.2f max %.2f" % ((en - st).min(), np.median(en - st), (en - st).max()))
print("rider end    us: median %.2f max %.2f" % (np.median(en), en.max()))
order = np.argsort(st)
print("first 8 by start:", [(round(float(st[i]), 2), round(float(en[i] - st[i]), 2)) for i in order[:8]])
print("last 8 by start:", [(round(float(st[i]), 2), round(float(en[i] - st[i]), 2)) for i in order[-8:]])
for slot in (86, 87):
    c = buf[slot]; lv = c[:, 0] > 0; c = c[lv].astype(np.int64)
    if len(c): print("slot", slot, "chain wgs", len(c), "start %.2f .. %.2f end %.2f .. %.2f (us rel. to first rider start)" % ((c[:, 6].min() - t0) / 100.0, (c[:, 6].max() - t0) / 100.0, (c[:, 7].min() - t0) / 100.0, (c[:, 7].max() - t0) / 100.0))
