"""Timing build (AMT_CHAIN_TIMING): when do the chain workgroups and the rider workgroups of a Block-backward launch run?"""
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
for i in range(20):
    g.replay()
torch.cuda.synchronize()
L = amtlib.load()
buf = np.zeros((128, 256, 12), dtype=np.uint64)
L.amt_debug_read_stamps(C.c_void_p(buf.ctypes.data))
s = buf.astype(np.int64)
for slot in range(64, 97):
    ch = s[slot]
    live = ch[:, 0] > 0
    rl = ch[:, 10] > 0
    if not live.any():
        continue
    t0 = ch[live, 6].min() / 100.0
    cs, ce = ch[live, 6] / 100.0 - t0, ch[live, 7] / 100.0 - t0
    line = f"slot {slot}: chain wgs(<256) {live.sum()} start {cs.min():.1f}..{cs.max():.1f} end {ce.min():.1f}..{ce.max():.1f} (mean {ce.mean():.1f})"
    if rl.any():
        rs, re = ch[rl, 10] / 100.0 - t0, ch[rl, 11] / 100.0 - t0
        line += f" | riders {rl.sum()} start {rs.min():.1f}..{rs.max():.1f} (mean {rs.mean():.1f}) end {re.min():.1f}..{re.max():.1f} (mean {re.mean():.1f}) dur mean {(re - rs).mean():.1f} max {(re - rs).max():.1f}"
    print(line)
