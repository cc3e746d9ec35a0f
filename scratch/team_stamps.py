"""Per-workgroup timeline of the team forward run (timing build, AMT_CHAIN_TIMING): work vs barrier wait per Block."""
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
lo, hi = int(os.environ.get("LO", "9")), int(os.environ.get("HI", "29"))
st, en = s[lo:hi + 1, :, 6] / 100.0, s[lo:hi + 1, :, 7] / 100.0     # [slot][wg] us
live = (s[lo:hi + 1, :, 0] > 0).all(axis=0)
print("slots", lo, hi, "workgroups with a tile in every slot:", live.sum())
st, en = st[:, live], en[:, live]
t0 = st.min()
work = (en - st).sum(axis=0)
wait = (st[1:] - en[:-1]).sum(axis=0)
print(f"run span {en.max() - t0:.2f} us; per workgroup: work {work.mean():.2f} us (min {work.min():.2f}, max {work.max():.2f}), "
      f"between tiles {wait.mean():.2f} us (min {wait.min():.2f}, max {wait.max():.2f})")
print("per block: work mean %.2f us, between-tile mean %.2f us" % ((en - st).mean(), (st[1:] - en[:-1]).mean()))
# how far apart the workgroups are when they start each block
print("start spread per slot (us):", " ".join(f"{x:.1f}" for x in (st.max(axis=1) - st.min(axis=1))))
# timeline of one workgroup
w = 0
print("wg0 timeline (start,end):", " ".join(f"({a - t0:.1f},{b - t0:.1f})" for a, b in zip(st[:, w], en[:, w])))
d = np.diff(s[lo:hi + 1, :, :6][:, live], axis=2).mean(axis=(0, 1))
print("phase cycles (row, stage1 mma, gelu+sync, stage2 mma, epilogue+copyout):", " ".join(f"{x:.0f}" for x in d))
# per team (observed placement: workgroup b sits on XCC b % 8): how long after the LAST member finished Block j does the
# FIRST / LAST member start Block j + 1, and how uneven is the members' work
idx = np.arange(256)[live]
for team in range(8):
    m = (idx % 8) == team
    s_, e_ = st[:, m], en[:, m]
    last_end = e_[:-1].max(axis=1)
    first_start = s_[1:].min(axis=1)
    last_start = s_[1:].max(axis=1)
    wk = (e_ - s_)
    print(f"team {team}: members {m.sum()}, barrier exit after last arrival: first member {np.mean(first_start - last_end):.2f} us, last member {np.mean(last_start - last_end):.2f} us; "
          f"work per block mean {wk.mean():.2f} max-member {wk.max(axis=1).mean():.2f} us; period {np.mean(np.diff(s_.min(axis=1))):.2f} us")
team0 = (idx % 8) == 0
j = 10
print("team 0, block", lo + j, "member start/end relative to the team's first start:")
base = st[j, team0].min()
print(" ".join(f"({a - base:.1f},{b - base:.1f})" for a, b in zip(st[j, team0], en[j, team0])))
b_in, b_out = s[lo:hi + 1, :, 8][:, live] / 100.0, s[lo:hi + 1, :, 9][:, live] / 100.0
print("team 0, block", lo + j, "per member: tile end -> barrier entered -> barrier left -> next tile start (us after the team's first start)")
print(" ".join(f"({e - base:.1f},{bi - base:.1f},{bo - base:.1f},{ns - base:.1f})" for e, bi, bo, ns in zip(en[j, team0], b_in[j, team0], b_out[j, team0], st[j + 1, team0])))
c_left = s[lo:hi + 1, :, 10][:, live]
c_start = s[lo:hi + 1, :, 0][:, live]
dc = (c_start[1:] - c_left[:-1])[:, team0]
print("team 0: shader cycles from barrier exit (block j) to tile start (block j+1), block", lo + j, ":", " ".join(str(int(x)) for x in dc[j]))
c_items = s[lo:hi + 1, :, 11][:, live]
d1 = (c_items[1:] - c_left[:-1])[:, team0]
d2 = (c_start[1:] - c_items[1:])[:, team0]
print("  of which barrier exit -> pointers of the next Block read from LDS:", " ".join(str(int(x)) for x in d1[j]))
print("  and pointers read -> tile start:", " ".join(str(int(x)) for x in d2[j]))
if int(os.environ.get("AMT_DBG_NOSTORE", "0")) & 16384:
    c_spin = s[lo:hi + 1, :, 5][:, live]
    print("  spin loop of 2000 scalar adds right after the barrier took (cycles):", " ".join(str(int(x)) for x in (c_spin[:-1] - c_left[:-1])[:, team0][j]))
    print("  spin end -> pointers read:", " ".join(str(int(x)) for x in (c_items[1:] - c_spin[:-1])[:, team0][j]))
ident = (s[126, :, 0][live] & 0xffffffff)
rank, xcc = ident & 0xff, (ident >> 8) & 0xff
first = st[j].argmin()
print("who starts a Block first, per team (slot %d): " % (lo + j), end="")
for team in range(8):
    m = np.where(xcc == team)[0]
    if len(m) == 0: continue
    w = m[st[j, m].argmin()]
    print(f"xcc {team}: wg {idx[w]} rank {rank[w]} (+{np.sort(st[j, m])[1] - st[j, w]:.1f} us ahead);", end=" ")
print()
m0 = np.where(xcc == 0)[0]
order = m0[np.argsort(rank[m0])]
print("team 0 by rank: (rank: cycles pointers-read -> tile start | tile work us)")
print(" ".join(f"{rank[w]}:{int(c_start[j + 1, w] - c_items[j + 1, w])}|{en[j + 1, w] - st[j + 1, w]:.1f}" for w in order))
c_loop = s[lo:hi + 1, :, 4][:, live]
print("  pointers read -> top of tile loop body:", " ".join(str(int(c_loop[j + 1, w] - c_items[j + 1, w])) for w in order))
print("  top of tile loop body -> tile start:", " ".join(str(int(c_start[j + 1, w] - c_loop[j + 1, w])) for w in order))
