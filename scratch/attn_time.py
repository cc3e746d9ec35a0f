"""Stand-alone timing of the attention kernels at the cfg2 shape (B=8, S=500, H=4) through the op-level ABI."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audio_to_midi_amd import lib as L
from audio_to_midi_amd.rope import precompute_frequencies
lib = L.load()
B, S, H = int(os.environ.get("B", 8)), 500, 4
N = B * S
rf = precompute_frequencies(64, 512)
cos = torch.tensor(rf.cos_freq, dtype=torch.float32, device="cuda"); sin = torch.tensor(rf.sin_freq, dtype=torch.float32, device="cuda")
g = torch.Generator(device="cuda").manual_seed(1)
qc = torch.randn(N, 320, device="cuda", generator=g).bfloat16(); kv = torch.randn(N, 512, device="cuda", generator=g).bfloat16()
dO = torch.randn(N, 256, device="cuda", generator=g).bfloat16()
O = torch.zeros(N, 256, device="cuda", dtype=torch.bfloat16); LSE = torch.zeros(B * H * S, device="cuda")
dQC = torch.zeros(N, 320, device="cuda", dtype=torch.bfloat16); dKV = torch.zeros(N, 512, device="cuda", dtype=torch.bfloat16)
delta = torch.zeros(B * H * S, device="cuda")
P = L.ptr
st = torch.cuda.current_stream().cuda_stream
def fwd(local, p): lib.amt_op_attn_fwd(1, local, P(qc), 320, P(kv), 512, P(O), 256, P(LSE), P(cos), P(sin), B, S, H, 7, 3, p, st)
def bwd(local, p): lib.amt_op_attn_bwd(1, local, P(qc), 320, P(kv), 512, P(O), 256, P(LSE), P(cos), P(sin), B, S, H, 7, 3, p, P(dO), 256, P(dQC), 320, P(dKV), 512, P(delta), st)
def timeit(f, *a, n=50):
    for _ in range(5): f(*a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f(*a)
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
for local in (0, 1):
    for p in (0.0, 0.1):
        print(f"local={local} p={p}: fwd {timeit(fwd, local, p):7.2f} us   bwd {timeit(bwd, local, p):7.2f} us", flush=True)
