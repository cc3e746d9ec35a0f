"""Single weight-gradient problems at 32-clip sizes through the op ABI: 64 x 64 tiles against the large-N tiles.
python scratch/wgrad_big_bench.py   (AMT_WGRAD_BIG=0 | default, AMT_WGRAD_TARGET_BIG=<wgs>)"""
import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from util import P, DT, stream
from audio_to_midi_amd import lib as L
lib = L.load()
shapes = [(16000, 1024, 256), (16000, 256, 512), (16000, 256, 256), (16000, 320, 256), (16000, 512, 64),
          (32000, 256, 128), (32000, 128, 256), (8000, 512, 256), (8000, 256, 512)]
def bench(N, M, Kp, iters=30):
    G = torch.randn(N, M, device='cuda', dtype=torch.bfloat16); A = torch.randn(N, Kp, device='cuda', dtype=torch.bfloat16)
    dW = torch.zeros(M, Kp, device='cuda'); db = torch.zeros(M, device='cuda')
    f = lambda: lib.amt_op_gemm_tn_wgrad(DT["bfloat16"], P(G), M, P(A), Kp, N, M, Kp, P(dW), Kp, P(db), 0, stream())
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us
print("env: BIG=%s TARGET_BIG=%s" % (os.environ.get("AMT_WGRAD_BIG"), os.environ.get("AMT_WGRAD_TARGET_BIG")))
tot = 0
for s in shapes:
    us = bench(*s); tot += us
    N, M, Kp = s
    print(f"N={N:6d} M={M:5d} Kp={Kp:4d}  {us:8.2f} us  {2.0 * N * M * Kp / us / 1e6:8.1f} TFLOP/s", flush=True)
print("sum %.1f us" % tot)
