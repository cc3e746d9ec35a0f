"""Does a weight-gradient launch get more efficient with more tiles at 16 000 tokens?  (stand-in for grouping the 1x1-conv
problems of several CNN Blocks into one launch): python scratch/wgrad_group_scaling.py"""
import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from util import P, DT, stream
from audio_to_midi_amd import lib as L
lib = L.load()
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
    return e0.elapsed_time(e1) * 1e3 / iters
for (N, M, Kp) in [(16000, 256, 128), (16000, 512, 128), (16000, 768, 128), (16000, 1536, 128), (16000, 128, 256), (16000, 384, 256), (16000, 768, 256)]:
    us = bench(N, M, Kp)
    print(f"N={N} M={M:5d} Kp={Kp:4d} tiles={((M+63)//64)*((Kp+63)//64):4d}  {us:8.2f} us  {2.0*N*M*Kp/us/1e6:7.1f} TFLOP/s", flush=True)
