import sys, torch, numpy as np, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from util import P, DT, stream
from audio_to_midi_amd import lib as L
lib = L.load()
shapes = [(256000,4,8),(256000,8,4),(128000,8,16),(128000,16,8),(64000,16,32),(64000,32,16),(32000,32,64),(32000,64,32),
          (16000,64,128),(16000,128,64),(8000,128,256),(8000,256,128),(4000,256,512),(4000,512,256),
          (128000,8,8),(4000,256,256),(4000,1024,256),(4000,512,64),(4000,320,256),(4000,96,256)]
def bench(N,M,Kp, iters=20):
    G = torch.randn(N,M,device='cuda',dtype=torch.bfloat16); A = torch.randn(N,Kp,device='cuda',dtype=torch.bfloat16)
    dW = torch.zeros(M,Kp,device='cuda'); db = torch.zeros(M,device='cuda')
    f = lambda: lib.amt_op_gemm_tn_wgrad(DT["bfloat16"], P(G), M, P(A), Kp, N, M, Kp, P(dW), Kp, P(db), 0, stream())
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)*1e3/iters
    print(f"N={N:7d} M={M:5d} Kp={Kp:4d}  {us:8.2f} us  {2.0*N*M*Kp/us/1e6:8.2f} TFLOP/s  {(N*(M+Kp)*2)/us/1e3:8.1f} GB/s", flush=True)
for s in shapes: bench(*s)
