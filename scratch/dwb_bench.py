import sys, torch, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from util import P, DT, stream
from audio_to_midi_amd import lib as L
lib = L.load()
def bench(B, Lr, C, iters=30):
    N = B*Lr
    bf = torch.bfloat16
    dLN = torch.randn(N, C, device='cuda', dtype=bf); X = torch.randn(N, C, device='cuda', dtype=bf)
    dY = torch.randn(N, C, device='cuda', dtype=bf); Z = torch.randn(N, C, device='cuda', dtype=bf)
    dX = torch.zeros(N, C, device='cuda', dtype=bf)
    wdw = torch.randn(7*C, device='cuda'); bdw = torch.randn(C, device='cuda'); lnw = torch.randn(C, device='cuda')
    mean = torch.zeros(N, device='cuda'); rstd = torch.ones(N, device='cuda')
    outs = [torch.zeros(n, device='cuda') for n in (C, C, 7*C, C, C)]
    f = lambda: lib.amt_op_dwblock_bwd(DT["bfloat16"], P(dLN), P(X), P(dY), P(Z), P(wdw), P(bdw), P(mean), P(rstd), P(lnw), P(dX), *[P(o) for o in outs], B, Lr, C, stream())
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    print(f"dbg={os.environ.get('AMT_DBG','0')} C={C:4d} rows={N:7d}: {e0.elapsed_time(e1)*1e3/iters:8.2f} us", flush=True)
for C, Lr in ((4, 32000), (128, 1000)):
    bench(8, Lr, C)
