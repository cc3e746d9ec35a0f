"""Round 4: can this torch / ROCm (a) record an EXTERNAL event inside a captured graph, (b) capture an RCCL all-reduce?"""
import os, time, torch
import torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
x = torch.zeros(1 << 22, device=dev)
y = torch.zeros(1 << 22, device=dev)
side = torch.cuda.Stream(device=dev)
# ---- (a) external event
try:
    ev = torch.cuda.Event(external=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        x.add_(1); x.add_(1)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        x.add_(1.0)
        ev.record()
        for _ in range(200):
            x.mul_(1.0)
    torch.cuda.synchronize()
    x.zero_(); y.zero_()
    torch.cuda.synchronize()
    g.replay()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        y.copy_(x)            # must see x == 1 (after the first node), and should run BEFORE the 200 muls are over
        e_side = torch.cuda.Event(enable_timing=False); e_side.record()
    e_main = torch.cuda.Event(); e_main.record()
    torch.cuda.synchronize()
    print("(a) external event inside a captured graph: OK, side stream saw x =", float(y[0]), float(y[-1]))
except Exception as e:
    print("(a) external event: FAILED:", type(e).__name__, e)
# (b) capturing an RCCL all-reduce (world 1) inside torch.cuda.graph ABORTED the process on this image (ProcessGroupNCCL's
# watchdog queries the work's events during capture: c10::Error from Watchdog::run, core dumped; gpurun_out r4, not kept)
def timeit(f, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
t = torch.ones(1 << 20, device=dev)
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3):
    t.add_(1.0)
print("one-kernel graph replay: %.1f us; eager kernel: %.1f us" % (timeit(g3.replay), timeit(lambda: t.add_(1.0))))
