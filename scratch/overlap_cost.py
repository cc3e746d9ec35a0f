"""Where do the ~0.15 ms of the overlapped step go at N = 1 (no collective at all)?  Variants of DataParallelStep.step()
around the SAME captured graph(s).  python scratch/overlap_cost.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audio_to_midi_amd as amt
from audio_to_midi_amd import train as T
dev = torch.device("cuda", 0)
model = amt.OutputSequenceGenerator(amt.model_config, key=1234, dtype="bfloat16", device=dev)
a_np, y_np = amt.synth.make_batch_fast(8, 160000, seed=100)
a, y = torch.from_numpy(a_np).to(dev, model.audio_dtype), torch.from_numpy(y_np).to(dev)
bucket = T.GradBucket.of_model(model)
be = T.HipTrainBackend(model, a, y, bucket, fused=True)
comm = torch.cuda.Stream(device=dev)

def capture(fn):
    fn(); be.discard(); be.sync_signals() if model._L.amt_first_half_status(model._h) >= 0 else None
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(); be.discard()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        be.advance_seed(); fn()
    return g

g_full = capture(lambda: be.full(1))
g_sig = capture(lambda: be.full_signal(1))
be.sync_signals(); torch.cuda.synchronize()

def run(name, step, n=60):
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    print(f"{name:70s} {(time.perf_counter() - t0) / n * 1e3:.4f} ms/step", flush=True)

def s_fused():
    g_full.replay(); be.apply(1, 1e-4)
def s_sig_nowait():          # the signalled graph, nobody waits (signals pile up: harmless), optimizer on the same stream
    g_sig.replay(); be.apply(1, 1e-4)
def s_sig_wait_nojoin():     # + the wait kernel on the comm stream, no join back
    g_sig.replay()
    with torch.cuda.stream(comm):
        be.wait_first_half(comm)
    be.apply(1, 1e-4)
def s_sig_wait_join():       # + the optimizer waits for the comm stream (DataParallelStep.step)
    g_sig.replay()
    with torch.cuda.stream(comm):
        be.wait_first_half(comm)
    torch.cuda.current_stream().wait_stream(comm)
    be.apply(1, 1e-4)
def s_full_join():           # the plain graph + a join with an idle stream in front of the optimizer
    g_full.replay()
    with torch.cuda.stream(comm):
        pass
    ev = torch.cuda.Event(); ev.record(comm)
    torch.cuda.current_stream().wait_event(ev)
    be.apply(1, 1e-4)
for rep in range(2):
    run("fused (one graph + optimizer)", s_fused)
    run("signalled graph, no wait kernel", s_sig_nowait)
    be.sync_signals()
    run("signalled graph + wait kernel on the comm stream, no join", s_sig_wait_nojoin)
    run("signalled graph + wait kernel + optimizer joins the comm stream", s_sig_wait_join)
    run("plain graph + optimizer waits for an event of an idle stream", s_full_join)
print("first-half wait time-outs:", model.first_half_status(), "optimizer time-outs:", model.optimizer_status())
