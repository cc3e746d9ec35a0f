"""What does the per-step upload cost the step?  Variants around the same captured step (cfg2).  python scratch/feeder_cost.py"""
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
dp = T.DataParallelStep(be, bucket, be.split, dist=None, overlap=False, graph=True)
dp.prepare(seed=1)
feeder = T.InputFeeder(be.audio, y)
ha, hy = T.InputFeeder.pin(a_np, y_np, model.audio_dtype)
copy = torch.cuda.Stream(device=dev)
sa, sy = torch.empty_like(a), torch.empty_like(y)

def run(name, step, n=60):
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    print(f"{name:78s} {(time.perf_counter() - t0) / n * 1e3:.4f} ms/step", flush=True)

def s_res(): dp.step()
def s_d2d():
    be.audio.copy_(sa, non_blocking=True); y.copy_(sy, non_blocking=True); dp.step()
def s_h2d_side():
    dp.step()
    with torch.cuda.stream(copy):
        sa.copy_(ha, non_blocking=True); sy.copy_(hy, non_blocking=True)
def s_h2d_main():
    sa.copy_(ha, non_blocking=True); sy.copy_(hy, non_blocking=True); dp.step()
def s_h2d_main_direct():
    be.audio.copy_(ha, non_blocking=True); y.copy_(hy, non_blocking=True); dp.step()
def s_feeder():
    feeder.take(); dp.step(); feeder.prefetch(ha, hy)
for rep in range(2):
    run("resident inputs", s_res)
    run("+ two device-to-device copies in front of the step (no upload)", s_d2d)
    run("+ upload into staging on a copy stream, no events, no D2D", s_h2d_side)
    run("+ upload into staging on the MAIN stream in front of the step (not overlapped)", s_h2d_main)
    run("+ upload straight into the inputs on the MAIN stream (not overlapped, no staging)", s_h2d_main_direct)
    run("InputFeeder: upload on the copy stream + events + D2D", s_feeder)

ready, taken = torch.cuda.Event(), torch.cuda.Event()
def s_ready_only():      # upload on the copy stream, main waits for it at the next step (no 'taken' hand-back)
    torch.cuda.current_stream().wait_event(ready)
    dp.step()
    with torch.cuda.stream(copy):
        sa.copy_(ha, non_blocking=True); sy.copy_(hy, non_blocking=True)
        ready.record(copy)
def s_taken_only():      # copy stream waits for a main-stream event, main never waits for the copy stream
    taken.record(torch.cuda.current_stream())
    dp.step()
    with torch.cuda.stream(copy):
        copy.wait_event(taken)
        sa.copy_(ha, non_blocking=True); sy.copy_(hy, non_blocking=True)
def s_events_no_copies():
    torch.cuda.current_stream().wait_event(ready)
    taken.record(torch.cuda.current_stream())
    dp.step()
    with torch.cuda.stream(copy):
        copy.wait_event(taken)
        ready.record(copy)
def s_feeder_late_wait():   # the feeder, but the main stream waits for the upload only AFTER the step's graph (D2D at the end)
    dp.step()
    torch.cuda.current_stream().wait_event(ready)
    be.audio.copy_(sa, non_blocking=True); y.copy_(sy, non_blocking=True)
    taken.record(torch.cuda.current_stream())
    with torch.cuda.stream(copy):
        copy.wait_event(taken)
        sa.copy_(ha, non_blocking=True); sy.copy_(hy, non_blocking=True)
        ready.record(copy)
ready.record(copy); torch.cuda.synchronize()
for rep in range(2):
    run("upload on the copy stream + 'ready' event, main waits at the next step", s_ready_only)
    run("copy stream waits for a main-stream event ('taken'), no wait back", s_taken_only)
    run("both events, no copies at all", s_events_no_copies)
    run("step, THEN wait for the previous upload + D2D + hand back (upload overlaps the next step)", s_feeder_late_wait)

evs = [torch.cuda.Event() for _ in range(4)]
cnt = [0]
def s_record_only():
    evs[cnt[0] % 4].record(torch.cuda.current_stream()); cnt[0] += 1
    dp.step()
def s_record_hostsync():    # triple-buffered staging: the host waits for the event of TWO steps ago before reusing its staging pair
    k = cnt[0]; cnt[0] += 1
    torch.cuda.current_stream().wait_event(ready)
    be.audio.copy_(sa, non_blocking=True); y.copy_(sy, non_blocking=True)
    evs[k % 4].record(torch.cuda.current_stream())
    dp.step()
    if k >= 2: evs[(k - 2) % 4].synchronize()
    with torch.cuda.stream(copy):
        sa.copy_(ha, non_blocking=True); sy.copy_(hy, non_blocking=True)
        ready.record(copy)
def s_record_after():       # the same, the event recorded BEHIND the step's launches instead of in front of them
    k = cnt[0]; cnt[0] += 1
    torch.cuda.current_stream().wait_event(ready)
    be.audio.copy_(sa, non_blocking=True); y.copy_(sy, non_blocking=True)
    dp.step()
    evs[k % 4].record(torch.cuda.current_stream())
    if k >= 2: evs[(k - 2) % 4].synchronize()
    with torch.cuda.stream(copy):
        sa.copy_(ha, non_blocking=True); sy.copy_(hy, non_blocking=True)
        ready.record(copy)
for rep in range(2):
    run("an event recorded on the main stream in front of the step, nobody waits", s_record_only)
    run("record in front of the step + host waits for the event of two steps ago + upload", s_record_hostsync)
    run("record BEHIND the step + host waits for the event of two steps ago + upload", s_record_after)
