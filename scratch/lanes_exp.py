import sys, time, torch
sys.path.insert(0, '/root/repo')
import audio_to_midi_amd as amt
def make(Bl, seed):
    m = amt.OutputSequenceGenerator(amt.model_config, key=1234, dtype="bfloat16")
    a, y = amt.synth.make_batch_fast(Bl, 160000, seed=seed)
    a = torch.from_numpy(a).cuda().to(torch.bfloat16); y = torch.from_numpy(y).cuda()
    g = torch.zeros(m.n_params, device="cuda")
    st = torch.cuda.Stream()
    def step():
        m.loss_and_grads(a, y, seed=1, train=True, grads_out=g, sync=False)
    with torch.cuda.stream(st):
        step(); step()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        step()
    return m, gr, st
def run(nl, Bl, iters=20):
    lanes = [make(Bl, 10 + i) for i in range(nl)]
    def go():
        for m, gr, st in lanes:
            with torch.cuda.stream(st):
                gr.replay()
    for _ in range(3): go()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): go()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"lanes={nl} B/lane={Bl}: {dt*1e3:.3f} ms per {nl*Bl} clips -> {nl*Bl*10/dt:.0f} audio-s/s", flush=True)
run(1, 8); run(2, 4); run(4, 2); run(2, 8); run(4, 8); run(8, 1)
