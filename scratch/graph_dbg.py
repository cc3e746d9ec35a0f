import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
import audio_to_midi_amd as amt
model = amt.OutputSequenceGenerator(amt.model_config, key=1234, dtype="bfloat16")
a, y = amt.synth.make_batch_fast(8, 160000, seed=100)
a = torch.from_numpy(a).cuda().to(torch.bfloat16); y = torch.from_numpy(y).cuda()
grads = torch.zeros(model.n_params + 2, device="cuda"); gflat = grads[:model.n_params]
loss_buf = torch.zeros(2, device="cuda")
def fwd_bwd(seed):
    loss, _, valid, _ = model.loss_and_grads(a, y, scale=1.0, seed=seed, train=True, grads_out=gflat, sync=False)
    loss_buf.copy_(loss)
fwd_bwd(0); torch.cuda.synchronize()
if len(sys.argv) > 2:
    model.adamw_step(gflat, lr_base=0.0, step=1); torch.cuda.synchronize()
print("eager", loss_buf.tolist(), float(gflat.norm()))
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    fwd_bwd(0)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize(); print("side", loss_buf.tolist(), float(gflat.norm()))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fwd_bwd(12345)
for i in range(4):
    g.replay(); torch.cuda.synchronize(); print("replay", i, loss_buf.tolist(), float(gflat.norm()), float(gflat.abs().max()))
    if len(sys.argv) > 1:
        model.adamw_step(gflat, lr_base=1e-4, step=i+1); torch.cuda.synchronize()
fwd_bwd(12345); torch.cuda.synchronize(); print("eager again", loss_buf.tolist(), float(gflat.norm()))
