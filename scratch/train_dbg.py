import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
import audio_to_midi_amd as amt
model = amt.OutputSequenceGenerator(amt.model_config, key=1234, dtype=sys.argv[1] if len(sys.argv)>1 else "bfloat16")
a, y = amt.synth.make_batch_fast(8, 160000, seed=100)
a = torch.from_numpy(a).cuda(); y = torch.from_numpy(y).cuda()
for step in range(1, 13):
    o = model.loss_and_grads(a, y, seed=step, train=True)
    g = o.grads_flat
    print(step, "loss", o.loss, "valid", o.grads_valid, "gnorm", float(g.norm()), "gmax", float(g.abs().max()), flush=True)
    model.adamw_step(g, lr_base=1e-4, step=step)
    w = model.weights_tensor()
    print("   wnorm", float(w.norm()), "wmax", float(w.abs().max()))
