"""fp8 global attention vs the bf16 path: logits / loss / gradients at full depth (cfg2 clip length)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import audio_to_midi_amd as amt
from audio_to_midi_amd import synth
from oracle import np_model

for T in (80000, 160000):
    ref = amt.OutputSequenceGenerator(np_model.MODEL_CONFIG, key=1234, dtype="bfloat16", gamma=1.0)
    f8 = amt.OutputSequenceGenerator(np_model.MODEL_CONFIG, key=1234, dtype="bfloat16", gamma=1.0)
    f8.set_attn_fp8(True)
    audio, labels = synth.make_batch(4, T, seed=6)
    a = torch.from_numpy(audio).to("cuda", torch.bfloat16)
    y = torch.from_numpy(labels).cuda()
    r = ref.loss_and_grads(a, y, seed=11, train=False)
    for it in range(3):
        q = f8.loss_and_grads(a, y, seed=11, train=False)
        gd = float((q.grads_flat - r.grads_flat).norm() / r.grads_flat.norm())
        print(f"T={T} call {it}: loss {q.loss:.5f} vs {r.loss:.5f} (rel {abs(q.loss-r.loss)/abs(r.loss):.2e})  grads relL2 {gd:.3e}")
    lr, _ = ref.predict(None, a.float().cpu())
    lq, _ = f8.predict(None, a.float().cpu())
    print("logits max abs diff", float((lr - lq).abs().max()), "rms", float((lr - lq).pow(2).mean().sqrt()), "logit rms", float(lr.pow(2).mean().sqrt()))
