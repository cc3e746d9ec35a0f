import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import *
from audio_to_midi_amd.rope import precompute_frequencies
rng = np.random.default_rng(0)
B, H, S = 8, 4, 500
N = B * S
dtype = sys.argv[1] if len(sys.argv) > 1 else "bfloat16"
rf = precompute_frequencies(64, 512)
cos_d, sin_d = f32(rf.cos_freq), f32(rf.sin_freq)
qcd = dev(rng.standard_normal((N, 320)), dtype); kvd = dev(rng.standard_normal((N, 512)), dtype); dOd = dev(rng.standard_normal((N, 256)), dtype)
O = torch.zeros((N, 256), device="cuda", dtype=qcd.dtype); LSE = torch.zeros(B * H * S, device="cuda")
call("amt_op_attn_fwd", DT[dtype], 0, P(qcd), 320, P(kvd), 512, P(O), 256, P(LSE), P(cos_d), P(sin_d), B, S, H, 1, 5, 0.0, stream())
outs = []
for it in range(6):
    dQC = torch.zeros((N, 320), device="cuda", dtype=qcd.dtype); dKV = torch.zeros((N, 512), device="cuda", dtype=qcd.dtype)
    delta = torch.zeros(B * H * S, device="cuda")
    call("amt_op_attn_bwd", DT[dtype], 0, P(qcd), 320, P(kvd), 512, P(O), 256, P(LSE), P(cos_d), P(sin_d), B, S, H, 1, 5, 0.0,
         P(dOd), 256, P(dQC), 320, P(dKV), 512, P(delta), stream())
    outs.append((dQC.clone(), dKV.clone(), delta.clone()))
for it in range(1, 6):
    dq = (outs[it][0] != outs[0][0]); dk = (outs[it][1] != outs[0][1]); dl = (outs[it][2] != outs[0][2])
    print(it, "dQ diff", int(dq.sum()), "dKV diff", int(dk.sum()), "delta diff", int(dl.sum()),
          "rows dq", dq.any(dim=1).nonzero().flatten()[:8].tolist(), "rows dkv", dk.any(dim=1).nonzero().flatten()[:8].tolist())
