import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_to_midi_amd as amt
from audio_to_midi_amd import synth, params
from oracle import np_model
conf = np_model.MODEL_CONFIG
model = amt.OutputSequenceGenerator(conf, key=1234, dtype="bfloat16", gamma=1.0)
audio, labels = synth.make_batch_fast(8, 160000, seed=1)
a, y = torch.from_numpy(audio), torch.from_numpy(labels)
outs = [model.loss_and_grads(a, y, train=False).grads_flat.clone() for _ in range(3)]
segs = params.device_layout(conf)
for i in (1, 2):
    d = outs[i] - outs[0]
    print("run", i, "rel", float(d.norm() / outs[0].norm()))
    rows = []
    for s in segs:
        n = s.rows * s.cols
        dd = d[s.offset:s.offset + n]; g = outs[0][s.offset:s.offset + n]
        rows.append((float(dd.norm()), float(g.norm()), s.name))
    if i == 1:
        for r in rows:
            if r[1] > 0 and (r[2].startswith("tl.") and (".w" in r[2]) or "conv.weight" in r[2] or "decoder" in r[2] or r[2].endswith("point_conv_1.weight")):
                print("   %-40s rel %.3g" % (r[2], r[0] / (r[1] + 1e-30)))

# train mode (dropout, fixed seed), eager and under hipGraph replay
dev = a.device
for mode in ("eager", "graph"):
    gbuf = torch.zeros(model.n_params, dtype=torch.float32, device="cuda")
    ac, yc = a.to("cuda", torch.bfloat16), y.cuda()
    def fb():
        model.loss_and_grads(ac, yc, seed=77, train=True, grads_out=gbuf, sync=False)
    fb(); torch.cuda.synchronize()
    if mode == "graph":
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fb()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fb()
        run = g.replay
    else:
        run = fb
    outs2 = []
    for _ in range(3):
        run(); torch.cuda.synchronize(); outs2.append(gbuf.clone())
    print(mode, "train-mode rel", [float((o - outs2[0]).norm() / outs2[0].norm()) for o in outs2[1:]], "norms", [float(o.norm()) for o in outs2], "nonfinite", [int((~torch.isfinite(o)).sum()) for o in outs2])
