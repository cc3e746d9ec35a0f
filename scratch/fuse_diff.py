import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_to_midi_amd as amt
from audio_to_midi_amd import synth, params
from oracle import np_model
conf = dict(np_model.MODEL_CONFIG); conf["depths"] = [1, 1, 2, 1, 1, 2, 1]; conf["num_transformer_layers"] = 2
p = np_model.init_params(conf, seed=21, gamma=1.0)
audio, labels = synth.make_batch(3, 320 * 77, seed=9)
outs = []
for fuse in ("1", "0"):
    os.environ["AMT_FUSE"] = fuse
    m = amt.OutputSequenceGenerator(conf, dtype=sys.argv[1] if len(sys.argv) > 1 else "bfloat16")
    m.load_equinox_params(p)
    outs.append(m.loss_and_grads(torch.from_numpy(audio), torch.from_numpy(labels), seed=5, train=(len(sys.argv) <= 2)).grads_flat.clone())
    m.close()
d = outs[0] - outs[1]
print("total rel", float(d.norm() / outs[1].norm()))
for s in params.device_layout(conf):
    n = s.rows * s.cols
    dd = d[s.offset:s.offset + n]; g = outs[1][s.offset:s.offset + n]
    if s.name.startswith("tl.") or s.name.startswith("decoder") or "layers.6" in s.name:
        print("   %-40s rel %.3g" % (s.name, float(dd.norm()) / (float(g.norm()) + 1e-30)))
