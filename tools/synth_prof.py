import json, torch, warnings
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
S, B = 160, 2
lab = synth.one_hot_volume(1, S, 32, dev).argmax(-1)[None, ..., None].to(torch.int32).repeat(B, 1, 1, 1, 1)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    gen = ne.models.labels_to_image((S, S, S), list(range(32)))
for _ in range(2): gen(lab)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(5): gen(lab)
e1.record(); torch.cuda.synchronize()
print(json.dumps({'labels_to_image_ms_per_call_B2': e0.elapsed_time(e1) / 5}))
