"""rocprofv3 target: kernel breakdown of labels_to_image_new at 160^3 (run under rocprofv3 --kernel-trace --stats)."""
import json, torch, warnings
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
S, B = 160, 2
lab = synth.one_hot_volume(1, S, 32, dev).argmax(-1)[None, ..., None].to(torch.float32).repeat(B, 1, 1, 1, 1)
gen = ne.models.labels_to_image_new(list(range(32)), in_shape=(S, S, S), aff_shift=10, aff_rotate=10, aff_scale=0.1, aff_shear=0.05)
for _ in range(2): gen(lab)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(5): gen(lab)
e1.record(); torch.cuda.synchronize()
print(json.dumps({'labels_to_image_new_ms_per_call_B2': e0.elapsed_time(e1) / 5}))
