"""labels_to_image / labels_to_image_new at 160^3, 32 labels, batch 4 -- the script tools/gpu_session-style rocprofv3 runs profile"""
import sys, warnings, torch
sys.path.insert(0, '.')
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
S, B = 160, 4
lab = synth.one_hot_volume(1, S, 32, dev).argmax(-1)[None, ..., None].to(torch.int32).repeat(B, 1, 1, 1, 1)
which = sys.argv[1] if len(sys.argv) > 1 else 'new'
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    if which == 'new':
        gen = ne.models.labels_to_image_new(list(range(32)), in_shape=(S, S, S), aff_shift=10, aff_rotate=10, aff_scale=0.1, aff_shear=0.05)
        x = lab.to(torch.float32)
    else:
        gen = ne.models.labels_to_image((S, S, S), list(range(32)))
        x = lab
for _ in range(8):
    gen(x)
torch.cuda.synchronize()
