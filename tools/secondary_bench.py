"""Secondary entries of the metric rows at the headline shape, 4 x 160^3: ms and fraction of the HBM roof
of their algorithmic bytes -- a sweep for outliers (the min-max reduction of round 1 hid 2000 same-address atomics)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def row(op, ms, nbytes):
    print(json.dumps({'op': op, 'ms': round(ms, 4), 'GBs': round(nbytes / ms / 1e6, 1), 'frac': round(nbytes / ms / 1e6 / 8000, 3)}), flush=True)

S, B, L = 160, 4, 32
mov, fix, trf = synth.cfg2_batch(B, S, L, dev)
nvox = B * S ** 3
lt, lp = mov.argmax(-1).to(torch.int32), fix.argmax(-1).to(torch.int32)
row('soft Dice [4,160^3,32]', timeit(lambda: ne.metrics.Dice().dice(fix, mov)), nvox * 256)
row('soft Dice, check_input_limits=False', timeit(lambda: ne.metrics.Dice(check_input_limits=False).dice(fix, mov)), nvox * 256)
row('soft Dice normalize=True', timeit(lambda: ne.metrics.Dice(normalize=True).dice(fix, mov)), nvox * 256)
row('hard Dice from probabilities (argmax)', timeit(lambda: ne.metrics.HardDice(L, input_type='prob').dice(fix, mov)), nvox * 256)
row('hard Dice from int32 label maps', timeit(lambda: ne.metrics.HardDice(L, input_type='max_label').dice(lt, lp)), nvox * 8)
blobs = synth.one_hot_volume(5, S, 8, dev).argmax(-1).to(torch.int32)[None].repeat(B, 1, 1, 1)
row('hard Dice from int32 label maps, 8 labels', timeit(lambda: ne.metrics.HardDice(8, input_type='max_label').dice(blobs, blobs)), nvox * 8)
m20, f20 = mov[..., :20].contiguous(), fix[..., :20].contiguous()
with __import__('warnings').catch_warnings():
    __import__('warnings').simplefilter('ignore')
    row('hard Dice from probabilities, 20 labels (lane-groups of 8, 5 real)', timeit(lambda: ne.metrics.HardDice(20, input_type='prob').dice(f20, m20), n=3), nvox * 160)
row('soft Dice, 20 labels (lane-groups of 8, 5 real)', timeit(lambda: ne.metrics.Dice().dice(f20, m20)), nvox * 160)
w = torch.rand(L, device=dev) + 0.5
p = torch.softmax(torch.randn(B, S, S, S, L, device=dev), -1)
row('weighted CCE [4,160^3,32]', timeit(lambda: ne.metrics.CategoricalCrossentropy(label_weights=w)(fix, p)), nvox * 256)
p20 = p[..., :20].contiguous()
row('weighted CCE, 20 labels (lane-groups of 8, 5 real)', timeit(lambda: ne.metrics.CategoricalCrossentropy(label_weights=w[:20])(f20, p20)), nvox * 160)
del p20
logits = torch.randn(B, S, S, S, L, device=dev)
row('weighted CCE from logits (softmax in registers, one pass) [4,160^3,32]',
    timeit(lambda: ne.metrics.CategoricalCrossentropy(label_weights=w, from_logits=True)(fix, logits)), nvox * 256)
lg5, t5 = torch.randn(1, 96, 96, 96, 16, device=dev), torch.softmax(torch.randn(1, 96, 96, 96, 16, device=dev), -1)
w5 = torch.rand(16, device=dev) + 0.5
row('config 5 shape: weighted CCE from logits [1,96^3,16] (launch-bound: 113 MB)',
    timeit(lambda: ne.metrics.CategoricalCrossentropy(label_weights=w5, from_logits=True)(t5, lg5), n=30), 96 ** 3 * 128)
row('config 5 shape: softmax then weighted CCE (two kernels) [1,96^3,16]',
    timeit(lambda: ne.metrics.CategoricalCrossentropy(label_weights=w5)(t5, torch.softmax(lg5, -1)), n=30), 96 ** 3 * 256)
row('MeanSquaredErrorProb', timeit(lambda: ne.metrics.MeanSquaredErrorProb()(fix, p)), nvox * 256)
row('mean_dice (weights [1, L])', timeit(lambda: ne.metrics.Dice(weights=w[None]).mean_dice(fix, mov)), nvox * 256)
st = ne.layers.SpatialTransformer(interp_method='nearest')
row('nearest warp [4,160^3,32]', timeit(lambda: st([mov, trf])), nvox * 268)
lab1 = lt[..., None].to(torch.float32)
# the output stage of labels_to_image as the product runs it (neurite/tf/models.py:806-807 + one_hot): nearest warp of the label map
# (4 + 12 B read, 4 B written per voxel), then the LUT + one-hot kernel (4 B read, 4 L written)
from neurite_amd import _lib
lut = torch.arange(L, dtype=torch.int32, device=dev)
oh = torch.empty((B, S, S, S, L), dtype=torch.float32, device=dev)
def warp_onehot():
    idx = st([lab1, trf])
    rc = _lib.lib().nrt_synth_labels_out(_lib.ptr(idx), _lib.ptr(lut), L, L, _lib.ptr(oh), None, nvox, _lib.stream_ptr(dev))
    assert rc == 0
row('nearest warp of a label map + LUT / one-hot (labels_to_image output stage)', timeit(warp_onehot), nvox * (4 + 12 + 4 + 4 + 4 * L))
# the segmentation loss pair (neurite/tf/losses.py:225-246) on one prediction: one joint pass each way (csrc/segloss.hip) against the
# two losses on their own; algorithmic bytes: forward 2 maps read once (8 L per voxel), forward + backward + 4 L written
del logits, oh
cce_o, dice_o = ne.losses.CategoricalCrossentropy(label_weights=w), ne.losses.Dice(check_input_limits=False)
joint = ne.losses.multiple_losses_decorator([cce_o.loss, dice_o.loss])
row('Dice + weighted CCE of one prediction, joint forward [4,160^3,32]', timeit(lambda: joint(fix, p)), nvox * 256)
row('Dice + weighted CCE, the two losses separately (forward)', timeit(lambda: cce_o.loss(fix, p) + dice_o.loss(fix, p)), nvox * 256)
pg = p.clone().requires_grad_()
def fb(fn):
    pg.grad = None
    fn(fix, pg).mean().backward()
row('Dice + weighted CCE, joint forward + backward wrt the prediction', timeit(lambda: fb(joint)), nvox * (256 + 256 + 128))
row('Dice + weighted CCE, separately, forward + backward', timeit(lambda: fb(lambda a, b: cce_o.loss(a, b) + dice_o.loss(a, b))), nvox * (256 + 256 + 128))
