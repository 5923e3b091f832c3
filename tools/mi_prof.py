"""MutualInformation.volumes forward + backward of two 4 x 160^3 images: the script rocprofv3 profiles"""
import contextlib, io, sys, torch
sys.path.insert(0, '.')
import neurite_amd as ne
dev = torch.device('cuda:0')
img = torch.randn(4, 160, 160, 160, 1, device=dev)
ya = img * 0.7 + 0.3 * torch.randn_like(img)
with contextlib.redirect_stdout(io.StringIO()):
    mi = ne.metrics.MutualInformation(nb_bins=16)
xg = img.clone().requires_grad_()
for _ in range(10):
    xg.grad = None
    (-mi.volumes(xg, ya).sum()).backward()
torch.cuda.synchronize()
