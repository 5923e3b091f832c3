"""unet (BASELINE config 3) training step on one GPU: forward + backward (CCE + Dice loss) + SGD update, ms per step."""
import contextlib, io, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurite_amd as ne

dev = torch.device('cuda:0')
size = int(sys.argv[1]) if len(sys.argv) > 1 else 160
L = 32
with contextlib.redirect_stdout(io.StringIO()):
    net = ne.models.unet(16, (size, size, size, 1), 3, 3, L, feat_mult=2).to(dev)
x = torch.randn(1, size, size, size, 1, device=dev)
t = torch.nn.functional.one_hot(torch.randint(0, L, (1, size, size, size), device=dev), L).float()
cce = ne.losses.CategoricalCrossentropy()
dice = ne.metrics.Dice(check_input_limits=False)
params = [p for p in net.parameters()]


def fwd_only():
    with torch.no_grad():
        return net(x)


def step():
    y = net(x)
    loss = cce(t, y) - dice.mean_dice(t, y)
    loss.backward()
    with torch.no_grad():
        for p in params:
            p -= 1e-4 * p.grad
            p.grad = None
    return loss


def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


net.eval()
res = {'size': size, 'fwd_eval_ms': round(timeit(fwd_only), 3)}
net.train()
l0 = float(step().detach())
res['train_step_ms'] = round(timeit(step), 3)
l1 = float(step().detach())
res['loss_first'], res['loss_after'] = round(l0, 5), round(l1, 5)
res['peak_mem_GB'] = round(torch.cuda.max_memory_allocated() / 1e9, 2)
gf = 283.8 * (size / 160) ** 3
res['GFLOP_fwd'] = round(gf, 1)
res['TFLOPs_train_3x'] = round(3 * gf / res['train_step_ms'], 1)
print(json.dumps(res))
