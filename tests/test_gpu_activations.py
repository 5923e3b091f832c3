"""
Activations of the conv / LocallyConnected3D epilogues and of the stand-alone Activation layers.  The reference hands `activation`
straight to Keras (neurite/tf/models.py:1346, 1429, 1507, 1588; layers.py:1101), so any Keras activation string may arrive:
every element-wise one of csrc/activations.h and the channel softmax, forward and backward, against torch (float64 autograd on
the CPU) of the same tf.keras.activations definitions (oracle/torch_unet_oracle.keras_activation).
"""

import numpy as np
import pytest
import torch

import neurite_amd as ne
from oracle import torch_unet_oracle as tuo

pytestmark = pytest.mark.gpu
ACTS = ['linear', 'elu', 'relu', 'sigmoid', 'tanh', 'softplus', 'softsign', 'selu', 'exponential', 'hard_sigmoid', 'leaky_relu', 'softmax']


def close(got, want, tol, what):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    err = float((got - want).abs().max())
    assert err <= tol * max(1.0, float(want.abs().max())), (what, err)


@pytest.mark.parametrize('act', ACTS)
def test_conv_layer_activation_forward_backward(dev, act):
    torch.manual_seed(5)
    from neurite_amd.models import _Conv
    for cin, cout, k, shape in ((3, 16, 3, (6, 7, 9)), (16, 8, 3, (5, 8, 8)), (4, 5, 1, (4, 5, 6))):
        m = _Conv('c', cin, cout, (k, k, k), activation=act).to(dev)
        with torch.no_grad():
            m.bias.normal_(0, 0.3)
        x = (torch.randn(2, *shape, cin, device=dev) * 0.8).requires_grad_()
        y = m(x)
        xr = x.detach().cpu().double().requires_grad_()
        kr, br = m.kernel.detach().cpu().double().requires_grad_(), m.bias.detach().cpu().double().requires_grad_()
        yr = tuo.conv3d_same(xr, kr, br, 1, act)
        close(y, yr, 2e-5, ('forward', act, cin, cout))
        with torch.no_grad():
            close(m(x.detach()), yr, 2e-5, ('inference path', act))
        g = torch.randn_like(y)
        y.backward(g)
        yr.backward(g.cpu().double())
        close(x.grad, xr.grad, 2e-4, ('dx', act))
        close(m.kernel.grad, kr.grad, 2e-4, ('dkernel', act))
        close(m.bias.grad, br.grad, 2e-4, ('dbias', act))


@pytest.mark.parametrize('act', ACTS)
def test_lc3d_activation_forward_backward(dev, act):
    torch.manual_seed(6)
    for dtype, tol in ((torch.float32, 2e-5), (torch.bfloat16, 2e-2)):
        lin = ne.layers.LocallyConnected3D(8, (3, 2, 3), activation=None).to(dev)
        lay = ne.layers.LocallyConnected3D(8, (3, 2, 3), activation=act).to(dev)
        x = (torch.randn(2, 6, 5, 7, 4, device=dev) * 0.7).to(dtype)
        lin(x), lay(x)                                                     # build
        with torch.no_grad():
            lin.bias.normal_(0, 0.2)
            lay.kernel.copy_(lin.kernel)
            lay.bias.copy_(lin.bias)
        xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
        ya = lay(xa)
        yb = tuo.keras_activation(lin(xb).float(), act)
        close(ya.float(), yb, tol, ('lc3d forward', act, dtype))
        assert ya.dtype == dtype
        if dtype == torch.float32:
            g = torch.randn_like(ya)
            ya.backward(g)
            yb.backward(g)
            close(xa.grad, xb.grad, 2e-4, ('lc3d dx', act))
            close(lay.kernel.grad, lin.kernel.grad, 2e-4, ('lc3d dkernel', act))
            close(lay.bias.grad, lin.bias.grad, 2e-4, ('lc3d dbias', act))
    assert lay.get_config()['activation'] == act
    with pytest.raises(NotImplementedError):
        ne.layers.LocallyConnected3D(4, 3, activation='no_such_activation')


@pytest.mark.parametrize('act,final', [('tanh', 'sigmoid'), ('leaky_relu', 'softmax'), ('selu', None), ('softplus', 'hard_sigmoid')])
def test_unet_with_keras_activations(dev, act, final):
    """models.unet(activation=..., final_pred_activation=...) as the reference builds it (conv activations, the Activation layer of
    the residual form, the prediction activation), forward and one gradient, against the torch graph interpreter"""
    torch.manual_seed(7)
    for kw in (dict(), dict(use_residuals=True, nb_conv_per_level=2)):
        model = ne.models.unet(4, (12, 10, 8, 2), 2, 3, 5, feat_mult=2, activation=act, final_pred_activation=final, **kw).to(dev)
        x = torch.randn(2, 12, 10, 8, 2, device=dev)
        with torch.no_grad():
            y = model(x)
        want = tuo.forward(model, x.cpu().double())
        close(y, want, 5e-5, ('unet', act, final, kw))
        model.train()
        loss = model(x).square().mean()
        loss.backward()
        params = [p for p in model.parameters() if p.grad is not None]
        assert params and all(torch.isfinite(p.grad).all() for p in params)
    with pytest.raises(NotImplementedError):
        ne.models.unet(4, (12, 10, 8, 2), 2, 3, 5, activation='gelu_like_thing')
