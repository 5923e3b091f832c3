"""
The layer graphs `neurite_amd.models` builds against the graphs the REFERENCE's own builders construct.

tests/golden/unet_graph.json is produced by tests/golden/make_golden.py::gen_unet_graph: it runs neurite/tf/models.py
(`unet` :88-246, `conv_enc` :1309-1442, `conv_dec` :1445-1617, `add_prior` :378-436, `dilation_net` :45-85) on recording
stand-ins for the Keras layer constructors (tests/golden/keras_record.py) and stores, per layer: name, Keras class,
constructor arguments (filters, kernel, dilation, activation, padding, pool / up-sampling size, dropout rate and noise
shape, batch-norm axis), the producers of its inputs and the output shape.  CPU only; no kernels run.
"""

import contextlib
import io
import json
import os
import warnings

import pytest

from neurite_amd import models

HERE = os.path.dirname(os.path.abspath(__file__))

with open(os.path.join(HERE, 'golden', 'unet_graph.json')) as f:
    GRAPHS = json.load(f)


def _canon(graph):
    """Keras numbers unnamed Dropout layers by creation order (including ones that end up outside the model), so a
    dropout is identified by the layer it follows instead."""
    ren = {}
    for l in graph['layers']:
        if l['class'] == 'Dropout':
            ren[l['name']] = 'dropout@' + l['inputs'][0]
    # chains of dropouts: resolve iteratively
    def r(n):
        while n in ren and ren[n] != n:
            m = ren[n]
            if m.startswith('dropout@') and m[8:] in ren:
                m = 'dropout@' + r(m[8:])
                ren[n] = m
            return m
        return n
    out = {}
    for l in graph['layers']:
        d = dict(l)
        d['name'] = r(l['name'])
        d['inputs'] = [r(i) for i in l['inputs']]
        assert d['name'] not in out, 'duplicate layer ' + d['name']
        out[d['name']] = d
    return out, [r(n) for n in graph['inputs']], [r(n) for n in graph['outputs']], [r(l['name']) for l in graph['layers']]


def _build(case):
    kwargs = dict(case['kwargs'])
    with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        warnings.simplefilter('ignore')
        return getattr(models, case['builder'])(*case['args'], **kwargs)


@pytest.mark.parametrize('tag', sorted(GRAPHS))
def test_graph_matches_reference_builder(tag):
    case = GRAPHS[tag]
    ref, ref_in, ref_out, ref_order = _canon(case['graph'])
    net = _build(case)
    got, got_in, got_out, got_order = _canon(net.keras_graph())
    assert got_in == ref_in
    assert got_out == ref_out
    assert sorted(got) == sorted(ref), 'layer sets differ: only ours %s, only reference %s' % (
        sorted(set(got) - set(ref)), sorted(set(ref) - set(got)))
    for name in ref_order:
        g, r = got[name], ref[name]
        assert g['class'] == r['class'], name
        assert g['inputs'] == r['inputs'], name
        assert g['output_shape'] == r['output_shape'], name
        assert g['config'] == r['config'], (name, g['config'], r['config'])
    # weights are exchanged as ordered lists (Keras get_weights order = layer order): the weighted layers must come in the
    # same order
    weighted = ('Conv1D', 'Conv2D', 'Conv3D', 'BatchNormalization')
    assert [n for n in got_order if got[n]['class'] in weighted] == [n for n in ref_order if ref[n]['class'] in weighted]


def test_residual_tail_conv_has_no_dilation_and_no_activation():
    """the case round 1 got wrong (neurite/tf/models.py:1384-1388, 1552-1555)"""
    net = _build(GRAPHS['res_dil'])
    g = {l['name']: l for l in net.keras_graph()['layers']}
    assert g['unet_conv_downarm_1_0']['config']['dilation_rate'] == [2, 2, 2]
    assert g['unet_conv_downarm_1_1']['config']['dilation_rate'] == [1, 1, 1]
    assert g['unet_conv_downarm_1_1']['config']['activation'] == 'linear'
    assert g['unet_conv_uparm_3_1']['config']['dilation_rate'] == [1, 1, 1]
    assert g['unet_expand_down_merge_2']['config']['dilation_rate'] == [4, 4, 4]


# variables a Keras layer owns, in `layer.weights` order (Conv*: kernel[, bias]; BatchNormalization with the reference's defaults
# center = scale = True: gamma, beta, moving_mean, moving_variance)
def _keras_variables(layer):
    if layer['class'] in ('Conv1D', 'Conv2D', 'Conv3D'):
        return ['kernel'] + (['bias'] if layer['config'].get('use_bias', True) else [])
    if layer['class'] == 'BatchNormalization':
        return ['gamma', 'beta', 'moving_mean', 'moving_variance']
    return []


@pytest.mark.parametrize('tag', sorted(GRAPHS))
def test_npz_keys_are_the_keras_variable_names(tag, tmp_path):
    """f-3 without h5py: what can be pinned of the Keras weight import.  tools/export_keras_weights.py stores
    `model.get_weights()` under `layer.name/variable` on the TensorFlow side; the archive our `load_weights` wants must have exactly
    those keys in exactly that order for every graph recorded from the reference's builders (neurite/tf/modelio.py:111-143 is what the
    pair replaces).  A save -> load round trip through an archive WRITTEN UNDER THE RECORDED NAMES closes the loop on this side."""
    import numpy as np
    case = GRAPHS[tag]
    expected = ['%s/%s' % (l['name'], v) for l in case['graph']['layers'] for v in _keras_variables(l)]
    net = _build(case)
    ours = [n for n, _, _ in net._weight_tensors()]
    assert ours == expected
    # an archive keyed by the recorded Keras names, filled with recognisable values, loads into the network slot by slot
    rng = np.random.default_rng(3)
    arrays = {}
    for n, t, nd in net._weight_tensors():
        shape = tuple(t.shape)
        if nd is not None and nd < 3:
            shape = shape[3 - nd:]                     # Conv1D / Conv2D kernels are stored with their own rank in Keras
        arrays[n] = rng.standard_normal(shape).astype(np.float32)
    path = str(tmp_path / 'keras_export.npz')
    np.savez(path, **{k: arrays[k] for k in expected})
    net.load_weights(path)
    for (n, t, nd), w in zip(net._weight_tensors(), net.get_weights()):
        np.testing.assert_array_equal(np.asarray(w).reshape(arrays[n].shape), arrays[n], err_msg=n)
