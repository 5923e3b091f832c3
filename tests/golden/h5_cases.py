"""What the Keras-layout HDF5 fixtures under tests/golden/h5/ hold (shared by make_h5_golden.py, which writes them with the HDF5
library, and tests/test_h5lite.py, which reads them back with neurite_amd.h5lite): for a recorded reference graph
(tests/golden/unet_graph.json) the list of Keras layers in model order with the variables each owns and seeded values of the
shapes Keras stores."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ('bn_dropout_res', 'two_d_pool')
KERAS_VERSION = '2.4.0'            # written into the fixtures' `keras_version` attribute (tf.keras of TF 2.4 wrote this)


def keras_variables(layer):
    """variables of a Keras layer in `layer.weights` order (tests/test_unet_graph.py::_keras_variables)"""
    if layer['class'] in ('Conv1D', 'Conv2D', 'Conv3D'):
        return ['kernel'] + (['bias'] if layer['config'].get('use_bias', True) else [])
    if layer['class'] == 'BatchNormalization':
        return ['gamma', 'beta', 'moving_mean', 'moving_variance']
    return []


def graph(tag):
    with open(os.path.join(HERE, 'unet_graph.json')) as f:
        return json.load(f)[tag]


def variable_shapes(case):
    """{layer: [(variable, shape)]} from the recorded constructor arguments alone (no network is built)"""
    layers = {l['name']: l for l in case['graph']['layers']}
    out = {}
    for l in case['graph']['layers']:
        vs = keras_variables(l)
        if not vs:
            continue
        cin = sum(layers[i]['output_shape'][-1] for i in l['inputs'])
        if l['class'].startswith('Conv'):
            shapes = {'kernel': tuple(l['config']['kernel_size']) + (cin, l['config']['filters']), 'bias': (l['config']['filters'],)}
        else:
            shapes = {v: (cin,) for v in vs}
        out[l['name']] = [(v, shapes[v]) for v in vs]
    return out


def values(tag, seed=7):
    """seeded float32 values per `layer/variable`, in model order"""
    case = graph(tag)
    rng = np.random.default_rng(seed)
    vals = {}
    for layer, vs in variable_shapes(case).items():
        for v, shape in vs:
            vals['%s/%s' % (layer, v)] = rng.standard_normal(shape).astype(np.float32)
    return case, vals
