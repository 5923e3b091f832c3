#!/usr/bin/env python3
"""TensorFlow-side exporter: a Keras model's weights -> the .npz archive `neurite_amd.models.*.load_weights` reads.

Run where the reference runs (TensorFlow / Keras installed; NOT needed on the MI355X side, and h5py is needed on neither side):

    python tools/export_keras_weights.py model_or_weights.h5 out.npz [--builder unet --args '[16, [160,160,160,1], 3, 3, 32]' --kwargs '{"feat_mult": 2}']

Every variable is stored under `layer/variable` with the variable's own Keras name (`kernel`, `bias`, `gamma`, `beta`,
`moving_mean`, `moving_variance`): exactly the keys, in the order, that the network built by the same arguments here expects
(tests/test_unet_graph.py::test_npz_keys_are_the_keras_variable_names pins that for the 20 recorded reference graphs).
The reference's own loaders this replaces: neurite/tf/modelio.py:111-143 (LoadableModel.load: h5py -> model_config -> load_weights).
"""
import json
import sys


def export(model, path):
    import numpy as np
    arrays = {}
    for layer in model.layers:
        for var, value in zip(layer.weights, layer.get_weights()):
            # Keras 2: 'conv/kernel:0' (sometimes prefixed by the model name), Keras 3: 'kernel' -- keep the last path component
            name = var.name.split('/')[-1].split(':')[0]
            arrays['%s/%s' % (layer.name, name)] = np.asarray(value)
    np.savez(path, **arrays)
    return list(arrays)


def main(argv):
    if len(argv) < 3:
        sys.exit(__doc__)
    src, dst = argv[1], argv[2]
    import tensorflow as tf                                                  # noqa: F401  (only here)
    if '--builder' in argv:
        import neurite as ne
        builder = argv[argv.index('--builder') + 1]
        args = json.loads(argv[argv.index('--args') + 1]) if '--args' in argv else []
        kwargs = json.loads(argv[argv.index('--kwargs') + 1]) if '--kwargs' in argv else {}
        model = getattr(ne.models, builder)(*args, **kwargs)
        model.load_weights(src)
    else:
        model = tf.keras.models.load_model(src, compile=False)
    print('\n'.join(export(model, dst)))


if __name__ == '__main__':
    main(sys.argv)
