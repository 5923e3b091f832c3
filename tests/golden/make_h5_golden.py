#!/usr/bin/env python3
"""
Writes tests/golden/h5/*.h5 with the HDF5 LIBRARY (h5py 3.3.0 / HDF5 1.10.6, found in this image under /opt/conda/bin/python3.9 --
the interpreter the package runs on has no h5py), issuing the h5py calls Keras issues when it saves a model, so that
neurite_amd.h5lite can be checked against files it did not write (neurite/tf/modelio.py:111-143 reads such files,
neurite/tf/callbacks.py:349-481 writes them).  TensorFlow / Keras themselves are not in the image: what is reproduced here is the
sequence of h5py calls of `keras.saving.hdf5_format` (TF 2.4 - 2.11), restated:
    save_weights_to_hdf5_group(f, layers):
        save_attributes_to_hdf5_group(f, 'layer_names', [layer.name.encode('utf8') ...])      # lists of bytes; split into
        f.attrs['backend'] = b'tensorflow';  f.attrs['keras_version'] = b'2.x'                 # name0, name1 .. past 64512 bytes
        per layer: g = f.create_group(layer.name); save_attributes_to_hdf5_group(g, 'weight_names', [w.name.encode('utf8') ...])
                   per weight: d = g.create_dataset(name, val.shape, dtype=val.dtype); d[()] = val  /  d[:] = val
    save_model_to_hdf5(model, f):  f.attrs['keras_version' | 'backend'], f.attrs['model_config' | 'training_config'] = json bytes,
        save_weights_to_hdf5_group(f.create_group('model_weights'), model.layers),
        optimizer: g = f.create_group('optimizer_weights'); weight_names attribute + one dataset per slot variable
Also written: files that exercise what a Keras file does not (chunked + deflate + shuffle + fletcher32 datasets, variable-length string
attributes, big-endian and integer data, a compact dataset, many links under one group) for the reader's own sake.

    python tests/golden/make_h5_golden.py            # stage 1 here, stage 2 re-invoked under the interpreter that has h5py
A re-run reproduces the committed files byte for byte (the modification times the library stamps on datasets are set to a constant
afterwards; the libver='latest' file is written with time tracking off).
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
OUT = os.path.join(HERE, 'h5')
H5PY_PYTHON = os.environ.get('H5PY_PYTHON', '/opt/conda/bin/python3.9')
HDF5_OBJECT_HEADER_LIMIT = 64512


def stage1():
    import h5_cases
    tmp = tempfile.mkdtemp()
    for tag in h5_cases.CASES:
        case, vals = h5_cases.values(tag)
        layers = [{'name': l['name'], 'weights': ['%s/%s:0' % (l['name'], v) for v in h5_cases.keras_variables(l)]} for l in case['graph']['layers']]
        config = {'class_name': case['builder'], 'config': {'args': case['args'], 'kwargs': case['kwargs']}}
        with open(os.path.join(tmp, tag + '.json'), 'w') as f:
            json.dump({'layers': layers, 'model_config': config, 'keras_version': h5_cases.KERAS_VERSION}, f)
        np.savez(os.path.join(tmp, tag + '.npz'), **{k + ':0': v for k, v in vals.items()})
    os.makedirs(OUT, exist_ok=True)
    subprocess.run([H5PY_PYTHON, os.path.abspath(__file__), '--stage2', tmp], check=True)
    # The library stamps every dataset with its modification time (message 0x0012: Keras files carry them too, the reader skips them).
    # Set the seconds to a constant so that a re-run reproduces the committed files byte for byte (version-1 headers have no checksum).
    for fn in sorted(os.listdir(OUT)):
        if fn.endswith('.h5') and fn != 'latest_small.h5':
            fix_times(os.path.join(OUT, fn))


def fix_times(path, seconds=1700000000):
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from neurite_amd import h5lite
    patches = []
    with h5lite.File(path, 'r') as f:
        def walk(node):
            for mtype, _, pos, _ in node._msgs.items:
                if mtype == 0x12 and f._buf.d[pos] == 1:
                    patches.append(pos + 4)
            if isinstance(node, h5lite.Group):
                for k in node.keys():
                    walk(node[k])
        walk(f)
    with open(path, 'r+b') as fh:
        for pos in patches:
            fh.seek(pos)
            fh.write(int(seconds).to_bytes(4, 'little'))
    return len(patches)


# ---- stage 2: runs under the interpreter with h5py --------------------------------------------------------------------------------

def save_attributes_to_hdf5_group(group, name, data):
    bad = [x for x in data if len(x) > HDF5_OBJECT_HEADER_LIMIT]
    if bad:
        raise RuntimeError('attribute entries too long: %s' % bad)
    data_npy = np.asarray(data)
    num_chunks = 1
    chunked = np.array_split(data_npy, num_chunks)
    while any(x.nbytes > HDF5_OBJECT_HEADER_LIMIT for x in chunked):
        num_chunks += 1
        chunked = np.array_split(data_npy, num_chunks)
    if num_chunks > 1:
        for i, c in enumerate(chunked):
            group.attrs['%s%d' % (name, i)] = c
    else:
        group.attrs[name] = data


def save_weights_to_hdf5_group(f, layers, arrays, keras_version, sort_layers=False):
    save_attributes_to_hdf5_group(f, 'layer_names', [l['name'].encode('utf8') for l in layers])
    f.attrs['backend'] = 'tensorflow'.encode('utf8')
    f.attrs['keras_version'] = str(keras_version).encode('utf8')
    for l in (sorted(layers, key=lambda x: x['name']) if sort_layers else layers):
        g = f.create_group(l['name'])
        weight_names = [w.encode('utf8') for w in l['weights']]
        save_attributes_to_hdf5_group(g, 'weight_names', weight_names)
        for name in weight_names:
            val = arrays[name.decode('utf8')]
            d = g.create_dataset(name, val.shape, dtype=val.dtype)
            if not val.shape:
                d[()] = val
            else:
                d[:] = val


def stage2(tmp):
    import h5py
    for fn in sorted(os.listdir(tmp)):
        if not fn.endswith('.json'):
            continue
        tag = fn[:-5]
        meta = json.load(open(os.path.join(tmp, fn)))
        arrays = dict(np.load(os.path.join(tmp, tag + '.npz')))
        # model.save_weights('x.h5')
        with h5py.File(os.path.join(OUT, tag + '_weights.h5'), 'w') as f:
            save_weights_to_hdf5_group(f, meta['layers'], arrays, meta['keras_version'])
        # model.save('x.h5') with a compiled model (Adam: iteration counter + two slots per variable)
        with h5py.File(os.path.join(OUT, tag + '_model.h5'), 'w') as f:
            f.attrs['keras_version'] = str(meta['keras_version']).encode('utf8')
            f.attrs['backend'] = 'tensorflow'.encode('utf8')
            f.attrs['model_config'] = json.dumps(meta['model_config']).encode('utf8')
            f.attrs['training_config'] = json.dumps({'loss': 'dice', 'metrics': None, 'optimizer_config': {'class_name': 'Adam', 'config': {'learning_rate': 1e-4}}}).encode('utf8')
            save_weights_to_hdf5_group(f.create_group('model_weights'), meta['layers'], arrays, meta['keras_version'], sort_layers=True)
            g = f.create_group('optimizer_weights')
            names, vals = ['Adam/iter:0'], [np.array(12345, dtype=np.int64)]
            for l in meta['layers']:
                for w in l['weights']:
                    for slot in ('m', 'v'):
                        names.append('Adam/%s/%s:0' % (w[:-2], slot))
                        vals.append(np.full(arrays[w].shape, 0.5 if slot == 'm' else 0.25, np.float32))
            save_attributes_to_hdf5_group(g, 'weight_names', [n.encode('utf8') for n in names])
            for n, v in zip(names, vals):
                d = g.create_dataset(n, v.shape, dtype=v.dtype)
                if not v.shape:
                    d[()] = v
                else:
                    d[:] = v
    # ---- beyond what Keras writes
    rng = np.random.default_rng(11)
    a = rng.standard_normal((37, 21, 5)).astype(np.float32)
    with h5py.File(os.path.join(OUT, 'storage_forms.h5'), 'w') as f:
        f.create_dataset('contiguous', data=a)
        f.create_dataset('chunked', data=a, chunks=(8, 8, 5))
        f.create_dataset('gzip_shuffle', data=a, chunks=(16, 7, 5), compression='gzip', compression_opts=4, shuffle=True)
        f.create_dataset('gzip_fletcher', data=a.astype(np.float64), chunks=(10, 21, 2), compression='gzip', fletcher32=True)
        f.create_dataset('big_endian', data=a.astype('>f4'))
        f.create_dataset('int16', data=(a * 100).astype(np.int16))
        f.create_dataset('uint8', data=(np.abs(a) * 20).astype(np.uint8), chunks=True)
        f.create_dataset('float16', data=a.astype(np.float16))
        f.create_dataset('scalar', data=np.float64(2.5))
        f.create_dataset('empty', shape=(0, 3), dtype=np.float32)
        f.create_dataset('never_written', shape=(4, 3), dtype=np.float32)
        f.create_dataset('bool', data=a > 0)
        f.create_dataset('strings_fixed', data=np.array([b'alpha', b'be', b'gamma!'], dtype='S6'))
        f.create_dataset('strings_vlen', data=np.array(['one', 'zwölf', ''], dtype=object), dtype=h5py.string_dtype('utf-8'))
        # compact layout through the low-level API
        dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
        dcpl.set_layout(h5py.h5d.COMPACT)
        small = np.arange(12, dtype=np.int32).reshape(3, 4)
        sid = h5py.h5s.create_simple(small.shape)
        did = h5py.h5d.create(f.id, b'compact', h5py.h5t.NATIVE_INT32, sid, dcpl=dcpl)
        did.write(h5py.h5s.ALL, h5py.h5s.ALL, small)
        f.attrs['str_scalar'] = 'variable-length ünicode'
        f.attrs['bytes_scalar'] = b'fixed bytes'
        f.attrs['str_list'] = ['a', 'bb', 'ccc']
        f.attrs['int'] = 7
        f.attrs['float_array'] = np.arange(5, dtype=np.float64) / 4
        f.attrs['bool'] = True
        f.attrs['empty_list'] = []
        f.attrs['empty'] = h5py.Empty('f')
        f.attrs['big_json'] = json.dumps({'layers': [{'name': 'layer_%05d' % i, 'config': {'filters': i, 'note': 'x' * 40}} for i in range(2000)]}).encode('utf8')   # > 64 KB: a variable-length string in the global heap, as a large model_config is
        f['chunked'].attrs['note'] = b'attribute on a dataset'
        g = f.create_group('nested/deeper/deepest')
        g.attrs['depth'] = 3
        g.create_dataset('x', data=np.arange(3))
    # many links under one group (a multi-level group B-tree in the library's file) and a chunked `layer_names` attribute
    with h5py.File(os.path.join(OUT, 'many_layers.h5'), 'w') as f:
        layers = [{'name': 'layer_with_a_long_name_to_fill_the_object_header_%s_%04d' % ('x' * 40, i), 'weights': []} for i in range(700)]
        arrays = {}
        for i in (0, 350, 699):
            n = layers[i]['name'] + '/kernel:0'
            layers[i]['weights'] = [n]
            arrays[n] = np.full((2, 2), float(i), np.float32)
        save_weights_to_hdf5_group(f, layers, arrays, '2.4.0')
    # the same small tree with libver='latest' (version-2 object headers, compact link messages).  Version-2 headers carry four time stamps
    # under a checksum: times are switched off on every object (low-level property lists) so that the file is reproducible
    fapl = h5py.h5p.create(h5py.h5p.FILE_ACCESS)
    fapl.set_libver_bounds(h5py.h5f.LIBVER_LATEST, h5py.h5f.LIBVER_LATEST)
    fcpl = h5py.h5p.create(h5py.h5p.FILE_CREATE)
    fcpl.set_obj_track_times(False)
    fid = h5py.h5f.create(os.path.join(OUT, 'latest_small.h5').encode(), h5py.h5f.ACC_TRUNC, fcpl=fcpl, fapl=fapl)
    with h5py.File(fid) as f:
        f.attrs['layer_names'] = [b'a', b'b']
        for n in ('a', 'b'):
            gcpl = h5py.h5p.create(h5py.h5p.GROUP_CREATE)
            gcpl.set_obj_track_times(False)
            g = h5py.Group(h5py.h5g.create(f.id, n.encode(), gcpl=gcpl))
            g.attrs['weight_names'] = [('%s/kernel:0' % n).encode()]
            gcpl2 = h5py.h5p.create(h5py.h5p.GROUP_CREATE)
            gcpl2.set_obj_track_times(False)
            h5py.h5g.create(g.id, n.encode(), gcpl=gcpl2)
            g.create_dataset('%s/kernel:0' % n, data=np.arange(6, dtype=np.float32).reshape(2, 3) + ord(n), track_times=False)
    print('h5py', h5py.__version__, 'HDF5', h5py.version.hdf5_version, '->', sorted(os.listdir(OUT)))


if __name__ == '__main__':
    if '--stage2' in sys.argv:
        stage2(sys.argv[sys.argv.index('--stage2') + 1])
    else:
        stage1()
