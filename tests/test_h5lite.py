"""
neurite_amd.h5lite (the package's HDF5 reader / writer) against the HDF5 LIBRARY, CPU only.

tests/golden/h5/*.h5 were written by h5py 3.3.0 / HDF5 1.10.6 with the h5py calls Keras makes in `save_weights` and `model.save`
(tests/golden/make_h5_golden.py; values: tests/golden/h5_cases.py from the graphs recorded off the reference's builders).  Reading:
every dataset and attribute of those files through h5lite, then through `ConvNet.load_weights` / `models.load_config` -- the calls that
replace neurite/tf/modelio.py:111-143.  Writing: what h5lite writes is read back by h5lite and, when an interpreter with h5py exists in
the image (/opt/conda/bin/python3.9 here and on the GPU box), by h5py and h5dump.
"""
import contextlib
import io
import json
import os
import subprocess
import sys
import warnings

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import h5_cases                       # noqa: E402
from neurite_amd import h5lite, models     # noqa: E402

H5 = os.path.join(HERE, 'golden', 'h5')
H5PY_PYTHON = os.environ.get('H5PY_PYTHON', '/opt/conda/bin/python3.9')


def _have_h5py():
    if not os.path.exists(H5PY_PYTHON):
        return False
    try:
        return subprocess.run([H5PY_PYTHON, '-c', 'import h5py'], capture_output=True, timeout=120).returncode == 0
    except Exception:      # noqa
        return False


def _build(case):
    with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        warnings.simplefilter('ignore')
        return getattr(models, case['builder'])(*case['args'], **dict(case['kwargs']))


def _expected_in_slots(net, vals):
    out = []
    for (n, t, nd) in net._weight_tensors():
        out.append(vals[n])
    return out


@pytest.mark.parametrize('tag', h5_cases.CASES)
def test_reads_keras_save_weights_file(tag):
    case, vals = h5_cases.values(tag)
    with h5lite.File(os.path.join(H5, tag + '_weights.h5'), 'r') as f:
        assert f.attrs['backend'] == b'tensorflow' and f.attrs['keras_version'] == h5_cases.KERAS_VERSION.encode()
        names = [n.decode() for n in f.attrs['layer_names']]
        assert names == [l['name'] for l in case['graph']['layers']]           # weight-less layers are listed too
        assert sorted(f.keys()) == sorted(names)
        seen = 0
        for l in case['graph']['layers']:
            wn = [w.decode() for w in f[l['name']].attrs['weight_names']]
            assert wn == ['%s/%s:0' % (l['name'], v) for v in h5_cases.keras_variables(l)]
            for w in wn:
                d = f[l['name']][w]                                             # nested group layer/layer/variable:0
                assert d.shape == vals[w[:-2]].shape and d.dtype == np.float32
                np.testing.assert_array_equal(np.asarray(d), vals[w[:-2]])
                seen += 1
        assert seen == len(vals) > 0


@pytest.mark.parametrize('tag', h5_cases.CASES)
def test_network_loads_keras_files(tag, monkeypatch):
    """ConvNet.load_weights on the save_weights layout and on the model.save layout (group model_weights, layers sorted by name as
    recent Keras writes them), by order and by name -- neurite/tf/modelio.py:111-123"""
    monkeypatch.setitem(sys.modules, 'h5py', None)                              # the h5lite branch even where an h5py exists
    case, vals = h5_cases.values(tag)
    net = _build(case)
    want = _expected_in_slots(net, vals)
    assert len(want) == len(vals)
    for fn in (tag + '_weights.h5', tag + '_model.h5'):
        net = _build(case)
        net.load_weights(os.path.join(H5, fn))
        for (n, t, nd), got, w in zip(net._weight_tensors(), net.get_weights(), want):
            np.testing.assert_array_equal(np.asarray(got).reshape(w.shape), w, err_msg='%s %s' % (fn, n))
    net = _build(case)
    before = [np.array(a, copy=True) for a in net.get_weights()]
    net.load_weights(os.path.join(H5, tag + '_weights.h5'), by_name=True)
    assert any(not np.array_equal(a, b) for a, b in zip(before, net.get_weights()))
    # the model file carries its configuration as JSON bytes (LoadableModel.load_config, modelio.py:125-143)
    with h5lite.File(os.path.join(H5, tag + '_model.h5')) as f:
        cfg = json.loads(f.attrs['model_config'].decode('utf-8'))
        assert cfg['class_name'] == case['builder'] and cfg['config']['args'] == case['args']
        assert json.loads(f.attrs['training_config'].decode())['optimizer_config']['class_name'] == 'Adam'
        ow = f['optimizer_weights']
        names = [n.decode() for n in ow.attrs['weight_names']]
        assert names[0] == 'Adam/iter:0' and ow[names[0]][()] == 12345 and ow[names[0]].dtype == np.int64
        assert len(names) == 1 + 2 * len(vals)
        np.testing.assert_array_equal(np.asarray(ow[names[1]]), np.full(want[0].shape, 0.5, np.float32))


def test_reads_the_storage_forms_of_the_library():
    ref = np.random.default_rng(11).standard_normal((37, 21, 5)).astype(np.float32)
    with h5lite.File(os.path.join(H5, 'storage_forms.h5')) as f:
        for k in ('contiguous', 'chunked', 'gzip_shuffle', 'big_endian'):
            assert f[k].shape == ref.shape
            np.testing.assert_array_equal(np.asarray(f[k]), ref, err_msg=k)
        np.testing.assert_array_equal(np.asarray(f['gzip_fletcher']), ref.astype(np.float64))
        np.testing.assert_array_equal(np.asarray(f['int16']), (ref * 100).astype(np.int16))
        np.testing.assert_array_equal(np.asarray(f['uint8']), (np.abs(ref) * 20).astype(np.uint8))
        np.testing.assert_array_equal(np.asarray(f['float16']), ref.astype(np.float16))
        np.testing.assert_array_equal(np.asarray(f['bool']), ref > 0)
        np.testing.assert_array_equal(np.asarray(f['compact']), np.arange(12, dtype=np.int32).reshape(3, 4))
        assert f['scalar'][()] == 2.5 and f['scalar'].shape == ()
        assert f['empty'].shape == (0, 3) and np.asarray(f['empty']).size == 0
        np.testing.assert_array_equal(np.asarray(f['never_written']), np.zeros((4, 3), np.float32))     # no storage allocated: fill value
        assert list(np.asarray(f['strings_fixed'])) == [b'alpha', b'be', b'gamma!']
        assert list(np.asarray(f['strings_vlen'])) == ['one', 'zwölf', '']
        a = f.attrs
        assert a['str_scalar'] == 'variable-length ünicode' and a['bytes_scalar'] == b'fixed bytes'
        assert list(a['str_list']) == ['a', 'bb', 'ccc'] and a['int'] == 7 and bool(a['bool']) is True
        np.testing.assert_array_equal(a['float_array'], np.arange(5) / 4)
        assert len(a['empty_list']) == 0 and a['empty'] is None
        big = json.loads(a['big_json'].decode('utf-8'))          # 190 KB through the global heap (how a large Keras model_config is stored)
        assert len(a['big_json']) > 100000 and len(big['layers']) == 2000 and big['layers'][1999]['config']['filters'] == 1999
        assert 'missing' not in a and sorted(a) == sorted(['str_scalar', 'bytes_scalar', 'str_list', 'int', 'float_array', 'bool', 'empty_list', 'empty', 'big_json'])
        with pytest.raises(KeyError):
            a['missing']
        assert f['chunked'].attrs['note'] == b'attribute on a dataset'
        assert f['nested/deeper/deepest'].attrs['depth'] == 3 and 'nested/deeper' in f and 'nested/nothing' not in f
        np.testing.assert_array_equal(np.asarray(f['nested']['deeper']['deepest/x']), np.arange(3))
        np.testing.assert_array_equal(np.asarray(f['/nested/deeper/deepest/x']), np.arange(3))
        with pytest.raises(KeyError):
            f['nested/nothing']
        with pytest.raises(h5lite.H5Error):
            f.attrs['x'] = 1                                                    # read-only


def test_many_links_and_split_attributes():
    """700 groups under the root (a two-level group B-tree in the library's file) and `layer_names` split into layer_names0 / 1 as Keras
    does past 64512 bytes (hdf5_format.py: save_attributes_to_hdf5_group)"""
    with h5lite.File(os.path.join(H5, 'many_layers.h5')) as f:
        assert 'layer_names' not in f.attrs and 'layer_names0' in f.attrs and 'layer_names1' in f.attrs
        names = [n.decode() for n in models._h5_attr(f, 'layer_names')]
        assert len(names) == 700 and len(f) == 700 and sorted(f.keys()) == sorted(names)
        for i in (0, 350, 699):
            wn = models._h5_attr(f[names[i]], 'weight_names')
            np.testing.assert_array_equal(np.asarray(f[names[i]][wn[0].decode()]), np.full((2, 2), float(i), np.float32))
        assert len(models._h5_attr(f[names[1]], 'weight_names')) == 0
        with pytest.raises(KeyError):
            models._h5_attr(f, 'no_such_attribute')


def test_reads_latest_format_small_groups():
    """libver='latest': superblock 3, version-2 object headers, compact link messages, layout version 4"""
    with h5lite.File(os.path.join(H5, 'latest_small.h5')) as f:
        assert list(f.attrs['layer_names']) == [b'a', b'b'] and f.keys() == ['a', 'b']
        np.testing.assert_array_equal(np.asarray(f['b']['b/kernel:0']), np.arange(6, dtype=np.float32).reshape(2, 3) + ord('b'))
        assert list(f['a'].attrs['weight_names']) == [b'a/kernel:0']


def _write_sample(path):
    rng = np.random.default_rng(5)
    expect = {}
    with h5lite.File(path, 'w') as f:
        names = ['layer_%04d' % i for i in range(300)]
        f.attrs['layer_names'] = [n.encode() for n in names]
        f.attrs['backend'] = b'tensorflow'
        f.attrs['model_config'] = json.dumps({'class_name': 'unet', 'config': {'x': [1, 2]}})
        f.attrs['count'] = 3
        f.attrs['rate'] = np.float32(2.5)
        f.attrs['table'] = np.arange(6, dtype=np.int16).reshape(2, 3)
        for i, n in enumerate(names):
            g = f.create_group(n)
            if i % 50 == 0:
                wn = ['%s/kernel:0' % n, '%s/bias:0' % n]
                g.attrs['weight_names'] = [w.encode() for w in wn]
                expect[n + '/' + wn[0]] = rng.standard_normal((3, 3, 2, 4)).astype(np.float32)
                expect[n + '/' + wn[1]] = np.full((4,), i, np.float64)
                g.create_dataset(wn[0], data=expect[n + '/' + wn[0]])
                g.create_dataset(wn[1], data=expect[n + '/' + wn[1]])
            else:
                g.attrs['weight_names'] = []
        f.create_dataset('scalar', data=np.int64(9))
        f.create_dataset('empty', data=np.zeros((0, 4), np.float32))
        f.create_dataset('half', data=np.arange(4, dtype=np.float16))
        f.create_dataset('flags', data=np.array([True, False, True]))
        with pytest.raises(ValueError):
            f.create_group('layer_0000')
        with pytest.raises(ValueError, match='too large'):
            f.attrs['huge'] = [b'x' * 100] * 700
            h5lite._Writer()._attr_messages(f)
        del f.attrs._load()['huge']
    return names, expect


def test_write_then_read_back(tmp_path):
    p = str(tmp_path / 'lite.h5')
    names, expect = _write_sample(p)
    with h5lite.File(p) as f:
        assert [n.decode() for n in f.attrs['layer_names']] == names and f.attrs['count'] == 3 and f.attrs['rate'] == np.float32(2.5)
        assert json.loads(f.attrs['model_config'].decode())['config'] == {'x': [1, 2]}
        np.testing.assert_array_equal(f.attrs['table'], np.arange(6, dtype=np.int16).reshape(2, 3))
        assert sorted(f.keys()) == sorted(names + ['scalar', 'empty', 'half', 'flags'])
        for k, v in expect.items():
            np.testing.assert_array_equal(np.asarray(f[k]), v)
            assert f[k].dtype == v.dtype
        assert f['scalar'][()] == 9 and f['empty'].shape == (0, 4) and len(f['layer_0001'].attrs['weight_names']) == 0
        np.testing.assert_array_equal(np.asarray(f['half']), np.arange(4, dtype=np.float16))
    with pytest.raises(h5lite.H5Error, match='signature'):
        q = tmp_path / 'not.h5'
        q.write_bytes(b'PK\x03\x04' + b'\0' * 600)
        h5lite.File(str(q))


@pytest.mark.skipif(not _have_h5py(), reason='no interpreter with h5py in this image')
def test_the_hdf5_library_reads_what_h5lite_writes(tmp_path):
    """h5py (HDF5 1.10.6) walks the written file -- 304 links under the root: a B-tree of symbol nodes -- and finds every value;
    h5dump parses all of it"""
    p = str(tmp_path / 'lite.h5')
    names, expect = _write_sample(p)
    np.savez(str(tmp_path / 'expect.npz'), **{k.replace('/', '|'): v for k, v in expect.items()})
    code = r'''
import sys, json, h5py, numpy as np
p, e = sys.argv[1], dict(np.load(sys.argv[2]))
with h5py.File(p, 'r') as f:
    names = [n.decode() for n in f.attrs['layer_names']]
    assert len(names) == 300 and sorted(f.keys()) == sorted(names + ['scalar', 'empty', 'half', 'flags'])
    assert f.attrs['backend'] == b'tensorflow' and f.attrs['count'] == 3 and f.attrs['rate'] == 2.5
    assert json.loads(f.attrs['model_config'])['class_name'] == 'unet'
    assert f.attrs['table'].tolist() == [[0, 1, 2], [3, 4, 5]]
    for k, v in e.items():
        d = f[k.replace('|', '/')]
        assert d.dtype == v.dtype and np.array_equal(d[()], v), k
    assert all(len(f[n].attrs['weight_names']) == (2 if i % 50 == 0 else 0) for i, n in enumerate(names))
    assert f['scalar'][()] == 9 and f['empty'].shape == (0, 4) and f['half'][()].tolist() == [0, 1, 2, 3]
print('ok', len(e))
'''
    r = subprocess.run([H5PY_PYTHON, '-W', 'ignore', '-c', code, p, str(tmp_path / 'expect.npz')], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith('ok 12'), r.stderr[-2000:]
    h5dump = os.path.join(os.path.dirname(H5PY_PYTHON), 'h5dump')
    if os.path.exists(h5dump):
        r = subprocess.run([h5dump, p], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and 'layer_0299' in r.stdout and 'error' not in r.stderr.lower(), r.stderr[-2000:]


@pytest.mark.skipif(not _have_h5py(), reason='no interpreter with h5py in this image')
def test_a_saved_network_opens_in_h5py(tmp_path, monkeypatch):
    """ConvNet.save_weights / save write the Keras layouts through h5lite; h5py lists the layers and returns the kernels"""
    monkeypatch.setitem(sys.modules, 'h5py', None)
    case, vals = h5_cases.values('two_d_pool')
    net = _build(case)
    net.load_weights(os.path.join(H5, 'two_d_pool_weights.h5'))
    p, m = str(tmp_path / 'w.h5'), str(tmp_path / 'm.h5')
    net.save_weights(p)
    net.save(m)
    back = models.load(m)
    for a, b in zip(back.get_weights(), net.get_weights()):
        np.testing.assert_array_equal(a, b)
    np.savez(str(tmp_path / 'vals.npz'), **{k.replace('/', '|'): v for k, v in vals.items()})
    code = r'''
import sys, json, h5py, numpy as np
w, m, e = sys.argv[1], sys.argv[2], dict(np.load(sys.argv[3]))
for path, grp in ((w, None), (m, 'model_weights')):
    with h5py.File(path, 'r') as f:
        g = f[grp] if grp else f
        names = [n.decode() for n in g.attrs['layer_names']]
        assert g.attrs['keras_version'] == b'2.4.0' and g.attrs['backend'] == b'tensorflow'
        n = 0
        for l in names:
            for wn in g[l].attrs['weight_names']:
                wn = wn.decode()
                assert np.array_equal(g[l][wn][()], e[wn[:-2].replace('/', '|')]), wn
                n += 1
        assert n == len(e), (n, len(e))
        if grp:
            assert json.loads(f.attrs['model_config'])['class_name'] == 'unet'
print('ok')
'''
    r = subprocess.run([H5PY_PYTHON, '-W', 'ignore', '-c', code, p, m, str(tmp_path / 'vals.npz')], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith('ok'), r.stderr[-2000:]
