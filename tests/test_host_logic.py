"""
CPU tests of the host side: the C-ABI library loads and exports every symbol the header declares,
argument validation and error types match the reference, shape/grid helpers reproduce the golden
vectors, and there is NO CPU compute path (the product raises without a ROCm device).
No kernel is launched here.
"""

import ctypes
import os
import sys
import warnings

import numpy as np
import pytest
import torch

import neurite_amd as ne
from conftest import golden_cases, load_golden


def test_library_loads_and_exports_header_symbols():
    lib = ne._lib.lib()
    declared = ne._lib.declared_symbols()
    assert len(declared) >= 13
    for name in declared:
        assert hasattr(lib, name), 'libneurite_amd.so does not export %s' % name
    # every exported symbol has a typed binding
    assert set(declared) == set(ne._lib._SIGNATURES)
    assert lib.nrt_abi_version() == 1
    assert lib.nrt_target_arch() == b'gfx950'
    assert lib.nrt_status_string(0) == b'ok'
    assert b'workspace' in lib.nrt_status_string(-4)
    assert os.path.samefile(ne.library_path(), os.path.join(os.path.dirname(ne.__file__), 'lib', 'libneurite_amd.so'))


def test_abi_rejects_bad_arguments_without_a_gpu():
    lib = ne._lib.lib()
    shp = ne._lib.ints([4, 4, 4])
    # NULL pointers / bad ranks are rejected before any launch
    assert lib.nrt_interpn_f32(None, None, None, 3, shp, shp, 1, 1, 64, 192, 0, 0, 0, 0.0, None) == -1
    assert lib.nrt_interpn_f32(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), 4, shp, shp, 1, 1,
                               64, 192, 0, 0, 0, 0.0, None) == -1
    assert lib.nrt_dice_soft_f32(None, None, 10, 4, 1, 0, 0.0, None, None, None, None, 0, None) == -1
    assert lib.nrt_dice_soft_f32(ctypes.c_void_p(16), ctypes.c_void_p(16), 10, 4, 1, 0, 0.0, ctypes.c_void_p(16),
                                 ctypes.c_void_p(16), None, None, 0, None) == -4
    assert lib.nrt_wcce(None, None, 0, None, 10, 4, 0, 0.0, None, None, None, 0, None) == -1
    assert lib.nrt_dice_workspace_bytes(100, 32, 2) > 0
    assert lib.nrt_wcce_workspace_bytes(100, 16) > 0
    # the conv / LocallyConnected3D epilogues fuse none / elu / relu only: any other activation code is an error, never a silent
    # linear result (include/neurite_amd.h "element-wise activations")
    p16 = ctypes.c_void_p(16)
    for act in (3, 4, 10, -1):
        assert lib.nrt_conv1x1_softmax_f32(p16, p16, p16, p16, 64, 16, 32, 0, act, None) == -1
        assert lib.nrt_lc3d_f(p16, p16, p16, p16, 0, 1, shp, 2, ne._lib.ints([3, 3, 3]), ne._lib.ints([1, 1, 1]), 4, act, 0, None) == -1
        assert lib.nrt_conv3d_up2_f32(p16, 16, p16, 16, p16, p16, p16, 1, shp, 16, act, None) == -1


def test_no_cpu_fallback():
    v = torch.zeros(4, 4, 4, 2)
    loc = torch.zeros(2, 2, 2, 3)
    with pytest.raises(ne.errors.NeuriteAmdError, match='no CPU fallback'):
        ne.utils.interpn(v, loc)
    with pytest.raises(ne.errors.NeuriteAmdError):
        ne.utils.resize(v, 2)
    with pytest.raises(ne.errors.NeuriteAmdError):
        ne.layers.SpatialTransformer()([v[None], loc.new_zeros(1, 4, 4, 4, 3)])
    with pytest.raises(ne.errors.NeuriteAmdError):
        ne.layers.Resize(2)(v[None])
    with pytest.raises(ne.errors.NeuriteAmdError):
        ne.metrics.Dice().dice(v[None], v[None])
    with pytest.raises(ne.errors.NeuriteAmdError):
        ne.metrics.CategoricalCrossentropy()(v[None], v[None])
    with pytest.raises(ne.errors.NeuriteAmdError):
        ne.distributed.dice_from_sums(torch.zeros(1, 3, 4))


def test_product_never_imports_the_oracle():
    root = os.path.dirname(ne.__file__)
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in text and 'from oracle' not in text, f
                assert 'liboracle' not in text, f


def test_interpn_argument_errors_match_reference():
    v = torch.zeros(4, 4, 4, 2)
    with pytest.raises(Exception, match='Number of loc Tensors 2 does not match volume dimension 3'):
        ne.utils.interpn(v, torch.zeros(5, 2))                    # utils.py:111-113
    with pytest.raises(Exception, match='does not match volume dimension'):
        ne.utils.interpn(torch.zeros(4), torch.zeros(5, 2))
    with pytest.raises(AssertionError, match='method should be linear or nearest, got: cubic'):
        ne.utils.interpn(v, torch.zeros(5, 3), interp_method='cubic')   # utils.py:194-195
    with pytest.raises(AssertionError, match='zoom_factor length'):
        ne.utils.resize(torch.zeros(4, 4), [2, 2, 2, 2])           # utils.py:241-242
    x = torch.zeros(3, 3, 3, 1)
    assert ne.utils.zoom(x, 1) is x and ne.utils.resize(x, [1, 1, 1]) is x      # utils.py:250-251
    assert ne.utils.zoom is ne.utils.resize


def test_layer_protocol():
    r = ne.layers.Resize(2, name='up')
    assert ne.layers.Zoom is ne.layers.Resize
    assert r.get_config() == {'name': 'up', 'zoom_factor': 2, 'interp_method': 'linear'}
    r.build((None, 4, 5, 6, 3))
    assert r.ndims == 3 and r.zoom_factor == [2, 2, 2]
    assert r.compute_output_shape((2, 4, 5, 6, 3)) == (2, 8, 10, 12, 3)
    g = load_golden('resize_small')
    assert tuple(g['layer__out_shape']) == r.compute_output_shape(g['layer__x'].shape)
    with pytest.raises(AssertionError, match='zoom factor length 2 does not match number of dimensions 3'):
        ne.layers.Resize([2, 2]).build((None, 4, 5, 6, 3))
    with pytest.raises(Exception, match='Resize must be called on a list of length 1'):
        ne.layers.Resize(2).build([(1, 2, 2, 1), (1, 2, 2, 1)])
    st = ne.layers.SpatialTransformer(interp_method='nearest', fill_value=0)
    cfg = st.get_config()
    assert cfg['interp_method'] == 'nearest' and cfg['indexing'] == 'ij' and cfg['fill_value'] == 0
    assert cfg['single_transform'] is False and cfg['shift_center'] is True
    with pytest.raises(AssertionError, match="indexing has to be"):
        ne.layers.SpatialTransformer(indexing='zz')
    with pytest.raises(TypeError):
        ne.layers.Resize(2, bogus=1)


def test_dice_constructor_contract():
    with pytest.raises(AssertionError):
        ne.metrics.Dice(input_type='nope')                           # metrics.py:406
    with pytest.raises(AssertionError, match='need nb_labels'):
        ne.metrics.Dice(dice_type='hard', input_type='max_label')    # :408-409
    with pytest.raises(AssertionError, match='probabilistic'):
        ne.metrics.Dice(dice_type='soft', input_type='max_label')    # :411-413
    d = ne.metrics.HardDice(5)
    assert d.dice_type == 'hard' and d.input_type == 'max_label' and d.nb_labels == 5
    s = ne.metrics.SoftDice(laplace_smoothing=0.1)
    assert s.dice_type == 'soft' and s.laplace_smoothing == 0.1 and s.check_input_limits is True
    assert issubclass(ne.losses.Dice, ne.metrics.Dice) and hasattr(ne.losses.Dice, 'mean_loss')
    assert ne.metrics.WeightedCategoricalCrossentropy is ne.metrics.CategoricalCrossentropy


def test_cce_label_weight_length_error():
    t = torch.zeros(1, 2, 2, 2, 6)
    with pytest.raises(ValueError, match='Label weights must be of len 6, but got 5.'):   # metrics.py:644-645
        ne.metrics.CategoricalCrossentropy(label_weights=np.ones(5))(t, t)
    with pytest.raises(ValueError, match='Label weights must be of len'):
        ne.losses.CategoricalCrossentropy(label_weights=[1., 2.]).loss(t, t)
    with pytest.raises(TypeError):
        ne.metrics.CategoricalCrossentropy(bogus=True)


def test_grid_and_index_helpers_golden():
    g = load_golden('index_helpers')
    for d, a in enumerate(ne.utils.volshape_to_ndgrid((3, 4, 2))):
        assert a.dtype == torch.int32 and np.array_equal(a.numpy(), g['ndgrid__%d' % d])
    for d, a in enumerate(ne.utils.volshape_to_meshgrid((3, 3, 2), indexing='xy')):
        assert np.array_equal(a.numpy(), g['meshgrid_xy__%d' % d])
    for d, a in enumerate(ne.utils.volshape_to_meshgrid((3, 4, 2), indexing='ij')):
        assert np.array_equal(a.numpy(), g['meshgrid_ij__%d' % d])
    subs = [torch.from_numpy(g['sub2ind__sub%d' % d]) for d in range(3)]
    out = ne.utils.sub2ind2d(tuple(g['sub2ind__siz']), subs)
    assert out.dtype == torch.int32 and np.array_equal(out.numpy(), g['sub2ind__out'])
    ws = [torch.from_numpy(g['prodn__w%d' % d]) for d in range(3)]
    assert np.array_equal(ne.utils.prod_n(ws).numpy(), g['prodn__out'])
    x = torch.from_numpy(g['bcf__x'])
    f = ne.utils.batch_channel_flatten(x)
    assert np.array_equal(f.numpy(), g['bcf__out']) and f.data_ptr() == x.data_ptr()      # a view
    assert np.array_equal(ne.utils.flatten_axes(x, [1, 2]).numpy(), g['flatten_axes_12__out'])
    with pytest.raises(ValueError, match='volshape needs to be a list of integers'):
        ne.utils.volshape_to_ndgrid((3, 4.5))
    with pytest.raises(ValueError, match="indexing parameter must be either"):
        ne.utils.meshgrid(torch.arange(2), indexing='zz')
    with pytest.raises(TypeError, match='invalid keyword argument'):
        ne.utils.meshgrid(torch.arange(2), foo=1)
    with pytest.raises(AssertionError, match='axes need to be contiguous'):
        ne.utils.flatten_axes(x, [1, 3])


def test_affine_to_dense_shift_matches_oracle():
    from oracle import np_oracle as npo
    rng = np.random.default_rng(4)
    A = (np.eye(3, 4) + 0.1 * rng.standard_normal((3, 4))).astype(np.float32)
    for center in (True, False):
        ours = ne.utils.affine_to_dense_shift(torch.from_numpy(A), (5, 6, 4), shift_center=center).numpy()
        ref = npo.affine_to_dense_shift(A, (5, 6, 4), shift_center=center)
        np.testing.assert_allclose(ours, ref, rtol=1e-5, atol=1e-5)
    # identity affine -> zero shift
    z = ne.utils.affine_to_dense_shift(torch.eye(4)[:3], (3, 3, 3))
    assert float(z.abs().max()) < 1e-6
    # a batch of affines (the form SpatialTransformer / AffineToDenseShift / compose hand over) = the stack of the single fields;
    # square [D + 1, D + 1] matrices drop their last row; a matrix that needs a gradient stays differentiable
    Ab = (np.eye(3, 4)[None] + 0.1 * rng.standard_normal((3, 3, 4))).astype(np.float32)
    both = ne.utils.affine_to_dense_shift(torch.from_numpy(Ab), (5, 6, 4))
    assert tuple(both.shape) == (3, 5, 6, 4, 3)
    for b in range(3):
        assert torch.equal(both[b], ne.utils.affine_to_dense_shift(torch.from_numpy(Ab[b]), (5, 6, 4)))
    sq = np.concatenate([Ab, np.tile(np.eye(4, dtype=np.float32)[-1:], (3, 1, 1))], 1)
    assert torch.equal(ne.utils.affine_to_dense_shift(torch.from_numpy(sq), (5, 6, 4)), both)
    g = torch.from_numpy(Ab[0]).requires_grad_()
    ne.utils.affine_to_dense_shift(g, (5, 6, 4)).sum().backward()
    assert g.grad is not None and tuple(g.grad.shape) == (3, 4)
    with pytest.raises(ValueError):
        ne.utils.affine_to_dense_shift(torch.zeros(3, 3), (5, 6, 4))


def test_shard_range():
    from neurite_amd.distributed import shard_range
    for n in (1, 7, 32, 33):
        for w in (1, 2, 3, 8):
            covered = []
            for r in range(w):
                lo, hi = shard_range(n, r, w)
                covered += list(range(lo, hi))
            assert covered == list(range(n))
    assert shard_range(32, 3, 8) == (12, 16)


def test_affine_helpers_cpu():
    """pure shape / matrix helpers of the voxelmorph companions need no device"""
    import torch
    from neurite_amd import utils
    assert utils.is_affine_shape((3, 4)) and utils.is_affine_shape((4, 4)) and utils.is_affine_shape((2, 3))
    assert not utils.is_affine_shape((8, 8, 8, 3)) and not utils.is_affine_shape((16, 1))
    with pytest.raises(ValueError, match='Affine matrix must be of shape'):
        utils.is_affine_shape((5, 6))
    m = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    sq = utils.make_square_affine(m)
    assert sq.shape == (4, 4) and sq[3].tolist() == [0, 0, 0, 1] and torch.equal(sq[:3], m)
    assert utils.make_square_affine(sq) is sq
    r = utils.rescale_affine(m, 0.5)
    assert torch.equal(r[:, :3], m[:, :3]) and torch.equal(r[:, 3], m[:, 3] * 0.5)
    with pytest.raises(ValueError, match='only supports ij'):
        utils.compose([m, m], indexing='xy')
    with pytest.raises(ValueError, match='greater than 1'):
        utils.compose([m])
    c = utils.compose([m, m])                      # affine o affine never touches the device
    assert torch.allclose(c, (sq @ sq)[:3])


def test_convnet_weight_api_cpu(tmp_path):
    """Keras-style get_weights / set_weights / save / load on the (un-run) graph; no device needed"""
    import contextlib, io
    import neurite_amd as ne
    with contextlib.redirect_stdout(io.StringIO()):
        net = ne.models.unet(4, (16, 16, 1), 2, 3, 3, batch_norm=-1)          # 2-D: kernels come back as [3, 3, Cin, Cout]
    w = net.get_weights()
    assert w[0].shape == (3, 3, 1, 4) and w[1].shape == (4,)
    assert any(a.shape == (1, 1, 4, 3) for a in w)                             # likelihood 1x1
    rng = np.random.default_rng(0)
    new = [rng.standard_normal(a.shape).astype(np.float32) for a in w]
    net.set_weights(new)
    assert all(np.array_equal(a, b) for a, b in zip(net.get_weights(), new))
    p = str(tmp_path / 'w.npz')
    net.save_weights(p)
    with contextlib.redirect_stdout(io.StringIO()):
        net2 = ne.models.unet(4, (16, 16, 1), 2, 3, 3, batch_norm=-1)
    net2.load_weights(p)
    assert all(np.array_equal(a, b) for a, b in zip(net2.get_weights(), new))
    with pytest.raises(ValueError, match='expecting'):
        net.set_weights(new[:-1])
    with pytest.raises(ValueError, match='not compatible'):
        net.set_weights([new[1]] + new[1:])


class _FakeH5Node(dict):
    """dict-backed stand-in for h5py.File / Group (the h5py branch of models._h5py: h5py is not importable here): attrs, create_group, create_dataset,
    item access by (possibly nested) name, `in`"""

    def __init__(self):
        super().__init__()
        self.attrs = {}

    def create_group(self, name):
        g = _FakeH5Node()
        dict.__setitem__(self, name, g)
        return g

    def create_dataset(self, name, data):
        dict.__setitem__(self, name, np.array(data))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _fake_h5py(store):
    import types
    mod = types.ModuleType('h5py')

    def File(path, mode='r'):
        if mode == 'w':
            store[path] = _FakeH5Node()
        return store[path]
    mod.File = File
    return mod


def test_loadable_model_cpu(tmp_path, monkeypatch):
    """LoadableModel behaviour (neurite/tf/modelio.py:78-143): the builder arguments travel with the weights; Keras HDF5
    layouts (save_weights and model.save) read back by layer name, in order, and by_name"""
    import contextlib, io, sys
    import neurite_amd as ne
    with contextlib.redirect_stdout(io.StringIO()):
        net = ne.models.unet(4, (12, 12, 12, 2), 2, 3, 5, name='seg', feat_mult=2, nb_conv_per_level=2, batch_norm=-1)
    cfg = net.get_config()
    assert cfg['nb_features'] == 4 and cfg['input_shape'] == [12, 12, 12, 2] and cfg['feat_mult'] == 2
    assert cfg['activation'] == 'elu' and cfg['metadata'] == {}                  # defaults recorded too
    net.metadata['trained_on'] = 'synthetic'
    rng = np.random.default_rng(1)
    new = [rng.standard_normal(a.shape).astype(np.float32) for a in net.get_weights()]
    net.set_weights(new)
    p = str(tmp_path / 'model.npz')
    net.save(p)
    assert ne.models.load_config(p)[0] == 'unet'
    with contextlib.redirect_stdout(io.StringIO()):
        back = ne.models.load(p)
    assert back.layer_names == net.layer_names and back.metadata == {'trained_on': 'synthetic'}
    assert all(np.array_equal(a, b) for a, b in zip(back.get_weights(), new))
    with contextlib.redirect_stdout(io.StringIO()), pytest.raises(ValueError):
        ne.models.load(p, nb_labels=7)                                           # override changes the head -> shape mismatch
    wp = str(tmp_path / 'w.npz')
    net.save_weights(wp)
    with pytest.raises(ValueError, match='weights only'):
        ne.models.load_config(wp)
    # without h5py the .h5 branch runs on the package's own HDF5 reader / writer (neurite_amd/h5lite.py; tests/test_h5lite.py)
    net.save_weights(str(tmp_path / 'w.h5'))
    with contextlib.redirect_stdout(io.StringIO()):
        lite = ne.models.unet(4, (12, 12, 12, 2), 2, 3, 5, name='seg', feat_mult=2, nb_conv_per_level=2, batch_norm=-1)
    lite.load_weights(str(tmp_path / 'w.h5'))
    assert all(np.array_equal(a, b) for a, b in zip(lite.get_weights(), new))
    # an encoder grafted into a decoder has no self-contained config
    with contextlib.redirect_stdout(io.StringIO()):
        enc = ne.models.conv_enc(4, (8, 8, 1), 2, 3, name='e')
        dec = ne.models.conv_dec(4, None, 2, 3, 2, name='d', prefix='e', input_model=enc, use_skip_connections=True)
    assert enc.get_config()['nb_levels'] == 2
    with pytest.raises(RuntimeError, match='part by part'):
        dec.save(str(tmp_path / 'dec.npz'))

    store = {}
    monkeypatch.setitem(sys.modules, 'h5py', _fake_h5py(store))
    net.save_weights('w.h5')
    f = store['w.h5']
    names = [n.decode() for n in f.attrs['layer_names']]
    assert names[0] == 'seg_conv_downarm_0_0' and f[names[0]].attrs['weight_names'][0] == b'seg_conv_downarm_0_0/kernel:0'
    assert f[names[0]]['seg_conv_downarm_0_0/kernel:0'].shape == (3, 3, 3, 2, 4)  # Keras layout
    with contextlib.redirect_stdout(io.StringIO()):
        other = ne.models.unet(4, (12, 12, 12, 2), 2, 3, 5, name='seg', feat_mult=2, nb_conv_per_level=2, batch_norm=-1)
    other.load_weights('w.h5')
    assert all(np.array_equal(a, b) for a, b in zip(other.get_weights(), new))
    # a Keras file also lists weight-less layers (inputs, pooling ...) and may use other layer names: order-based load
    k = _FakeH5Node()
    knames = []
    for i, n in enumerate(names):
        for extra in ('in%d' % i,):
            k.create_group(extra).attrs['weight_names'] = []
            knames.append(extra)
        g = k.create_group('L%d' % i)
        wn = [w.decode().replace(n, 'L%d' % i) for w in f[n].attrs['weight_names']]
        g.attrs['weight_names'] = [w.encode() for w in wn]
        for w_old, w_new in zip(f[n].attrs['weight_names'], wn):
            g.create_dataset(w_new, f[n][w_old.decode()])
        knames.append('L%d' % i)
    full = _FakeH5Node()
    dict.__setitem__(full, 'model_weights', k)
    k.attrs['layer_names'] = [n.encode() for n in knames]
    store['keras_model.h5'] = full
    with contextlib.redirect_stdout(io.StringIO()):
        third = ne.models.unet(4, (12, 12, 12, 2), 2, 3, 5, name='seg', feat_mult=2, nb_conv_per_level=2, batch_norm=-1)
    third.load_weights('keras_model.h5')
    assert all(np.array_equal(a, b) for a, b in zip(third.get_weights(), new))
    # by_name: only layers present in the file change
    part = _FakeH5Node()
    part.attrs['layer_names'] = [names[0].encode()]
    dict.__setitem__(part, names[0], f[names[0]])
    store['part.h5'] = part
    with contextlib.redirect_stdout(io.StringIO()):
        fourth = ne.models.unet(4, (12, 12, 12, 2), 2, 3, 5, name='seg', feat_mult=2, nb_conv_per_level=2, batch_norm=-1)
    before = fourth.get_weights()
    fourth.load_weights('part.h5', by_name=True)
    after = fourth.get_weights()
    assert np.array_equal(after[0], new[0]) and np.array_equal(after[1], new[1])
    assert all(np.array_equal(a, b) for a, b in zip(after[2:], before[2:]))
    with pytest.raises(ValueError):
        fourth.load_weights('part.h5')
    net.save('m.h5')
    with contextlib.redirect_stdout(io.StringIO()):
        fifth = ne.models.load('m.h5')
    assert all(np.array_equal(a, b) for a, b in zip(fifth.get_weights(), new))


def test_gaussian_kernel_and_synthesis_tables_cpu():
    """host-side pieces of the synthesis front-end: Gaussian kernels vs the reference's own output, label lookup tables"""
    import warnings
    import neurite_amd as ne
    from conftest import load_golden
    g = load_golden('filter_small')
    for tag, kw in (('gk_iso3', dict(sigma=[1.5, 1.5, 1.5])), ('gk_aniso', dict(sigma=[0.7, 2.0])),
                    ('gk_win', dict(sigma=[1.0, 2.0], windowsize=[5, 4])), ('gk_xy', dict(sigma=[1.0, 2.0], indexing='xy'))):
        np.testing.assert_allclose(ne.utils.gaussian_kernel(**kw).numpy(), g[tag + '__joint'], rtol=2e-6, atol=1e-9)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = ne.models.labels_to_image((8, 8, 8), [0, 3, 3, 17, 5], out_label_list={3: 1, 17: 1, 5: 2})
    c = m.cfg
    assert c['num_in_labels'] == 4 and c['in_lut'][[0, 3, 5, 17]].tolist() == [0, 1, 2, 3]       # np.unique sorts: 0, 3, 5, 17
    assert c['depth'] == 2 and c['out_lut'].tolist() == [-1, 0, 1, 0]             # one-hot over {1, 2}; background dropped (-1)
    assert c['mean_min'] == [0, 25, 25, 25] and c['std_max'] == [25] * 4
    # models.py:763-769: with `input_model` the generator is appended to it and takes ITS inputs
    calls = []

    class Front(torch.nn.Module):
        def forward(self, x):
            calls.append(tuple(x.shape))
            return [x]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        chained = ne.models.labels_to_image((8, 8, 8), [0, 1], input_model=Front())
    assert isinstance(chained.first, Front) and chained.second.cfg['num_in_labels'] == 2
    with pytest.raises(ne.errors.NeuriteAmdError):                       # the generator itself has no CPU path
        chained(torch.zeros(1, 8, 8, 8, 1))
    assert calls == [(1, 8, 8, 8, 1)]


def test_affine_sampling_helpers_cpu():
    """vxm-style helpers behind labels_to_image_new (host-side): parameter layout, matrix composition, flips, swaps"""
    from neurite_amd import augment, utils
    p = utils.draw_affine_params(shift=5, rot=10, scale=0.1, shear=0.05, ndims=3, batch_shape=[6], seeds=dict(shift=1, rot=2))
    assert p.shape == (6, 12)
    assert (p[:, :3].abs() <= 5).all() and (p[:, 3:6].abs() <= 10).all() and (p[:, 6:9].abs() <= 0.1).all()
    again = utils.draw_affine_params(shift=5, rot=10, scale=0.1, shear=0.05, ndims=3, batch_shape=[6], seeds=dict(shift=1, rot=2))
    assert torch.equal(p[:, :6], again[:, :6])                               # seeded components reproduce
    t = utils.draw_affine_params(scale=0.2, normal_scale=True, ndims=2, batch_shape=[500])
    assert t.shape == (500, 6) and (t[:, 3:5].abs() <= 0.4 + 1e-6).all()     # truncated at two SDs
    eye = utils.params_to_affine_matrix(torch.zeros(12), shift_scale=True, last_row=True)
    assert torch.equal(eye, torch.eye(4))
    m = utils.params_to_affine_matrix(torch.tensor([1., 2., 3., 90., 0., 0., 2., 3., 4., 0.5, 0., 0.]))
    want = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]], np.float64) @ np.diag([2., 3., 4.]) @ np.array([[1, .5, 0], [0, 1, 0], [0, 0, 1]])
    np.testing.assert_allclose(m[:, :3].numpy(), want, atol=1e-6)
    np.testing.assert_allclose(m[:, 3].numpy(), [1, 2, 3])
    r = utils.angles_to_rotation_matrix(torch.tensor([[10., 20., 30.], [0., 0., 45.]]))
    np.testing.assert_allclose((r @ r.transpose(1, 2)).numpy(), np.broadcast_to(np.eye(3), (2, 3, 3)), atol=1e-6)
    np.testing.assert_allclose(torch.linalg.det(r).numpy(), 1.0, atol=1e-6)
    r2 = utils.params_to_affine_matrix(torch.tensor([0., 0., 90.]), ndims=2)
    np.testing.assert_allclose(r2.numpy(), [[0, -1, 0], [1, 0, 0]], atol=1e-6)
    with pytest.raises(ValueError, match='exceeds'):
        utils.params_to_affine_matrix(torch.zeros(13))
    f = utils.draw_flip_matrix((8, 9, 10), shift_center=False, seed=3)
    assert f.shape == (4, 4) and set(torch.diag(f)[:3].tolist()) <= {1.0, -1.0}
    for d in range(3):                                                          # a flipped axis maps index i to (n - 1) - i
        if f[d, d] < 0:
            assert f[d, 3] == (8, 9, 10)[d] - 1
        else:
            assert f[d, 3] == 0
    s = utils.draw_swap_matrix(3, seed=1)
    assert sorted(s[:3, :3].sum(0).tolist()) == [1, 1, 1] and sorted(s[:3, :3].sum(1).tolist()) == [1, 1, 1] and s[3, 3] == 1
    down, up = utils.subsample_axis_indices(16, 3.3)
    assert down.tolist() == [0, 4, 8, 11, 15] and len(up) == 16 and up[0] == 0 and up[-1] == 4
    assert augment.normalize_axes(-1, (2, 3, 4)) == (2,) and augment.normalize_axes(None, (2, 3, 4), (1, 2), True) == (1, 2)
    with pytest.raises(IndexError):
        augment.normalize_axes(0, (2, 3, 4), allowed=(1, 2))
    x = torch.zeros(2, 10, 6, 1)
    mask = augment.draw_crop_mask(x, crop_min=0.3, crop_max=0.3, axis=1, prob=1, seed=5)
    assert mask.shape == (1, 10, 1, 1) and mask.sum() == 7 and (mask[0, :3, 0, 0].sum() == 0 or mask[0, 7:, 0, 0].sum() == 0)
    both = augment.draw_crop_mask(x, crop_min=0.4, crop_max=0.4, axis=(1, 2), bilateral=True, seed=6)
    assert sorted(both.shape)[-1] in (10, 6) and int(both.sum()) in (6, 3, 4)
    assert augment.draw_crop_mask(x, crop_max=0.5, prob=0, axis=1).sum() == 10


def test_labels_to_image_new_tables_cpu():
    """label look-up tables of labels_to_image_new (models.py:1145-1153, 1243-1258) and its argument checks"""
    import neurite_amd as ne
    m = ne.models.labels_to_image_new([0, 1, 2, 3], in_shape=(16, 16, 16))
    assert m.cfg['depth'] == 4 and m.cfg['gen_lut'].tolist() == [0, 1, 2, 3] and m.cfg['out_lut'].tolist() == [0, 1, 2, 3]
    m = ne.models.labels_to_image_new({0: 0, 3: 'a', 4: 'a', 9: 'b'}, labels_out={3: 1, 4: 1, 9: 2}, in_shape=(16, 16, 16))
    g = m.cfg['gen_lut']
    assert g[3] == g[4] and len({g[0], g[3], g[9]}) == 3 and m.cfg['num_label'] == 3        # left / right share an intensity
    assert m.cfg['out_lut'].tolist() == [-1, -1, -1, 0, 0, -1, -1, -1, -1, 1] and m.cfg['depth'] == 2
    m = ne.models.labels_to_image_new([0, 5], labels_out=[5], in_shape=(8, 8), one_hot=False)
    assert m.cfg['out_lut'].tolist() == [0, 0, 0, 0, 0, 5]
    with pytest.raises(AssertionError, match='unknown seeds'):
        ne.models.labels_to_image_new([0, 1], in_shape=(8, 8, 8), seeds=dict(nonsense=1))
    with pytest.raises(AssertionError, match='gamma value'):
        ne.models.labels_to_image_new([0, 1], in_shape=(8, 8, 8), gamma=1.5)
    front = torch.nn.Identity()
    chained = ne.models.labels_to_image_new([0, 1], in_shape=(8, 8, 8), input_model=front)     # models.py:1074-1077, 1301
    assert chained.first is front and chained.second.cfg['out_shape'].tolist() == [8, 8, 8]
    assert ne.models.labels_to_image_new([0, 1], in_shape=(8, 8, 8), out_shape=(8, 8, 8), half_res=True).cfg['out_shape'].tolist() == [4, 4, 4]


def test_synthstrip_config_roundtrip_cpu(tmp_path):
    """SynthStrip (models.py:1888-1967) records its constructor arguments and reloads them with the unet weights"""
    import contextlib, io
    import neurite_amd as ne
    with contextlib.redirect_stdout(io.StringIO()):
        m = ne.models.SynthStrip((16, 16, 16), [0, 1, 2, 3], {1: 1, 2: 1, 3: 1}, nb_unet_features=4, nb_unet_levels=2,
                                 gen_args=dict(warp_std=0.5))
    assert m.get_strip_model() is m.unet and m.unet.layer_names[-1] == 'unet_prediction'
    assert m.unet.output_shape[-1] == 1 and m.gen_model.cfg['one_hot'] is False
    assert m.gen_model.cfg['out_lut'].tolist() == [0, 1, 1, 1]
    rng = np.random.default_rng(0)
    new = [rng.standard_normal(a.shape).astype(np.float32) for a in m.unet.get_weights()]
    m.unet.set_weights(new)
    m.metadata['note'] = 'x'
    p = str(tmp_path / 'ss.npz')
    m.save(p)
    with contextlib.redirect_stdout(io.StringIO()):
        back = ne.models.SynthStrip.load(p)
    assert back.get_config()['labels_out'] == {1: 1, 2: 1, 3: 1} and back.metadata == {'note': 'x'}
    assert back.gen_model.cfg['out_lut'].tolist() == [0, 1, 1, 1] and back.gen_model.cfg['warp_std'] == 0.5
    assert all(np.array_equal(a, b) for a, b in zip(back.unet.get_weights(), new))
    assert back.unet.training is False
    back.train()
    assert back.unet.training is True
    with contextlib.redirect_stdout(io.StringIO()):
        net = ne.models.unet(4, (16, 16, 16, 1), 2, 3, 2)
    net.save(str(tmp_path / 'u.npz'))
    with pytest.raises(ValueError, match='not saved from a SynthStrip'):
        ne.models.SynthStrip.load(str(tmp_path / 'u.npz'))


def test_add_prior_and_dilation_net_graphs_cpu():
    """models.add_prior as a stand-alone builder (models.py:378-436) and the dilation_net wrapper (:45-85)"""
    import contextlib, io
    import neurite_amd as ne
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        ref = ne.models.unet(4, (8, 8, 8, 1), 2, 3, 3, add_prior_layer=True)
        base = ne.models.unet(4, (8, 8, 8, 1), 2, 3, 3, final_pred_activation='linear')
        post = ne.models.add_prior(base, (8, 8, 8, 3), name='unet_prior')
        prob = ne.models.add_prior(base, (8, 8, 8, 3), name='p', use_logp=False, final_pred_activation='linear')
        dil = ne.models.dilation_net(4, (8, 8, 8, 1), 2, 3, 3, dilation_rate_mult=2, feat_mult=7, nb_conv_per_level=3)
        plain = ne.models.unet(4, (8, 8, 8, 1), 2, 3, 3, dilation_rate_mult=2)
    assert post.layer_names == ref.layer_names and post.input_shapes == ref.input_shapes and post.name == 'unet_prior'
    assert prob.layer_names[-4:] == ['p-input', 'p_likelihood_sigmoid', 'p_posterior', 'p_prediction']
    assert [tuple(w.shape) for w in post.get_weights()] == [tuple(w.shape) for w in base.get_weights()]
    assert dil.layer_names == plain.layer_names and dil.name == 'unet'             # every other argument is ignored, as upstream
    assert [tuple(w.shape) for w in dil.get_weights()] == [tuple(w.shape) for w in plain.get_weights()]
    with pytest.raises(ValueError, match='does not match'):
        ne.models.add_prior(base, (8, 8, 8, 4))
    with pytest.raises(AssertionError, match='cannot do softmax'):
        ne.models.add_prior(base, (8, 8, 8, 3), use_logp=False)
    with pytest.raises(TypeError):
        ne.models.add_prior(object(), (8, 8, 8, 3))


def test_header_is_c99_and_a_c_program_links(tmp_path):
    """include/neurite_amd.h is a plain C header: examples/c_abi_example.c builds with gcc -std=c99 -pedantic, links against the
    shared library and (query mode, no GPU needed) runs"""
    import shutil
    import subprocess
    if shutil.which('gcc') is None or not os.path.isdir('/opt/rocm/lib'):
        pytest.skip('gcc / ROCm runtime not available')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(ne.library_path())
    exe = str(tmp_path / 'c_abi_example')
    cmd = ['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I', os.path.join(root, 'include'),
           os.path.join(root, 'examples', 'c_abi_example.c'), '-L', libdir, '-lneurite_amd', '-L', '/opt/rocm/lib', '-lamdhip64',
           '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib', '-o', exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and 'C ABI 1 for gfx950' in out.stdout and '"ok"' in out.stdout


def test_augment_golden_cpu():
    """draw_crop_mask and the index logic of subsample_axis against the REFERENCE'S OWN SOURCE run on scripted uniform draws
    (tests/golden/augment_small.npz): same number of draws in the same order, same masks / slices"""
    from neurite_amd import augment, utils
    g = load_golden('augment_small')
    for n in range(int(g['ncrop'])):
        kw = {}
        for k in ('crop_min', 'crop_max', 'prob'):
            if 'crop%d__%s' % (n, k) in g:
                kw[k] = float(g['crop%d__%s' % (n, k)])
        if 'crop%d__bilateral' % n in g:
            kw['bilateral'] = bool(g['crop%d__bilateral' % n])
        ax = g['crop%d__axis' % n]
        kw['axis'] = None if ax.ndim == 0 and int(ax) == -1 else (int(ax) if ax.ndim == 0 else tuple(int(v) for v in ax))
        draws = [float(d) for d in g['crop%d__draws' % n]]
        x = torch.zeros(tuple(int(v) for v in g['crop%d__shape' % n]))
        mask = augment.draw_crop_mask(x, _draws=draws, **kw).numpy()
        want = g['crop%d__mask' % n]
        assert mask.shape == want.shape and np.array_equal(mask, want), (n, kw, draws)
    for n in range(int(g['nsub'])):
        x, want = g['sub%d__x' % n], g['sub%d__out' % n]
        axes = g['sub%d__axes' % n]
        axes = [int(axes)] if axes.ndim == 0 else [int(v) for v in axes]
        prob = float(g['sub%d__prob' % n]) if 'sub%d__prob' % n in g else 1
        upsample = bool(g['sub%d__upsample' % n]) if 'sub%d__upsample' % n in g else True
        draws = [float(d) for d in g['sub%d__draws' % n]]
        ai, thick = utils._subsample_draws(len(axes), float(g['sub%d__stride_min' % n]), float(g['sub%d__stride_max' % n]), prob,
                                           None, draws)
        ax = axes[ai]
        down, up = utils.subsample_axis_indices(x.shape[ax], thick)
        got = np.take(x, down[up] if upsample else down, axis=ax)
        assert got.shape == want.shape and np.array_equal(got, want), (n, ax, float(thick))
    for n in range(int(g['ngk'])):                         # gaussian_kernel(random=True): one uniform draw per axis
        sig, mn = g['gk%d__sigma' % n], g['gk%d__min_sigma' % n]
        sig = float(sig) if sig.ndim == 0 else [float(v) for v in sig]
        mn = float(mn) if mn.ndim == 0 else [float(v) for v in mn]
        ks = utils.gaussian_kernel(sig, separate=True, random=True, min_sigma=mn, _draws=[float(d) for d in g['gk%d__draws' % n]])
        ks = ks if isinstance(ks, list) else [ks]
        for i, k in enumerate(ks):
            np.testing.assert_allclose(k.numpy(), g['gk%d__k%d' % (n, i)], rtol=2e-6, atol=1e-9)


def test_packed_weight_cache_is_dropped_when_weights_may_have_changed():
    """writes through `.data` bump no version counter; the MFMA-packed kernel copy must not survive a mode switch,
    set_weights or load_state_dict (neurite_amd/models.py::_Conv.invalidate_packed)."""
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        net = ne.models.unet(4, (8, 8, 8, 1), 2, 3, 2)
    convs = [m for m in net.layers_by_name.values() if hasattr(m, '_packed')]
    assert convs

    def poison():
        for m in convs:
            m._packed, m._packed_version = 'stale', (m.kernel._version, m.kernel.data_ptr(), m.kernel.device)

    for action in (lambda: net.train(), lambda: net.eval(), lambda: net.set_weights(net.get_weights()),
                   lambda: net.load_state_dict(net.state_dict())):
        poison()
        action()
        assert all(m._packed is None for m in convs)


def test_lc3d_relayout_plan_reproduces_the_reference_layer():
    """LocallyConnected3D implementations 1 (channels_first), 2 (dense-masked) and 3 (sparse) with 'valid' / 'same' padding:
    the host-side re-layout table (neurite_amd/layers.py::_lc3d_plan) applied to the reference-generated kernels, followed by a
    plain 'valid' un-shared convolution on the zero-padded input, gives the outputs of the reference LAYER
    (tests/golden/lc3d_impl.npz); implementation 3 orders its values like the reference's sorted kernel_idxs.  NumPy only."""
    from neurite_amd import layers as L
    cases = golden_cases(load_golden('lc3d_impl'))
    assert len(cases) == 10
    for tag, g in cases.items():
        x = g['x']
        ks, st = tuple(int(v) for v in g['ks']), tuple(int(v) for v in g['strides'])
        fmt, impl, pad, cout = str(g['data_format']), int(g['implementation']), str(g['padding']), int(g['filters'])
        cf = fmt == 'channels_first'
        xs = np.moveaxis(x, 1, -1) if cf else x
        ins, cin = xs.shape[1:4], xs.shape[-1]
        outs = tuple(L._conv_output_length(n, k, pad, s) for n, k, s in zip(ins, ks, st))
        plan = L._lc3d_plan(ins, cin, ks, st, pad, outs, cout, impl, fmt)
        if impl == 3:
            assert np.array_equal(np.stack(plan['pairs'], 1), g['kernel_idxs']), tag
            assert plan['nnz'] == g['kernel'].size
        W1 = g['kernel'] if plan['gather'] is None else \
            g['kernel'].reshape(-1)[plan['gather']] * (1 if plan['mask'] is None else plan['mask'])
        xp = xs.astype(np.float64)
        if plan['padded'] is not None:
            pb, P = plan['pad_before'], plan['padded']
            xp = np.zeros((xs.shape[0],) + tuple(P) + (cin,))
            xp[:, pb[0]:pb[0] + ins[0], pb[1]:pb[1] + ins[1], pb[2]:pb[2] + ins[2]] = xs
        y = np.zeros((xs.shape[0],) + outs + (cout,))
        oi = 0
        for r in range(outs[0]):
            for c in range(outs[1]):
                for z in range(outs[2]):
                    patch = xp[:, r * st[0]:r * st[0] + ks[0], c * st[1]:c * st[1] + ks[1], z * st[2]:z * st[2] + ks[2], :]
                    y[:, r, c, z] = patch.reshape(xs.shape[0], -1) @ W1[oi].astype(np.float64)
                    oi += 1
        b = g['bias']
        y = y + (np.transpose(b.reshape((cout,) + outs), (1, 2, 3, 0)) if cf else b)[None]
        act = str(g['activation'])
        y = np.maximum(y, 0) if act == 'relu' else (np.where(y > 0, y, np.exp(np.minimum(y, 0)) - 1) if act == 'elu' else y)
        y = np.moveaxis(y, -1, 1) if cf else y
        np.testing.assert_allclose(y, g['out'], rtol=1e-5, atol=1e-6 * np.abs(g['out']).max(), err_msg=tag)


def test_scoped_switches_are_thread_local_and_nest():
    """`deferred.scope()` / `checked.scope()`: an override for one block of one thread on top of the process-wide `enabled` attribute
    (a global mutable switch is a surprising thing for a drop-in user to have to touch)"""
    import threading
    from neurite_amd import checked, deferred
    for mod in (deferred, checked):
        base = mod.enabled
        assert mod.is_enabled() is base
        seen = {}

        def other():
            seen['inside'] = mod.is_enabled()                          # another thread does not see this thread's scope
            with mod.scope(not base):
                seen['own'] = mod.is_enabled()
        with mod.scope(not base):
            assert mod.is_enabled() is (not base) and mod.enabled is base
            with mod.scope(base):
                assert mod.is_enabled() is base
            assert mod.is_enabled() is (not base)
            t = threading.Thread(target=other)
            t.start()
            t.join()
        assert seen == {'inside': base, 'own': not base} and mod.is_enabled() is base
        try:
            with mod.scope(not base):
                raise KeyError('leaves the scope by an exception')
        except KeyError:
            pass
        assert mod.is_enabled() is base
        keep = mod.enabled
        try:
            mod.enabled = not keep                                        # the process-wide switch still works where no scope is open
            assert mod.is_enabled() is (not keep)
            with mod.scope(keep):
                assert mod.is_enabled() is keep
        finally:
            mod.enabled = keep


def test_deferred_warp_tensor_mechanics_cpu():
    """neurite_amd/deferred.py without a GPU: a DeferredWarp carries real metadata, evaluates its thunk exactly once on the first use by
    a torch op or by a host accessor that bypasses the dispatcher, and then behaves like the tensor it stands for"""
    from neurite_amd import deferred
    calls = []

    def thunk():
        calls.append(1)
        return torch.arange(48, dtype=torch.float32).reshape(1, 2, 2, 3, 4)

    d = deferred.DeferredWarp((1, 2, 2, 3, 4), torch.float32, torch.device('cpu'), thunk, dict(vol=None))
    assert isinstance(d, torch.Tensor) and d.pending and not calls
    assert tuple(d.shape) == (1, 2, 2, 3, 4) and d.dtype == torch.float32 and d.dim() == 5 and d.numel() == 48
    assert 'pending' in repr(d) and not calls                          # printing does not evaluate
    assert float((d * 2).sum()) == 2 * sum(range(48)) and calls == [1] and not d.pending
    assert type(d + 1) is torch.Tensor and type(d[0, 1]) is torch.Tensor and calls == [1]
    assert d.data_ptr() != 0 and d.tolist()[0][0][0][0] == [0.0, 1.0, 2.0, 3.0] and d.numpy().shape == (1, 2, 2, 3, 4)
    assert deferred.materialize(d) is d.materialize() and deferred.materialize(torch.ones(2)).shape == (2,)
    import pickle
    assert torch.equal(pickle.loads(pickle.dumps(d)), d.materialize())
    # accessors that bypass the dispatcher evaluate a still-pending tensor too
    for access in (lambda t: t.data_ptr(), lambda t: t.tolist(), lambda t: t.numpy(), lambda t: t.sum().item()):
        calls.clear()
        e = deferred.DeferredWarp((1, 2, 2, 3, 4), torch.float32, torch.device('cpu'), thunk, dict(vol=None))
        access(e)
        assert calls == [1] and not e.pending
    # gradients flow through a materialised operand like through any constant tensor
    x = torch.nn.Parameter(torch.ones(1, 2, 2, 3, 4))
    e = deferred.DeferredWarp((1, 2, 2, 3, 4), torch.float32, torch.device('cpu'), thunk, dict(vol=None))
    (x * e).sum().backward()
    assert torch.equal(x.grad, e.materialize())


def test_deferred_warp_detects_modified_inputs_cpu():
    """TensorFlow tensors are immutable: the value of SpatialTransformer's result is fixed when the layer returns.  A deferred warp that
    aliases its inputs must not silently compute the warp of data written afterwards: any in-place change of
    the volume or the transform -- through the tensor itself or through a view -- is detected when the warp is evaluated."""
    from neurite_amd import deferred

    def make():
        vol = torch.arange(24, dtype=torch.float32).reshape(1, 2, 3, 4)
        shift = torch.zeros(1, 2, 3, 3)
        d = deferred.DeferredWarp((1, 2, 3, 4), torch.float32, torch.device('cpu'), lambda: vol.clone(),
                                  dict(vol=vol, shift=shift, single_transform=False, fill_value=None))
        return vol, shift, d

    vol, shift, d = make()
    assert torch.equal(d.materialize(), vol)                            # untouched inputs: evaluates
    for mutate in (lambda v, s: v.copy_(torch.ones_like(v)), lambda v, s: v.mul_(2.0), lambda v, s: v[0, 0].zero_(),
                   lambda v, s: s.add_(1.0), lambda v, s: v.view(-1)[3:5].fill_(7.0)):
        vol, shift, d = make()
        mutate(vol, shift)
        with pytest.raises(deferred.DeferredWarpError):
            d.materialize()
        with pytest.raises(deferred.DeferredWarpError):
            d.check_sources()
        assert d.pending
        with pytest.raises(deferred.DeferredWarpError):                 # any torch op on the pending tensor evaluates it, and raises
            d + 1
    vol, shift, d = make()
    _ = vol + 1                                                          # out-of-place uses of the inputs are fine
    _ = shift * 2
    d.check_sources()
    # inference tensors have no version counter (reading it raises): the stamp must not; SpatialTransformer warps them eagerly
    with torch.inference_mode():
        vi = torch.ones(1, 2, 3, 4)
        di = deferred.DeferredWarp((1, 2, 3, 4), torch.float32, torch.device('cpu'), lambda: vi.clone(),
                                   dict(vol=vi, shift=torch.zeros(1, 2, 3, 3), single_transform=False, fill_value=None))
        assert torch.equal(di.materialize(), vi)
    assert torch.equal(d.materialize(), vol)
    vol.mul_(3.0)                                                        # after the evaluation the result is its own tensor
    assert torch.equal(d.materialize(), torch.arange(24, dtype=torch.float32).reshape(1, 2, 3, 4))


def test_lean_kernel_size_guard_cpu():
    """the few-channel tile kernel forms the byte offset of a voxel's location (12 B) and of its output row (4 C B) in 32-bit
    arithmetic; `nrt_lean_supported` has to refuse outputs beyond that, not only extents of 4096 and more (a 768^3 output passed the old
    guard and read its locations from wrapped addresses)."""
    import ctypes as C
    h = ne._lib.lib()
    fn = getattr(h, '_Z18nrt_lean_supportedPKiS0_iiPKvS2_S2_xx')
    fn.restype = C.c_bool
    fn.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong]
    ints = ne._lib.ints
    aligned = 0x10000

    def ok(vol_shape, out_shape, ch):
        return bool(fn(ints(vol_shape), ints(out_shape), ch, 3, aligned, aligned, aligned, 0, 0))

    assert ok([160, 160, 160], [160, 160, 160], 1) and ok([160, 160, 160], [160, 160, 160], 4)
    assert ok([64, 64, 64], [700, 700, 700], 1)                           # 343e6 voxels x 12 B < 2^32
    assert not ok([64, 64, 64], [768, 768, 768], 1)                       # 453e6 voxels x 12 B wraps
    assert not ok([64, 64, 64], [720, 720, 720], 1)                       # 373e6 x 12 B = 4.48e9 wraps
    assert ok([64, 64, 64], [640, 640, 640], 4) and not ok([64, 64, 64], [660, 660, 660], 4)      # output rows: 16 B per voxel
    assert not ok([64, 64, 64], [4096, 8, 8], 1) and not ok([64, 64, 64], [16, 16, 16], 5)


def test_folded_decoder_backward_matrices_cpu():
    """the host-side fold / unfold of the decoder convolution's backward (models._fold_dgrad_weights, _unfold_wgrad) against
    torch float64 autograd through UpSampling3D(2) + Conv3D 3x3x3 'same' on the CPU"""
    import torch
    import torch.nn.functional as Fn
    from neurite_amd import models as nm
    torch.manual_seed(0)
    c1, cout, S = 6, 5, (6, 4, 8)
    X1, Y1, Z1 = [v // 2 for v in S]
    lo = torch.randn(1, X1, Y1, Z1, c1, dtype=torch.float64, requires_grad=True)
    W = torch.randn(3, 3, 3, c1, cout, dtype=torch.float64, requires_grad=True)
    upt = lo.repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3)
    y = Fn.conv3d(upt.permute(0, 4, 1, 2, 3), W.permute(4, 3, 0, 1, 2), padding=1).permute(0, 2, 3, 4, 1)
    dpre = torch.randn_like(y)
    (y * dpre).sum().backward()
    s2d = dpre.reshape(1, X1, 2, Y1, 2, Z1, 2, cout).permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(1, X1, Y1, Z1, 8, cout)
    # gradient of the low-resolution input: a 3x3x3 convolution of the space-to-depth gradient with 2x2x2 live taps per group
    wf = nm._fold_dgrad_weights(W.detach())
    assert wf.shape == (3, 3, 3, 8 * cout, c1)
    dlo = Fn.conv3d(s2d.reshape(1, X1, Y1, Z1, 8 * cout).permute(0, 4, 1, 2, 3), wf.permute(4, 3, 0, 1, 2), padding=1).permute(0, 2, 3, 4, 1)
    assert float((dlo - lo.grad).abs().max()) < 1e-12
    for P in range(8):                                   # the taps nrt_conv3d_s2d_taps_f32 visits: e = (p ? 1 : 2) - t per axis
        live = (wf[:, :, :, P * cout:(P + 1) * cout, :].abs().sum((3, 4)) > 0).nonzero().tolist()
        p = ((P >> 2) & 1, (P >> 1) & 1, P & 1)
        want = sorted([[(1 if p[0] else 2) - tx, (1 if p[1] else 2) - ty, (1 if p[2] else 2) - tz]
                       for tx in (0, 1) for ty in (0, 1) for tz in (0, 1)])
        assert sorted(live) == want
    # folded weight gradient as nrt_conv3d_wgrad_s2d_f32 defines it, unfolded to the 27 taps
    lop = Fn.pad(lo.detach(), (0, 0, 1, 1, 1, 1, 1, 1))
    dwf = torch.zeros(8, 8, c1, cout, dtype=torch.float64)
    for P in range(8):
        p = ((P >> 2) & 1, (P >> 1) & 1, P & 1)
        for k in range(8):
            e = [p[0] + ((k >> 2) & 1), p[1] + ((k >> 1) & 1), p[2] + (k & 1)]
            xs = lop[0, e[0]:e[0] + X1, e[1]:e[1] + Y1, e[2]:e[2] + Z1]
            dwf[P, k] = torch.einsum('xyzi,xyzo->io', xs, s2d[0, :, :, :, P])
    assert float((nm._unfold_wgrad(dwf) - W.grad).abs().max()) < 1e-12


def test_bench_quotes_traffic_only_for_the_timed_kernel(tmp_path):
    """bench.py's roofline.traffic comes from a rocprofv3 --pmc pass recorded in profiles/hbm_traffic.json; it is keyed by the exact
    kernel name (the library names the instantiation it launches) and the batch, and another instantiation's counters are refused
    (the round-3 counters were taken on warp_dice_tile<8, 1, false, 4, float>, the timed kernel was <..., 3, float>)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    f = tmp_path / 't.json'
    f.write_text(json.dumps({'kernels': {'warp_dice_tile<8, 1, false, 4, float>': {'B4': {'bytes_per_launch': 123, 'source': 'x'}}}}))
    t, why = bench.lookup_traffic(str(f), 'warp_dice_tile<8, 1, false, 3, float>', 4)
    assert t is None and "warp_dice_tile<8, 1, false, 3, float>" in why
    t, why = bench.lookup_traffic(str(f), 'warp_dice_tile<8, 1, false, 4, float>', 4)
    assert t == 123 and 'B4' in why
    assert bench.lookup_traffic(str(f), 'warp_dice_tile<8, 1, false, 4, float>', 1)[0] is None          # another batch
    assert bench.lookup_traffic(str(f), None, 4)[0] is None
    # ... and only for the BINARY the counters were taken on: an entry carries the id of the library and of the gather's
    # sources; a kernel edit that keeps the kernel's name must not keep its old bytes
    k = 'warp_dice_wc<1, false, false, false, true, true>'
    f.write_text(json.dumps({'kernels': {k: {'B4': {'bytes_per_launch': 456, 'build_id': 'aaaa', 'gather_sources_id': 'gggg'}},
                                         'old': {'B4': {'bytes_per_launch': 7}}, 'lib_only': {'B4': {'bytes_per_launch': 8, 'build_id': 'aaaa'}}}}))
    assert bench.lookup_traffic(str(f), k, 4, ('aaaa', 'aaaa', 'gggg'))[0] == 456
    assert bench.lookup_traffic(str(f), k, 4, ('bbbb', 'bbbb', 'gggg'))[0] == 456          # another file of the library changed: the gather did not
    t, why = bench.lookup_traffic(str(f), k, 4, ('bbbb', 'bbbb', 'hhhh'))
    assert t is None and 'stale' in why and 'gggg' in why and 'hhhh' in why
    t, why = bench.lookup_traffic(str(f), k, 4, ('aaaa', 'cccc', 'gggg'))                  # the loaded library is not the tree's build
    assert t is None and 'not the build of this tree' in why
    t, why = bench.lookup_traffic(str(f), 'old', 4, ('aaaa', 'aaaa', 'gggg'))              # recorded before the ids existed
    assert t is None and 'no build id' in why
    assert bench.lookup_traffic(str(f), 'lib_only', 4, ('aaaa', 'aaaa', 'gggg'))[0] == 8
    assert bench.lookup_traffic(str(f), 'lib_only', 4, ('bbbb', 'bbbb', 'gggg'))[0] is None
    from neurite_amd import build as nbuild
    assert len(nbuild.gather_sources_id()) == 16 and nbuild.gather_sources_id() != nbuild.build_id()
    # the library names what it launches (no GPU needed: geometry only)
    lib = ne._lib.lib()
    n160 = ne._lib.ints([160] * 3)
    assert lib.nrt_warp_dice_kernel_name(n160, n160, 32, 4, 1, 0, 0, 0, 0) in (b'warp_dice_tile<8, 1, false, 3, float>',
                                                                              b'warp_dice_wc<1, false, false, false, true, true>')
    assert lib.nrt_warp_dice_kernel_name(n160, n160, 32, 4, 1, 0, 0, 0, 1 << 30) == b'warp_dice_tile<8, 1, false, 3, float>'
    assert lib.nrt_warp_dice_kernel_name(n160, n160, 32, 4, 1, 0, 0, 0, 1 << 29) == b'warp_dice_wc<1, false, false, false, true, true>'
    # the committed file itself has the keyed layout
    tj = json.load(open(os.path.join(root, 'profiles', 'hbm_traffic.json')))
    assert 'kernels' in tj and all('<' in k for k in tj['kernels'])


def test_multiple_losses_decorator_host_side():
    """neurite/tf/losses.py:225-246: weighted sum of callables; the Dice + CCE pairing logic of this package (no device work here)"""
    from neurite_amd import losses, metrics
    f = ne.losses.multiple_losses_decorator([lambda a, b: a + b, lambda a, b: a * b], [1, 2])
    assert f(1.0, 3.0) == 10.0
    assert ne.losses.multiple_losses_decorator([lambda a, b: a - b])(5.0, 3.0) == 2.0                     # weights default to ones
    c, d, h = ne.losses.CategoricalCrossentropy(), ne.losses.Dice(), ne.losses.HardDice(4)
    assert losses._owner(c.loss, metrics.CategoricalCrossentropy, ('loss', 'cce', '__call__')) is c
    assert losses._owner(c, metrics.CategoricalCrossentropy, ('loss',)) is c                              # the object itself is callable
    assert losses._owner(d.mean_loss, metrics.Dice, ('loss', 'mean_loss', 'dice', 'mean_dice')) is d
    assert losses._owner(lambda a, b: d.loss(a, b), metrics.Dice, ('loss',)) is None                      # hidden behind a lambda: not paired
    assert losses._owner(d.loss, metrics.CategoricalCrossentropy, ('loss',)) is None
    # CPU tensors never open a joint evaluation: the two losses run on their own and refuse the host tensors as always
    t = torch.zeros(1, 4, 4, 4, 8)
    assert metrics.JointSegLoss.open(d, c, t, t) is None and metrics.JointSegLoss.open(h, c, t, t) is None
    joint = ne.losses.multiple_losses_decorator([c.loss, d.mean_loss])
    with pytest.raises(ne.errors.NeuriteAmdError, match='no CPU fallback'):
        joint(t, t)
    assert getattr(metrics.JointSegLoss._tls, 'current', None) is None
    # a joint evaluation that raises leaves no context behind
    class Boom(metrics.JointSegLoss):
        def result(self):
            raise RuntimeError('boom')
    with pytest.raises(RuntimeError, match='boom'):
        with Boom(d, c, t, t) as j:
            assert metrics.JointSegLoss.lookup(d, t, t) is j and metrics.JointSegLoss.lookup(d, t, t.clone()) is None
            j.result()
    assert getattr(metrics.JointSegLoss._tls, 'current', None) is None


def test_checked_tensor_mechanics_on_cpu():
    """neurite_amd/checked.py without a device: a CheckedTensor lets device-side (here: any torch) operations through to the real values,
    host access looks at the assert first, operations that return the same values (detach, views, slices, clone) keep the pending assert,
    derived values are plain tensors.  (The extrema's asynchronous journey is GPU business: tests/test_gpu_deferred.py.)"""
    from neurite_amd import checked
    from neurite_amd.errors import InvalidArgumentError

    class Fake:
        def __init__(self, fail):
            self.fail, self.done, self.failed, self.looks = fail, False, False, 0

        def resolve(self, block=True):
            self.looks += 1
            self.done, self.failed = True, self.fail
            if self.fail:
                raise InvalidArgumentError('value outside range')
            return True

    v = torch.arange(6.).reshape(2, 3)
    ok = Fake(False)
    d = checked.CheckedTensor(v, ok)
    assert tuple(d.shape) == (2, 3) and d.dtype == v.dtype and d.data_ptr() == v.data_ptr()
    s = (d * 2).sum()
    assert type(s) is torch.Tensor and float(s) == 30.0 and ok.looks == 0          # derived values: no look at the assert
    assert d.tolist() == v.tolist() and ok.looks == 1
    assert checked.unwrap(d) is v and checked.unwrap(v) is v
    for touch in (lambda t: t.tolist(), lambda t: t[0, 0].item(), lambda t: t.numpy(), lambda t: repr(t), lambda t: torch.equal(t, v),
                  lambda t: np.asarray(t), lambda t: t.detach().reshape(-1)[:2].clone().tolist()):
        bad = Fake(True)
        with pytest.raises(InvalidArgumentError, match='value outside range'):
            touch(checked.CheckedTensor(v, bad))
        assert bad.looks == 1


def test_trace_service_time_tool(tmp_path):
    """tools/trace_service_time.py on a synthetic dispatch trace: three launches in flight -> own duration 3x the service time"""
    import json
    import subprocess
    rows = ['"Kind","Kernel_Name","Start_Timestamp","End_Timestamp"']
    for k in range(30):                                # a launch completes every 1 ms, each one is in flight for 3 ms
        rows.append('"KERNEL_DISPATCH","void (anonymous namespace)::warp_dice_wc<1, false>(int)",%d,%d' % (k * 1000000, k * 1000000 + 3000000))
    rows.append('"KERNEL_DISPATCH","other_kernel(int)",5,6')
    f = tmp_path / 'x_kernel_trace.csv'
    f.write_text('\n'.join(rows) + '\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'trace_service_time.py'), str(tmp_path), 'warp_dice_wc<1, false>'],
                         capture_output=True, text=True, timeout=60)
    j = json.loads(out.stdout)
    assert j['dispatches'] == 30 and j['mean_own_duration_ms'] == 3.0
    run = j['longest_runs_of_back_to_back_launches'][0]
    assert run['launches'] == 30 and abs(run['service_ms_per_launch'] - 32.0 / 30) < 1e-3 and run['fraction_of_span_with_two_or_more_in_flight'] > 0.9
