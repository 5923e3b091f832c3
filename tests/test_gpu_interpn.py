"""
GPU parity tests of interpn / SpatialTransformer / Resize: the HIP path (through the C ABI) against
(a) the golden vectors produced by the reference's own source and (b) the oracle on seeded inputs.
Tolerance: BIT-EXACT for linear as well as nearest -- the kernels reproduce the reference's float32 op
sequence with one rounding per op (the north-star tolerance of 1e-5 relative is therefore met with
margin; a separate assertion states it).
"""

import numpy as np
import pytest
import torch

import neurite_amd as ne
from conftest import bits_equal, golden_cases, load_golden
from neurite_amd import synth
from oracle import c_oracle as co
from oracle import np_oracle as npo

pytestmark = pytest.mark.gpu
F = np.float32


def G(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def ijk(shape):
    return np.stack(np.meshgrid(*[np.arange(s) for s in shape], indexing='ij'), -1).astype(F)


def test_native_library_is_loaded(dev):
    lib = ne._lib.lib()
    assert lib.nrt_target_arch() == b'gfx950'
    maps = open('/proc/self/maps').read()
    assert 'libneurite_amd.so' in maps


def test_interpn_golden_bit_exact(dev):
    cases = golden_cases(load_golden('interpn_small'))
    assert len(cases) >= 19
    for tag, c in cases.items():
        fill = float(c['fill']) if bool(c['hasfill']) else None
        loc = c['loc']
        if tag == 'd3_nochan_list':
            loc_arg = [G(np.ascontiguousarray(loc[..., d]), dev) for d in range(loc.shape[-1])]
        else:
            loc_arg = G(loc, dev)
        out = ne.utils.interpn(G(c['vol'], dev), loc_arg, str(c['method']), fill)
        assert out.shape == c['out'].shape, tag
        assert N(out).dtype == c['out'].dtype, tag
        assert bits_equal(N(out), c['out']), tag


def test_interpn_cfg1_golden_bit_exact(dev):
    g = load_golden('interpn_cfg1_32')
    out = N(ne.utils.interpn(G(g['vol'], dev), G(g['loc'], dev)))
    assert bits_equal(out, g['out'])
    # the tolerance north_star states, for the record
    np.testing.assert_allclose(out, npo.interpn_f64(g['vol'], g['loc']), rtol=1e-5, atol=1e-5)


def _tile(lx, ly, lz, zo):
    return lx | (ly << 4) | (lz << 8) | (zo << 12)


@pytest.mark.parametrize('variant,tune', [(1, 0), (2, 0), (2, 1), (2, 7), (3, 0), (3, 5), (3, 8), (3, 40), (4, 0),
                                          (4, 13), (5, 0), (5, _tile(0, 0, 5, 0)), (5, _tile(1, 1, 3, 1)),
                                          (5, _tile(2, 3, 4, 1)), (5, _tile(3, 3, 3, 0)), (5, _tile(4, 4, 3, 1)),
                                          (5, _tile(0, 0, 0, 0)), (5, (1 << 13) | (7 << 16)), (5, (1 << 13) | (1 << 12)),
                                          (5, (1 << 13) | (20 << 16))])
def test_c32_every_kernel_variant(dev, variant, tune):
    """All kernels that can serve C = 32 must agree bit-for-bit with the oracle, for sizes that are not
    multiples of the 4x8 patch / z-chunk / 8-voxel shift batch, smooth and rough fields, all loc modes."""
    rng = np.random.default_rng(100 + variant * 17 + tune)
    S = (13, 10, 27)
    vol = rng.standard_normal(S + (32,)).astype(F)
    for kind in ('smooth', 'rough', 'edge'):
        if kind == 'smooth':
            shift = N(synth.smooth_displacement(5, 27, 2.0, coarse=6))[:13, :10, :27].copy()
        elif kind == 'rough':
            shift = rng.uniform(-30, 30, S + (3,)).astype(F)
        else:   # exactly-integer and half-integer displacements, pushes past both borders
            shift = rng.choice(np.array([-3, -1, -0.5, 0, 0.5, 1, 2, 7.5], F), S + (3,)).astype(F)
        for fill in (None, 0.0):
            want = co.interpn(vol, shift, 'linear', fill, loc_mode=1)
            st = ne.layers.SpatialTransformer(fill_value=fill)
            st._variant, st._tune = variant, tune
            got = N(st([G(vol[None], dev), G(shift[None], dev)]))[0]
            assert bits_equal(got, want), (kind, fill)
            got = N(ne.utils.interpn(G(vol, dev), G(ijk(S) + shift, dev), fill_value=fill, _variant=variant, _tune=tune))
            assert bits_equal(got, co.interpn(vol, ijk(S) + shift, 'linear', fill)), (kind, fill, 'abs')


@pytest.mark.parametrize('C', [12, 20])
def test_channel_counts_multiple_of_four_off_the_row_kernels(dev, C):
    """float32 volumes with 12 / 20 channels (4 k, but no power of two): the auto-selection takes the rank-templated kernel of
    interpn_any.hip with 4 channels per thread -- bit-identical to the oracle and to the per-element kernel (variant 1)"""
    rng = np.random.default_rng(70 + C)
    S = (11, 9, 14)
    vol = rng.standard_normal(S + (C,)).astype(F)
    shift = rng.normal(0, 3, S + (3,)).astype(F)
    for method in ('linear', 'nearest'):
        for fill in (None, 0.5):
            st = ne.layers.SpatialTransformer(interp_method=method, fill_value=fill)
            got = N(st([G(vol[None], dev), G(shift[None], dev)]))[0]
            assert bits_equal(got, co.interpn(vol, shift, method, fill, loc_mode=1)), (method, fill)
            one = N(ne.utils.interpn(G(vol, dev), G(ijk(S) + shift, dev), interp_method=method, fill_value=fill, _variant=1))
            assert bits_equal(got, one), (method, fill, 'variant 1')
    up = N(ne.utils.resize(G(vol, dev), 1.5))
    assert bits_equal(up, npo.resize(vol, 1.5))
    vb = rng.standard_normal((2, 6, 7, 8, C)).astype(F)
    tb = rng.normal(0, 2, (2, 6, 7, 8, 3)).astype(F)
    assert bits_equal(N(ne.layers.SpatialTransformer()([G(vb, dev), G(tb, dev)])), npo.spatial_transformer(vb, tb))


def test_lean_kernel_warp_add(dev):
    """compose / VecInt update b + transform(a, b) at a size that takes the lean kernel (C = 3, 16-byte aligned tensors)"""
    rng = np.random.default_rng(72)
    S = (20, 16, 32)
    a = (rng.standard_normal(S + (3,)) * 2).astype(F)
    b = N(synth.smooth_displacement(9, 32, 2.0, coarse=4))[:20, :16, :32].copy()
    assert bits_equal(N(ne.utils.compose([G(a, dev), G(b, dev)])), npo.compose([a, b]))
    out = N(ne.layers.VecInt(int_steps=4)(G(b[None], dev)))
    assert bits_equal(out[0], npo.integrate_vec(b, 'ss', 4))


def test_lds_staged_warp_add(dev):
    """compose / integrate update b + transform(a, b) at a size that takes the LDS kernel"""
    rng = np.random.default_rng(71)
    S = (20, 18, 33)
    a = (rng.standard_normal(S + (3,)) * 2).astype(F)
    b = N(synth.smooth_displacement(9, 33, 2.0, coarse=6))[:20, :18, :33].copy()
    got = N(ne.utils.compose([G(a, dev), G(b, dev)]))
    assert bits_equal(got, npo.compose([a, b]))
    vel = b[None]
    out = N(ne.layers.VecInt(int_steps=4)(G(vel, dev)))
    assert bits_equal(out[0], npo.integrate_vec(vel[0], 'ss', 4))


@pytest.mark.parametrize('C', [1, 2, 3, 4, 8, 16, 64, 128, 5])
def test_channel_counts(dev, C):
    rng = np.random.default_rng(C)
    S = (11, 9, 14)
    vol = rng.standard_normal(S + (C,)).astype(F)
    shift = rng.normal(0, 2.5, S + (3,)).astype(F)
    for method in ('linear', 'nearest'):
        for fill in (None, -1.5):
            want = co.interpn(vol, shift, method, fill, loc_mode=1)
            got = N(ne.utils.transform(G(vol, dev), G(shift, dev), method, fill_value=fill))
            assert bits_equal(got, want), (method, fill)


def test_spatial_transformer_batch_single_xy_and_shapes(dev):
    rng = np.random.default_rng(9)
    B, S, So, C = 3, (9, 8, 12), (5, 11, 7), 32
    vol = rng.standard_normal((B,) + S + (C,)).astype(F)
    trf = rng.normal(0, 2, (B,) + So + (3,)).astype(F)      # output grid differs from the volume grid
    want = npo.spatial_transformer(vol, trf)
    got = N(ne.layers.SpatialTransformer()([G(vol, dev), G(trf, dev)]))
    assert got.shape == (B,) + So + (C,) and bits_equal(got, want)
    want1 = npo.spatial_transformer(vol, trf[:1], single_transform=True)
    got1 = N(ne.layers.SpatialTransformer(single_transform=True)([G(vol, dev), G(trf[:1], dev)]))
    assert bits_equal(got1, want1)
    wantxy = npo.spatial_transformer(vol, trf, indexing='xy', interp_method='nearest', fill_value=0)
    gotxy = N(ne.layers.SpatialTransformer('nearest', indexing='xy', fill_value=0)([G(vol, dev), G(trf, dev)]))
    assert bits_equal(gotxy, wantxy)
    # label maps through the nearest path as the reference does (models.py:806-807): ids survive exactly
    lab = rng.integers(0, 40, (B,) + S + (1,)).astype(F)
    trf_s = rng.normal(0, 2, (B,) + S + (3,)).astype(F)
    w = N(ne.layers.SpatialTransformer('nearest', fill_value=0)([G(lab, dev), G(trf_s, dev)]))
    assert bits_equal(w, npo.spatial_transformer(lab, trf_s, 'nearest', fill_value=0))
    # affine transform
    A = (np.eye(3, 4) + 0.05 * rng.standard_normal((B, 3, 4))).astype(F)
    wa = npo.spatial_transformer(vol, A)
    ga = N(ne.layers.SpatialTransformer()([G(vol, dev), G(A, dev)]))
    np.testing.assert_allclose(ga, wa, rtol=1e-4, atol=1e-4)    # the affine->shift matmul is host glue (torch vs numpy)


def test_low_dims_int_volumes_and_edge_shapes(dev):
    rng = np.random.default_rng(21)
    # 2-D and 1-D SpatialTransformer
    v2 = rng.standard_normal((2, 13, 9, 4)).astype(F)
    t2 = rng.normal(0, 2, (2, 13, 9, 2)).astype(F)
    assert bits_equal(N(ne.layers.SpatialTransformer()([G(v2, dev), G(t2, dev)])), npo.spatial_transformer(v2, t2))
    v1 = rng.standard_normal((2, 17, 3)).astype(F)
    t1 = rng.normal(0, 2, (2, 17, 1)).astype(F)
    assert bits_equal(N(ne.layers.SpatialTransformer(fill_value=1.0)([G(v1, dev), G(t1, dev)])),
                      npo.spatial_transformer(v1, t1, fill_value=1.0))
    # int32 / uint8 / int16 label volumes: nearest is pure data movement, with and without fill
    for dt in (np.int32, np.uint8, np.int16):
        vi = rng.integers(0, 100, (7, 6, 5, 2)).astype(dt)
        loc = rng.uniform(-2, 8, (4, 4, 4, 3)).astype(F)
        for fill in (None, 0, 3):
            got = N(ne.utils.interpn(G(vi, dev), G(loc, dev), 'nearest', fill))
            want = npo.interpn(vi, loc, 'nearest', fill)
            assert got.dtype == dt and np.array_equal(got, want), (dt, fill)
    with pytest.raises(TypeError, match='integer volume'):
        ne.utils.interpn(G(vi, dev), G(loc, dev))
    # empty output, singleton dims, single voxel
    e = ne.utils.interpn(G(v2[0], dev), torch.zeros((0, 2), device=dev))
    assert e.shape == (0, 4)
    vs = rng.standard_normal((1, 1, 6, 32)).astype(F)
    ls = rng.uniform(-2, 7, (3, 2, 5, 3)).astype(F)
    assert bits_equal(N(ne.utils.interpn(G(vs, dev), G(ls, dev))), npo.interpn(vs, ls))
    one = rng.standard_normal((1, 1, 1, 32)).astype(F)
    assert bits_equal(N(ne.utils.interpn(G(one, dev), G(ls, dev), fill_value=0.)), npo.interpn(one, ls, fill_value=0.))


@pytest.mark.parametrize('C', [1, 2, 3, 4])
def test_few_channel_box_form(dev, C):
    """Round 6 (utils.py:137-191 at the call site models.py:804): the LDS-staged form of the few-channel linear warp --
    a block stages the source bounding box of an 8 x 8 x 32 / 4 x 8 x 32 output tile in LDS (LDS-DMA, 16-byte pieces consecutive along z)
    and gathers the corners with ds_read (`interpn_lean_box`, variant 11) instead of sending 4-8 scattered lane accesses per voxel
    through the texture unit (`interpn_lean_tile`, variant 8, the default: the box form measured slower on every field,
    profiles/r06_lab/box_ab.jsonl).  Bit-identical to the tile form and to the oracle: ragged volumes (partial tiles on every side),
    locations outside the volume, fill, absolute locations, the addend epilogue (VecInt), NaN / inf locations (memory-safe), and a field
    steep enough that the boxes do not fit the LDS budget (the kernel then gathers that tile directly)."""
    rng = np.random.default_rng(600 + C)
    for S, O in (((40, 37, 70), (40, 37, 70)), ((17, 9, 33), (21, 13, 45)), ((64, 64, 96), (64, 64, 96))):
        vol = rng.standard_normal(S + (C,)).astype(F)
        base = np.stack(np.meshgrid(*[np.linspace(0, s - 1, o) for s, o in zip(S, O)], indexing='ij'), -1)
        for amp, fill in ((1.5, None), (4.0, 0.25), (60.0, None)):
            loc = (base + rng.normal(0, amp, O + (3,))).astype(F)
            got = N(ne.utils.interpn(G(vol, dev), G(loc, dev), fill_value=fill, _variant=11))
            tile = N(ne.utils.interpn(G(vol, dev), G(loc, dev), fill_value=fill, _variant=8))
            auto = N(ne.utils.interpn(G(vol, dev), G(loc, dev), fill_value=fill))
            assert bits_equal(got, tile) and bits_equal(auto, got), (S, O, amp)
            if np.prod(O) <= 40 * 37 * 70:
                assert bits_equal(got, npo.interpn(vol, loc, 'linear', fill)), (S, O, amp)
    # SpatialTransformer (identity grid + shift), batched, and the addend epilogue of VecInt / compose
    S = (24, 40, 64)
    vol = rng.standard_normal((2,) + S + (C,)).astype(F)
    trf = rng.normal(0, 2.5, (2,) + S + (3,)).astype(F)
    st = ne.layers.SpatialTransformer()
    a = N(st([G(vol, dev), G(trf, dev)]))
    st8 = ne.layers.SpatialTransformer()
    st8._variant = 8
    assert bits_equal(a, N(st8([G(vol, dev), G(trf, dev)])))
    for bi in range(2):
        assert bits_equal(a[bi], npo.interpn(vol[bi], ijk(S) + trf[bi]))
    # non-finite locations: same bits as the tile form wherever the reference is defined, no fault anywhere
    loc = (ijk(S) + trf[0]).astype(F)
    loc[3, 3, 3] = [np.inf, -np.inf, 1e30]
    loc[5, 6, 7] = [np.nan, np.nan, np.nan]
    g11 = N(ne.utils.interpn(G(vol[0], dev), G(loc, dev), _variant=11))
    g8 = N(ne.utils.interpn(G(vol[0], dev), G(loc, dev), _variant=8))
    ok = np.ones(S, bool)
    ok[5, 6, 7] = False
    assert bits_equal(g11[ok], g8[ok])
    torch.cuda.synchronize()
    # a z extent below the tile's 32 keeps the tile form; variant 11 then says so
    small = rng.standard_normal((8, 8, 16, C)).astype(F)
    ls = rng.uniform(0, 7, (8, 8, 16, 3)).astype(F)
    assert bits_equal(N(ne.utils.interpn(G(small, dev), G(ls, dev))), npo.interpn(small, ls))
    with pytest.raises(ne._lib.NeuriteAmdError):
        ne.utils.interpn(G(small, dev), G(ls, dev), _variant=11)


def test_non_finite_locations_are_memory_safe(dev):
    rng = np.random.default_rng(2)
    vol = rng.standard_normal((6, 6, 6, 32)).astype(F)
    loc = rng.uniform(0, 5, (4, 4, 8, 3)).astype(F)
    loc[0, 0, 0] = [np.inf, -np.inf, 1e30]
    loc[1, 1, 1] = [np.nan, np.nan, np.nan]
    for variant in (1, 2, 3, 5):
        for method in ('linear', 'nearest'):
            if variant in (3, 5) and method == 'nearest':
                continue
            out = N(ne.utils.interpn(G(vol, dev), G(loc, dev), method, _variant=variant))
            loc_ref = loc.copy()
            loc_ref[1, 1, 1] = 0                     # NumPy cannot index with int32(NaN)
            with np.errstate(invalid='ignore'):
                want = npo.interpn(vol, loc_ref, method)
            ok = np.ones(loc.shape[:3], bool)
            ok[1, 1, 1] = False                      # NaN locations are undefined in the reference
            if method == 'nearest':
                ok[0, 0, 0] = False                  # int32(round(+-inf / 1e30)) is undefined in the reference
            assert bits_equal(out[ok], want[ok])
            assert out.shape == want.shape
    torch.cuda.synchronize()


def test_properties(dev):
    rng = np.random.default_rng(77)
    S = (20, 17, 33)
    vol = rng.standard_normal((1,) + S + (32,)).astype(F)
    zero = np.zeros((1,) + S + (3,), F)
    st = ne.layers.SpatialTransformer()
    # identity warp returns the input bit-exactly
    assert bits_equal(N(st([G(vol, dev), G(zero, dev)])), vol)
    # integer shift = shifted copy with edge replication
    sh = zero.copy()
    sh[..., 2] = 3
    out = N(st([G(vol, dev), G(sh, dev)]))
    assert bits_equal(out[:, :, :, :-3], vol[:, :, :, 3:]) and bits_equal(out[:, :, :, -1], vol[:, :, :, -1])
    # a channel permutation commutes with interpolation
    trf = rng.normal(0, 3, (1,) + S + (3,)).astype(F)
    perm = rng.permutation(32)
    a = N(st([G(vol[..., perm], dev), G(trf, dev)]))
    b = N(st([G(vol, dev), G(trf, dev)]))[..., perm]
    assert bits_equal(a, b)
    # inputs are never written
    v = G(vol, dev)
    t = G(trf, dev)
    st([v, t])
    assert bits_equal(N(v), vol) and bits_equal(N(t), trf)


def test_backward_is_loud(dev):
    """ops without a backward kernel raise in backward instead of returning a silent zero gradient (the reference's hard Dice is
    not differentiable either, metrics.py:454-458); linear and nearest interpolation have one (tests/test_gpu_backward.py)"""
    vol = torch.randn(1, 6, 6, 6, 4, device=dev, requires_grad=True)
    trf = torch.zeros(1, 6, 6, 6, 3, device=dev)
    for method in ('nearest', 'linear'):
        vol.grad = None
        ne.layers.SpatialTransformer(interp_method=method)([vol, trf]).sum().backward()
        assert vol.grad is not None and torch.allclose(vol.grad, torch.ones_like(vol.grad))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        d = ne.metrics.HardDice(nb_labels=4, input_type='prob').dice(torch.softmax(vol, -1), torch.softmax(vol, -1))
    if d.requires_grad:
        with pytest.raises(NotImplementedError, match='backward'):
            d.sum().backward()


def test_wave_cache_warp_variant(dev):
    """nrt_interpn_f32 variant 10: the wave-private LDS row cache kernel (csrc/fused_wc.h) as a stand-alone warp, in the three location
    modes (SpatialTransformer, interpn, Resize), with and without fill, coherent and incoherent fields, ragged shapes."""
    rng = np.random.default_rng(1010)
    B, S = 5, (17, 61, 67)
    vol = rng.standard_normal((B,) + S + (32,)).astype(F)
    for name, trf in (('smooth', np.stack([N(synth.smooth_displacement(3 + b, 67, sigma=2.5, coarse=5, device='cpu'))[:17, :61, :67] for b in range(B)])),
                      ('iid', rng.normal(0, 1.5, (B,) + S + (3,)).astype(F)), ('far', rng.uniform(-70, 70, (B,) + S + (3,)).astype(F))):
        trf = np.ascontiguousarray(trf, F)
        for fill in (None, -1.5):
            st = ne.layers.SpatialTransformer(fill_value=fill)
            st._variant = 10
            got = N(st([G(vol, dev), G(trf, dev)]))
            assert bits_equal(got, npo.spatial_transformer(vol, trf, fill_value=fill)), (name, fill)
    # absolute locations through the un-batched API (one volume: enough patches for the schedule)
    S1 = (16, 92, 180)
    v1 = rng.standard_normal(S1 + (32,)).astype(F)
    loc = (ijk(S1) + rng.normal(0, 0.8, S1 + (3,))).astype(F)
    got = N(ne.utils.interpn(G(v1, dev), G(loc, dev), fill_value=0.0, _variant=10))
    assert bits_equal(got, npo.interpn(v1, loc, 'linear', 0.0))
    # regular grid (Resize: tf.linspace locations)
    half = rng.standard_normal((5, 8, 30, 32, 32)).astype(F)
    from neurite_amd import utils as U, _lib as L
    got = U._interp_op(G(half, dev), None, [16, 60, 64], L.LOC_LINSPACE, U._METHODS['linear'], None, batched=True, variant=10)
    assert bits_equal(N(got), npo.resize_layer(half, 2))
    # shapes the schedule does not take are refused, not silently routed elsewhere
    with pytest.raises(ne.errors.NeuriteAmdError):
        ne.utils.interpn(G(v1[:8, :8, :8], dev), G(loc[:4, :4, :4], dev), _variant=10)


@pytest.mark.parametrize('X', [16, 17, 18, 19, 20, 21])
def test_wave_cache_warp_three_states_every_march_length(dev, X):
    """The stand-alone warp keeps THREE pass states (fused_wc.h: NST; row requests two steps ahead of their use): marches of 3 k, 3 k + 1
    and 3 k + 2 planes -- whole columns and the pieces the schedule cuts them into -- against the oracle, bit for bit, with fill and with
    locations far outside (the tail steps prepare passes beyond the end that must never be blended)"""
    rng = np.random.default_rng(300 + X)
    B, S = 4, (X, 64, 64)
    vol = rng.standard_normal((B,) + S + (32,)).astype(F)
    trf = rng.normal(0, 1.2, (B,) + S + (3,)).astype(F)
    trf[1] = rng.uniform(-40, 40, S + (3,)).astype(F)                     # one entry incoherent: lists longer than 32, then the fallback
    trf[2, X // 2:] = 0                                                   # exactly on the grid from the middle on
    for fill in (None, 0.5):
        st = ne.layers.SpatialTransformer(fill_value=fill)
        st._variant = 10
        got = N(st([G(vol, dev), G(trf, dev)]))
        assert bits_equal(got, npo.spatial_transformer(vol, trf, fill_value=fill)), (X, fill)


def test_full_size_cfg2_spatial_transformer(dev):
    """BASELINE config 2 size: 160^3 x 32-label one-hot, smooth and worst-case fields, vs the C oracle."""
    mov, _, trf = synth.cfg2_batch(1, 160, 32, device=dev, seed0=1)
    mov_h, trf_h = N(mov)[0], N(trf)[0]
    st = ne.layers.SpatialTransformer()
    got = N(st([mov, trf]))[0]
    want = co.interpn(mov_h, trf_h, 'linear', None, loc_mode=1)
    assert bits_equal(got, want)
    # warped one-hot stays a partition of unity up to rounding
    assert np.abs(got.sum(-1) - 1).max() < 1e-5
    for variant, tune in ((2, 0), (3, 0), (3, 40), (4, 40), (5, 0), (5, _tile(3, 3, 3, 1)), (10, 0)):
        st._variant, st._tune = variant, tune
        assert bits_equal(N(st([mov, trf]))[0], want), (variant, tune)
    gotn = N(ne.layers.SpatialTransformer('nearest', fill_value=0)([mov, trf]))[0]
    assert bits_equal(gotn, co.interpn(mov_h, trf_h, 'nearest', 0.0, loc_mode=1))
    rough = synth.rough_displacement(7, 160, device=dev)
    st._variant, st._tune = 0, 0
    gotr = N(st([mov, rough[None]]))[0]
    assert bits_equal(gotr, co.interpn(mov_h, N(rough), 'linear', None, loc_mode=1))
    # C = 1 image at full size (20 B/voxel path)
    img = torch.randn(1, 160, 160, 160, 1, device=dev)
    goti = N(st([img, trf]))[0]
    assert bits_equal(goti, co.interpn(N(img)[0], trf_h, 'linear', None, loc_mode=1))


# ------------------------------------------------------------------------------------------- resize
def test_resize_golden_bit_exact(dev):
    cases = golden_cases(load_golden('resize_small'))
    for tag in ('x2', 'half', 'aniso', 'x3_nearest', 'mixed1'):
        c = cases[tag]
        z = c['zoom'].tolist()
        z = z[0] if len(z) == 1 else z
        out = N(ne.utils.resize(G(c['vol'], dev), z, str(c['method'])))
        assert bits_equal(out, c['out']), tag
    out = ne.layers.Resize(2)(G(cases['layer']['x'], dev))
    assert tuple(out.shape) == tuple(cases['layer']['out_shape'])
    assert bits_equal(N(out), cases['layer']['out'])
    assert bits_equal(N(ne.layers.Zoom([0.5, 1.5])(G(cases['layer2d']['x'], dev))), cases['layer2d']['out'])


def test_resize_deformation_upsample_full_size(dev):
    """Resize(2) of an 80^3 x 3 deformation field -> 160^3 x 3 (neurite/tf/models.py:804)."""
    rng = np.random.default_rng(31)
    x = rng.standard_normal((2, 80, 80, 80, 3)).astype(F)
    got = N(ne.layers.Resize(2)(G(x, dev)))
    tabs = [npo.tf_linspace(0., 79., 160)] * 3
    for b in range(2):
        assert bits_equal(got[b], co.interpn(x[b], tabs, 'linear', loc_mode=2))
    # C = 32 goes through the vectorised kernels; zoom 0.5 and nearest
    y = rng.standard_normal((1, 24, 20, 28, 32)).astype(F)
    for z, method in ((2, 'linear'), (0.5, 'linear'), ([1.5, 2, 0.75], 'nearest')):
        got = N(ne.layers.Resize(z, method)(G(y, dev)))[0]
        assert bits_equal(got, npo.resize(y[0], z, method)), (z, method)


# ----------------------------------------------------------------------------------------------------------------------
# volume dtypes and ranks beyond float32 1-3-D (csrc/interpn_any.hip; neurite/tf/utils/utils.py:106-127, 137-213 are generic)
# ----------------------------------------------------------------------------------------------------------------------

_TORCH_DT = {np.dtype(np.float16): torch.float16, np.dtype(np.float64): torch.float64, np.dtype(np.float32): torch.float32,
             np.dtype(np.int32): torch.int32}


def test_interpn_dtypes_and_ranks_golden_bit_exact(dev):
    """float16 / float64 volumes (arithmetic in the volume dtype, loc cast to it) and 4-D / 5-D volumes against the fixtures
    the reference's own interpn produced"""
    cases = golden_cases(load_golden('interpn_dtypes'))
    assert len(cases) >= 25
    for tag, c in cases.items():
        fill = float(c['fill']) if bool(c['hasfill']) else None
        out = ne.utils.interpn(G(c['vol'], dev), G(c['loc'], dev), str(c['method']), fill)
        assert out.dtype == _TORCH_DT[c['vol'].dtype], tag
        assert tuple(out.shape) == c['out'].shape, tag
        assert np.array_equal(N(out), c['out'], equal_nan=True), tag


def _bf(a):
    return npo.round_bf16(np.asarray(a, F))


def test_interpn_bfloat16_vs_emulated_oracle(dev):
    """bfloat16 volumes: NumPy has no bfloat16, the oracle emulates it on float32 arrays with a rounding after every op
    (oracle/np_oracle.py: interpn_emulated -- the same emulation with binary16 rounding reproduces the reference-generated
    float16 fixtures, tests/test_oracle.py).  Bit-exact."""
    rng = np.random.default_rng(91)
    for S, Cc in (((9, 7, 11), 3), ((6, 5, 7), 32), ((13, 5), 4), ((17,), 2), ((3, 4, 3, 5), 2)):
        D = len(S)
        vol = _bf(rng.standard_normal(S + (Cc,)))
        osh = (6, 5, 4, 3)[:D]
        loc = rng.uniform(-2, max(S) + 1, osh + (D,)).astype(F)
        for method in ('linear', 'nearest'):
            for fill in (None, 0.3):
                got = ne.utils.interpn(G(vol, dev).to(torch.bfloat16), G(loc, dev), method, fill)
                assert got.dtype == torch.bfloat16
                ref = npo.interpn_emulated(vol, loc, method, fill, rnd=npo.round_bf16)
                assert np.array_equal(N(got.to(torch.float32)), ref, equal_nan=True), (S, Cc, method, fill)
    # a one-hot label map warps exactly like its float32 original wherever the weights are bfloat16 numbers
    lab = rng.integers(0, 8, (10, 9, 8))
    oh = np.eye(8, dtype=F)[lab]
    loc = (ijk((10, 9, 8)) + rng.integers(-2, 3, (10, 9, 8, 3)) * 0.5).astype(F)
    a = N(ne.utils.interpn(G(oh, dev).to(torch.bfloat16), G(loc, dev)).to(torch.float32))
    b = N(ne.utils.interpn(G(oh, dev), G(loc, dev)))
    assert np.array_equal(a, b)


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16, torch.float64])
def test_spatial_transformer_and_resize_in_other_dtypes(dev, dt):
    """SpatialTransformer (shift + identity grid formed in float32, then cast to the volume dtype by interpn) and Resize
    (tf.linspace grid in float32) on float16 / bfloat16 / float64 volumes against the oracle's transform / resize"""
    rng = np.random.default_rng(17)
    S, Cc = (8, 9, 7), 5
    vol32 = rng.standard_normal((2,) + S + (Cc,)).astype(F)
    trf = rng.normal(0, 1.5, (2,) + S + (3,)).astype(F)
    if dt == torch.bfloat16:
        vol_np = _bf(vol32)
        ref_st = np.stack([npo.interpn_emulated(vol_np[b], ijk(S) + trf[b], 'linear', None) for b in range(2)])
        from oracle.np_oracle import tf_linspace
        grid = np.stack(np.meshgrid(*[tf_linspace(0., s - 1., 2 * s) for s in S], indexing='ij'), -1).astype(F)
        ref_rs = np.stack([npo.interpn_emulated(vol_np[b], grid, 'linear', None) for b in range(2)])
        vt = G(vol_np, dev).to(dt)
        cmp = lambda t: N(t.to(torch.float32))
    else:
        npdt = np.float16 if dt == torch.float16 else np.float64
        vol_np = vol32.astype(npdt)
        ref_st = npo.spatial_transformer(vol_np, trf)
        ref_rs = npo.resize_layer(vol_np, 2)
        vt = G(vol_np, dev)
        cmp = N
    out = ne.layers.SpatialTransformer()([vt, G(trf, dev)])
    assert out.dtype == dt
    assert np.array_equal(cmp(out), ref_st, equal_nan=True)
    out = ne.layers.Resize(2)(vt)
    assert out.dtype == dt and tuple(out.shape) == (2, 16, 18, 14, Cc)
    assert np.array_equal(cmp(out), ref_rs, equal_nan=True)


def test_interpn_seven_and_eight_dimensional_volumes(dev):
    """the reference's corner loop is itertools.product([0, 1], repeat=D) for any D (utils.py:159); TensorFlow itself stops at rank-8
    tensors.  7- and 8-D float32 / float64 volumes against the oracle, linear (128 / 256 corners per output) and nearest."""
    rng = np.random.default_rng(78)
    for S in ((3, 2, 3, 2, 3, 2, 3), (2, 3, 2, 2, 3, 2, 2, 3)):
        D = len(S)
        for dt, tdt in ((F, torch.float32), (np.float64, torch.float64)):
            vol = rng.standard_normal(S + (2,)).astype(dt)
            loc = rng.uniform(-0.7, max(S) - 0.3, (11, D)).astype(dt)
            for method in ('linear', 'nearest'):
                for fill in (None, 0.5):
                    got = N(ne.utils.interpn(G(vol, dev), G(loc, dev), interp_method=method, fill_value=fill))
                    ref = npo.interpn(vol, loc, method, fill)
                    assert got.dtype == dt and np.array_equal(got, ref, equal_nan=True), (D, dt, method, fill)


def test_interpn_rank_limit_is_stated(dev):
    v = torch.zeros((2,) * 9 + (1,), device=dev)
    with pytest.raises(NotImplementedError, match='1- to 8-D'):
        ne.utils.interpn(v, torch.zeros(3, 9, device=dev))
