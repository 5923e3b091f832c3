#!/usr/bin/env python
"""
bench.py -- headline benchmark of the neurite hot path on MI355X.

Metric (BASELINE.json): Mvoxels/sec of interpn+Dice on 160^3 x 32-label volumes.
A step = one pass of the hot path over one batch of synthetic volumes already resident in HBM:
    dice = SpatialTransformer('linear')([moving, trf])  ->  Dice(fixed, warped)   [B, L]
    N > 1: one RCCL all-reduce of [sum of dice, count] (mean Dice over the global batch); the collective of step k runs
           on RCCL's stream while step k + 1's kernels run, every mean is collected before the closing synchronize
written as a user of the reference writes it -- two calls with the reference's signatures.  When nothing needs a gradient
SpatialTransformer defers the warp and Dice launches the fused kernel on (moving, trf, fixed) (neurite_amd/deferred.py; the
warped volume is consumed in registers, never written -- SURVEY.md 8d anticipates this form).  --direct times the same kernel
through neurite_amd.fused.warp_dice, --unfused the eager two-kernel pipeline (deferral off: `warped` written and read back).
The JSON line always carries all three (`fused_pipeline` = the timed one, `fused_direct_pipeline`, `dropin_pipeline`) plus
`roofline_dropin` (the stand-alone interpn kernel), `config2_batch1`, `bf16_storage` and `default_args_pipeline` (the reference's
default range asserts on), measured in the same process.
Workload: BASELINE config 2 / 4 (SpatialTransformer + Dice, 160^3 x 32 one-hot, fp32).  N = 1: a step is 4 volumes (config 4's
per-GPU share at N = 8); the line also carries `value_strong_b32`, all 32 volumes of config 4 on the one GPU.  N > 1: the parsed line
is BASELINE config 4 AS WRITTEN -- global batch 32 fixed, 32 / N volumes per rank, "scaling": "strong" ("total voxels/s =
32 V / wall-time at world sizes 1, 2, 4, 8") -- and the weak run (`--batch-per-gpu` volumes on every rank) is the extra key
`weak_per_gpu`; `--weak` swaps the two.  1 voxel = 1 spatial output location.

    python bench.py                       # 1 GPU
    python bench.py --gpus N              # N GPUs of this node: re-launches itself under torch.distributed.run, one rank per GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W            # the same, launched by the driver
    python bench.py --gpus N --weak       # N > 1 with 4 volumes per GPU whatever N is ("scaling": "weak") as the parsed line
    python bench.py --global-batch 32     # N = 1 with all 32 volumes of config 4 as the parsed line

Steps are INDEPENDENT (each warps and scores its own batch), so they are issued round-robin over `--streams` HIP streams (default 3): the
first blocks of step k + 1's gather run on the CUs step k's last round has left, the ~25 us of small second-stage kernels of a step run
beside the next gather -- what a data-parallel host does with independent work, and what the all-reduce overlap above already did for the
collective.  Every result is bit-identical to serial steps (tools/two_stream_probe.py).  `--streams 1` issues them strictly one after
the other.  In front of every measurement, BEFORE its --warmup steps, the step runs untimed for `--prewarm-ms` (250 ms): a fresh device
needs 100-200 ms of continuous work to reach steady clocks; the timed region is exactly --steps steps.

Rank 0 prints ONE JSON line on stdout.  `roofline` is the dominant kernel: achieved = algorithmic bytes per launch / the time a launch
costs, measured with HIP events over the timed region -- with pipelined steps that is the device time of the region / launches (the rate at
which launches complete; a launch's own start-to-end time, `kernel_own_duration_ms`, overlaps its neighbours'), and `roofline.isolated_launch`
carries the same launch with the device to itself (serial steps: the figure rocprofv3 --stats of a `--streams 1` run reproduces).  Algorithmic bytes per
voxel: fused kernel 4C (moving row) + 12 (shift) + 4L (fixed row) = 268 B at C=L=32;
unfused interpn 4C + 12 + 4C = 268 B, Dice 2*4L = 256 B.  `cpu_baseline` is the C oracle (a port of the
reference algorithm, oracle/oracle.c) on the host cores this process may use, over a bounded sample, plus BASELINE config 1
(one 32^3 volume) on the NumPy restatement, a torch-CPU form and the C port -- reported, not a target.
At N = 1 the line also carries `unet_fwd` (BASELINE config 3 forward, per-layer fraction of the fp32 MFMA peak), `lc3d_wcce`
(config 5) and `training` (registration step = fused warp+Dice forward + backward on the bench volumes; unet training step);
at N > 1 `unet_fwd` is the slowest rank's forward with one volume per GPU.
"""

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
INTERPN_BYTES_PER_VOXEL = lambda C, D: 4 * C + 4 * D + 4 * C      # read row once + loc + write row
DICE_BYTES_PER_VOXEL = lambda L: 2 * 4 * L


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch-per-gpu', type=int, default=4)
    ap.add_argument('--global-batch', type=int, default=0,
                    help='strong scaling (BASELINE config 4: 32): the global batch is fixed and split over the ranks; 0 = weak scaling '
                         'with --batch-per-gpu volumes on every GPU')
    ap.add_argument('--weak', action='store_true',
                    help='N > 1 only: make the weak-scaling run (--batch-per-gpu volumes on every rank) the parsed line; by default an N > 1 '
                         'run is BASELINE config 4 as written (global batch 32, 32 / N per rank, "scaling": "strong") and carries the weak '
                         'figure as `weak_per_gpu`')
    ap.add_argument('--no-strong', action='store_true',
                    help='skip the second scaling mode of the run (cfg4_strong at N = 1, weak_per_gpu at N > 1)')
    ap.add_argument('--stub-step', action='store_true',
                    help='CPU-only self-test of the launch / timing / JSON plumbing: gloo ranks, a stub step instead of the kernels '
                         '(tests/test_distributed_cpu.py); never a measurement')
    ap.add_argument('--size', type=int, default=160)
    ap.add_argument('--labels', type=int, default=32)
    ap.add_argument('--rough', action='store_true', help='worst-case incoherent field U(-80,80)')
    ap.add_argument('--variant', type=int, default=0, help='interpn kernel variant (0 = library default)')
    ap.add_argument('--tune', type=int, default=0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--sweep', action='store_true', help='time every interpn kernel variant and exit')
    ap.add_argument('--unet', action='store_true', help='only run the unet forward benchmark (BASELINE config 3)')
    ap.add_argument('--no-unet', action='store_true', help='skip the unet forward measurement in the default run')
    ap.add_argument('--unfused', action='store_true', help='time the eager two-kernel pipeline (deferred warps off) instead')
    ap.add_argument('--direct', action='store_true',
                    help='time neurite_amd.fused.warp_dice called directly instead of the reference-signature calls (same kernel)')
    ap.add_argument('--no-batch1', action='store_true',
                    help='skip the extra batch = 1 runs (profiling: every launch of the gather kernels then has the headline shape)')
    ap.add_argument('--streams', type=int, default=3,
                    help='independent steps are issued round-robin on this many HIP streams, so that the head of step k + 1 fills the tail of '
                         'step k and the small second-stage kernels of a step run beside the next gather (1: strictly serial steps; the '
                         'line always carries the serial figures as roofline.isolated_launch)')
    ap.add_argument('--prewarm-ms', type=float, default=250.0,
                    help='untimed device activity (the step itself, repeated) in front of every measurement, BEFORE its --warmup steps: a '
                         'fresh device needs 100-200 ms of continuous work to reach its steady clocks (the same 0.22 ms step ran 0.266 ms in '
                         'its first 40 repetitions and 0.223 ms from the 500th on, tools/lab/b1_dbg.py); 0 = none.  The timed region stays '
                         'EXACTLY --steps steps between barrier + synchronize pairs')
    ap.add_argument('--graph', action='store_true',
                    help="capture a step's compute launches (gather + Dice second stage + mean pair) in one hipGraph")
    return ap.parse_args()


PREWARM_MS = 0.0             # set from --prewarm-ms in main()


def usable_cores():
    """host threads this process may really use: the affinity mask and the cgroup CPU quota, not just os.cpu_count()"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:   # noqa
        pass
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
        except Exception:   # noqa
            pass
    return n


def cpu_baseline(mov, fix, trf, budget_s=12.0):
    """C oracle (port of the reference algorithm, oracle/oracle.c) on all host cores over a bounded sample of the workload.
    The output volume is allocated and touched ONCE before the clock starts: a fresh 524 MB buffer per repetition made
    the round-1 figure a measure of first-touch page faults, not of the algorithm."""
    from oracle import c_oracle as co
    m, f, t = mov[0].cpu().numpy(), fix[0].cpu().numpy(), trf[0].cpu().numpy()
    cores = usable_cores()
    V = int(np.prod(m.shape[:-1]))
    warped = np.zeros(t.shape[:-1] + (m.shape[-1],), np.float32)        # pre-touched

    def once():
        w = co.interpn(m, t, 'linear', None, loc_mode=1, out=warped)
        sums, _ = co.dice_sums(f[None], w[None])
        return co.dice_from_sums(sums)

    # the OpenMP port does not scale to every hardware thread of a 2-socket host (NUMA, SMT): try a few team sizes inside
    # the time budget and report the best, with the thread count it was measured at
    tried, best = {}, None
    cand = sorted({c for c in (8, 32, 64, 128, cores) if c <= cores} | {cores})
    d = None
    for nt in cand:
        co.set_num_threads(nt)
        d = once()                                  # warm (first touch of this team's pages)
        t0 = time.perf_counter()
        once()
        t1 = time.perf_counter() - t0
        reps = int(max(1, min(10, budget_s / len(cand) / max(t1, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(reps):
            once()
        dt = (time.perf_counter() - t0) / reps
        tried[str(nt)] = round(V / dt / 1e6, 3)
        if best is None or dt < best[0]:
            best = (dt, nt, reps)
    dt, nt, reps = best
    return {'value': round(V / dt / 1e6, 3), 'unit': 'Mvoxels/s', 'cores': nt, 'kind': 'port',
            'host': {'os_cpu_count': os.cpu_count(), 'usable_cores': cores},
            'by_threads': tried,
            'sample': '%d x (SpatialTransformer linear + Dice on one %s x %d-label volume of the bench batch), '
                      'C oracle with OpenMP at %d threads (best of the team sizes in by_threads), output buffer pre-touched'
                      % (reps, 'x'.join(str(s) for s in m.shape[:-1]), m.shape[-1], nt)}, d


def cpu_cfg1(budget_s=6.0):
    """BASELINE config 1 -- "interpn linear warp of one 32^3 fp32 volume on CPU (reference path)" -- with the inputs of
    SURVEY.md 8(d): the op-for-op NumPy restatement of neurite/tf/utils/utils.py:73-220 on ONE thread, a torch-CPU
    vectorised form on all host cores, and the C/OpenMP port.  TensorFlow is not installed, so "reference path" means
    these restatements (all three are pinned to the fixture the reference's own source produced, tests/test_oracle.py)."""
    from oracle import c_oracle as co
    from oracle import np_oracle as npo
    from oracle import torch_cpu as tco
    rng = np.random.default_rng(0)
    vol = rng.standard_normal((32, 32, 32)).astype(np.float32)
    ijk = np.stack(np.meshgrid(*[np.arange(32)] * 3, indexing='ij'), -1).astype(np.float32)
    loc = (ijk + rng.normal(0, 3, (32, 32, 32, 3)).astype(np.float32)).astype(np.float32)
    V = 32 ** 3
    cores = usable_cores()
    few = min(8, cores)               # 32^3 is 128 KB: a 256-thread team costs more to wake than the work takes

    def rate(fn):
        fn()
        n, t0 = 0, time.perf_counter()
        while True:
            fn()
            n += 1
            dt = time.perf_counter() - t0
            if dt > budget_s / 5 or n >= 2000:
                return round(V * n / dt / 1e6, 3), n

    ref = npo.interpn(vol, loc)
    vt, lt = torch.from_numpy(vol), torch.from_numpy(loc)
    old = torch.get_num_threads()
    torch.set_num_threads(cores)
    try:
        same_t = bool(np.array_equal(tco.interpn_linear(vt, lt).numpy(), ref))
        r_t, n_t = rate(lambda: tco.interpn_linear(vt, lt))
    finally:
        torch.set_num_threads(old)
    torch.set_num_threads(few)
    try:
        r_t8, n_t8 = rate(lambda: tco.interpn_linear(vt, lt))
    finally:
        torch.set_num_threads(old)
    r_np, n_np = rate(lambda: npo.interpn(vol, loc))
    out = np.zeros((32, 32, 32, 1), np.float32)
    co.set_num_threads(cores)
    same_c = bool(np.array_equal(co.interpn(vol[..., None], loc, out=out)[..., 0], ref))
    r_c, n_c = rate(lambda: co.interpn(vol[..., None], loc, out=out))
    co.set_num_threads(few)
    r_c8, n_c8 = rate(lambda: co.interpn(vol[..., None], loc, out=out))
    return {'workload': 'BASELINE config 1: interpn linear, one 32^3 fp32 volume, loc = grid + N(0, 3), seed 0', 'unit': 'Mvoxels/s',
            'numpy_1thread': {'value': r_np, 'cores': 1, 'reps': n_np},
            'torch_cpu_allcores': {'value': r_t, 'cores': cores, 'reps': n_t, 'bit_identical_to_numpy': same_t},
            'torch_cpu_%dthreads' % few: {'value': r_t8, 'cores': few, 'reps': n_t8},
            'c_openmp_allcores': {'value': r_c, 'cores': cores, 'reps': n_c, 'bit_identical_to_numpy': same_c},
            'c_openmp_%dthreads' % few: {'value': r_c8, 'cores': few, 'reps': n_c8}}


def sweep(args, dev, mov, fix, trf):
    """Time every interpn kernel variant (and the Dice kernel) on the bench batch; JSON lines to stderr."""
    import neurite_amd as ne
    B, S = mov.shape[0], args.size
    V = S ** 3
    res = []
    def T(lx, ly, lz, zo):
        return lx | (ly << 4) | (lz << 8) | (zo << 12)
    cfgs = [(1, 0), (2, 1), (2, 2), (2, 4), (3, 40), (3, 20), (3, 10)]
    for zo in (0, 1):
        cfgs += [(5, T(0, 0, 5, zo)), (5, T(1, 1, 3, zo)), (5, T(1, 1, 5, zo)), (5, T(2, 2, 3, zo)), (5, T(2, 2, 4, zo)),
                 (5, T(2, 3, 4, zo)), (5, T(3, 3, 3, zo)), (5, T(3, 3, 4, zo)), (5, T(2, 2, 5, zo)), (5, T(3, 2, 4, zo)),
                 (5, T(4, 4, 3, zo))]
    if os.environ.get('NRT_SWEEP'):
        cfgs = [tuple(int(v) for v in c.split(':')) for c in os.environ['NRT_SWEEP'].split(',')]
    ref = None
    for variant, tune in cfgs:
        st = ne.layers.SpatialTransformer()
        st._variant, st._tune = variant, tune
        try:
            out = st([mov, trf])
        except Exception as e:      # noqa
            log('variant', variant, tune, 'failed:', e)
            continue
        if ref is None:
            ref = out
        same = bool(torch.equal(out, ref))
        for _ in range(2):
            st([mov, trf])
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        n = 10
        e0.record()
        for _ in range(n):
            st([mov, trf])
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        gbs = INTERPN_BYTES_PER_VOXEL(args.labels, 3) * V * B / ms / 1e6
        r = {'kernel': 'interpn', 'variant': variant, 'tune': tune,
             'tile': [1 << (tune & 15), 1 << ((tune >> 4) & 15), 1 << ((tune >> 8) & 15), (tune >> 12) & 1] if variant == 5 else None,
             'ms': round(ms, 4), 'GBs': round(gbs, 1),
             'frac': round(gbs / HBM_PEAK_GBS, 4), 'bit_identical_to_first': same}
        log(json.dumps(r))
        res.append(r)
    D = ne.metrics.Dice(check_input_limits=False)
    warped = ref
    for _ in range(2):
        D.dice(fix, warped)
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(10):
        D.dice(fix, warped)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gbs = DICE_BYTES_PER_VOXEL(args.labels) * V * B / ms / 1e6
    r = {'kernel': 'dice_soft', 'ms': round(ms, 4), 'GBs': round(gbs, 1), 'frac': round(gbs / HBM_PEAK_GBS, 4)}
    log(json.dumps(r))
    res.append(r)
    # fused SpatialTransformer+Dice, tile shapes
    def PM(lz, zo):
        return (1 << 13) | (zo << 12) | (lz << 16)
    ftunes = (0, T(3, 3, 3, 0), T(3, 3, 4, 0), PM(10, 0), PM(20, 0), PM(40, 0), PM(160, 0), PM(20, 1), PM(40, 1))
    if os.environ.get('NRT_FUSED_SWEEP'):
        ftunes = tuple(int(v) for v in os.environ['NRT_FUSED_SWEEP'].split(','))
    for tune in ftunes:
        for _ in range(2):
            d = ne.fused.warp_dice(mov, trf, fix, _tune=tune)
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(10):
            ne.fused.warp_dice(mov, trf, fix, _tune=tune)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        r = {'kernel': 'fused_warp_dice', 'tune': tune,
             'tile': (['plane-major 4x8', 'LZ', tune >> 16, (tune >> 12) & 1] if (tune >> 13) & 1 else
                      [1 << (tune & 15), 1 << ((tune >> 4) & 15), 1 << ((tune >> 8) & 15), (tune >> 12) & 1]),
             'ms': round(ms, 4), 'Mvox_s': round(V * B / ms / 1e3, 1),
             'GBs_524': round((INTERPN_BYTES_PER_VOXEL(args.labels, 3) + DICE_BYTES_PER_VOXEL(args.labels)) * V * B / ms / 1e6, 1),
             'max_abs_diff_vs_unfused': float((d - D.dice(fix, warped)).abs().max())}
        log(json.dumps(r))
        res.append(r)
    # how fast is the gather when re-use is perfect?  zero / constant half-voxel displacement
    if os.environ.get('NRT_SWEEP_COHERENT', '1') == '1':
        for name, val in (('zero', 0.0), ('half', 0.5)):
            tr0 = torch.full_like(trf, val)
            for variant, tune in ((2, 1), (3, 20), (5, T(2, 2, 4, 0)), (5, T(3, 3, 3, 0))):
                st = ne.layers.SpatialTransformer()
                st._variant, st._tune = variant, tune
                for _ in range(2):
                    st([mov, tr0])
                e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
                e0.record()
                for _ in range(10):
                    st([mov, tr0])
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 10
                gbs = INTERPN_BYTES_PER_VOXEL(args.labels, 3) * V * B / ms / 1e6
                r = {'kernel': 'interpn_' + name + '_shift', 'variant': variant, 'tune': tune, 'ms': round(ms, 4),
                     'GBs': round(gbs, 1), 'frac': round(gbs / HBM_PEAK_GBS, 4)}
                log(json.dumps(r))
                res.append(r)
            for tune in (0, 318784291):
                for _ in range(2):
                    ne.fused.warp_dice(mov, tr0, fix, _tune=tune)
                e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
                e0.record()
                for _ in range(10):
                    ne.fused.warp_dice(mov, tr0, fix, _tune=tune)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 10
                r = {'kernel': 'fused_' + name + '_shift', 'tune': tune, 'ms': round(ms, 4),
                     'frac_268': round(INTERPN_BYTES_PER_VOXEL(args.labels, 3) * V * B / ms / 1e6 / HBM_PEAK_GBS, 4)}
                log(json.dumps(r))
                res.append(r)
            del tr0
    dst = torch.empty_like(mov)
    # reference point: plain device-to-device copy of the same volume (achievable HBM rate on this box)
    for _ in range(2):
        dst.copy_(mov)
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(10):
        dst.copy_(mov)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    r = {'kernel': 'torch_copy_d2d', 'ms': round(ms, 4), 'GBs': round(2 * mov.numel() * 4 / ms / 1e6, 1)}
    log(json.dumps(r))
    res.append(r)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'sweep_%s.json' % ('rough' if args.rough else 'smooth')), 'w') as f:
        json.dump(res, f, indent=1)


MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X fp32 matrix peak (v_mfma_f32_16x16x4_f32), MI355X_MICROARCH.md


def warm_calls(fn, at_least):
    """`fn` at least `at_least` times and for at least PREWARM_MS of wall time (steady clocks: see --prewarm-ms), untimed"""
    t0, n = time.perf_counter(), 0
    while n < at_least or (time.perf_counter() - t0) * 1e3 < PREWARM_MS:
        fn()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return n


def unet_fwd_ms(dev, size=160, labels=32, reps=20, warmup=5):
    """median forward ms of the BASELINE config 3 unet on this rank's GPU (used for the N > 1 line of the metric)"""
    import contextlib
    import neurite_amd as ne
    torch.manual_seed(5)
    with contextlib.redirect_stdout(sys.stderr):
        model = ne.models.unet(16, (size, size, size, 1), 3, 3, labels, feat_mult=2).to(dev)
    x = torch.randn(1, size, size, size, 1, device=dev)
    warm_calls(lambda: model(x), warmup)
    times = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        model(x)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    return float(np.median(times))


def conv_layer_error(m, src, lo, up, out, nsamp=2048):
    """Element-wise error of one Conv3D layer of the timed network on `nsamp` sampled output voxels (all output channels): the
    layer's own input tensors, gathered and contracted in float64 with torch on the device -- a plain PyTorch reference of the same op,
    not the oracle.  Reported as tests/test_gpu_unet.py states the tolerance: `max_rel_err` over the well-conditioned outputs
    (sum of absolute terms <= 25 |result|, in the identity range of the activation) and the largest error of ANY sampled output in
    units of 2^-24 of the sum of its absolute terms (a float32 dot product: a few units whatever its length)."""
    import torch.nn.functional as TF
    from neurite_amd import models as nm
    g = torch.Generator(device='cpu')
    g.manual_seed(11)
    xin = src if lo is None else nm._upsample_concat(src, lo, tuple(up))
    X, Y, Z, C = xin.shape[1:]
    k = m.ksize3
    pb = [((kk - 1) * m.dilation) // 2 for kk in k]
    pe = [(kk - 1) * m.dilation - b0 for kk, b0 in zip(k, pb)]
    xp = TF.pad(xin[0], (0, 0, pb[2], pe[2], pb[1], pe[1], pb[0], pe[0]))
    ix = torch.randint(0, X, (nsamp,), generator=g).to(xin.device)
    iy = torch.randint(0, Y, (nsamp,), generator=g).to(xin.device)
    iz = torch.randint(0, Z, (nsamp,), generator=g).to(xin.device)
    taps = []
    for dx in range(k[0]):
        for dy in range(k[1]):
            for dz in range(k[2]):
                taps.append(xp[ix + dx * m.dilation, iy + dy * m.dilation, iz + dz * m.dilation, :])
    patches = torch.stack(taps, 1).reshape(nsamp, -1).double()
    w = m.kernel.detach().reshape(-1, m.cout).double()
    b = m.bias.detach().double()
    pre = patches @ w + b
    absref = patches.abs() @ w.abs() + b.abs()
    act = m.act
    want = torch.where(pre > 0, pre, torch.exp(pre) - 1) if act == 1 else (pre.clamp_min(0) if act == 2 else pre)
    got = out[0, ix, iy, iz, :].double()
    err = (got - want).abs()
    well = (absref <= 25.0 * pre.abs()) & ((pre > 0) | (act == 0))
    rel = float((err[well] / want[well].abs()).max()) if bool(well.any()) else None
    return {'max_rel_err': rel, 'max_rel_err_over': 'sampled outputs with sum|terms| <= 25 |result|: %.3f of %d' % (float(well.double().mean()), err.numel()),
            'max_err_in_2^-24_of_abs_sum': round(float((err / (absref * 2.0 ** -24)).max()), 2)}


def unet_bench(dev, size=160, labels=32, nb_conv_per_level=1, reps=20, warmup=5):
    """BASELINE config 3: unet(16, (160,160,160,1), 3, 3, nb_labels, feat_mult=2) forward on one fp32 volume.
    Returns total forward ms (median) and per-conv-layer time / TFLOP/s / fraction of the fp32 MFMA peak."""
    import neurite_amd as ne
    torch.manual_seed(5)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):          # the builder prints like the reference does; stdout is the JSON line
        model = ne.models.unet(16, (size, size, size, 1), 3, 3, labels, feat_mult=2,
                               nb_conv_per_level=nb_conv_per_level).to(dev)
    x = torch.randn(1, size, size, size, 1, device=dev)
    y = model(x)
    warm_calls(lambda: model(x), warmup)
    times = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        y = model(x)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    total = float(np.median(times))
    # per-layer timing: run the graph op by op with events (same kernels, same tensors)
    layers = []
    t = {}
    with torch.no_grad():
        for op in model.ops:
            kind, name = op['kind'], op['name']
            if kind != 'conv':
                continue
            m = model.layers_by_name[name]
            sp, cin = op['shape'][0], m.cin
            flops = 2.0 * sp[0] * sp[1] * sp[2] * m.ksize3[0] * m.ksize3[1] * m.ksize3[2] * m.cin * m.cout
            l = {'name': name, 'cin': m.cin, 'cout': m.cout, 'shape': list(sp), 'gflop': round(flops / 1e9, 2)}
            if op.get('lo') and tuple(op.get('up') or ()) == (2, 2, 2) and m.ksize3 == (3, 3, 3):
                # decoder form: the up-sampled channels run as 8 folded taps (nrt_conv3d_up2_f32); the fraction of the MFMA
                # peak below is priced on the matrix work that is EXECUTED, not on the 27-tap count
                c1 = [o for o in model.ops if o['name'] == op['lo']][0]['shape'][1]
                c0 = m.cin - c1
                from neurite_amd import _lib
                if _lib.lib().nrt_conv3d_up2_supported(c0, c1, m.cout, _lib.ints(list(sp))) == 1:
                    l['gflop_executed'] = round(2.0 * sp[0] * sp[1] * sp[2] * (27 * c0 + 8 * c1) * m.cout / 1e9, 2)
                    l['folded_upsampling'] = True
            layers.append(l)
    inter = model(x, return_tensors=[l['name'] for l in layers] + [o['name'] for o in model.ops if o['kind'] == 'maxpool'])
    for l in layers:
        op = model.ops[model.layer_names.index(l['name'])]
        m = model.layers_by_name[l['name']]
        src = x.reshape(1, size, size, size, 1) if op['src'].endswith('_input') else inter[op['src']]
        lo = inter[op['lo']] if op.get('lo') else None
        for _ in range(3):
            m(src, lo=lo, up=op.get('up'))
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(10):
            m(src, lo=lo, up=op.get('up'))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        l['ms'] = round(ms, 4)
        gfx = l.get('gflop_executed', l['gflop'])
        l['tflops'] = round(gfx / ms, 2)
        l['frac_of_fp32_mfma_peak'] = round(gfx / ms / MFMA_F32_PEAK_TFLOPS, 4)
        try:
            l.update(conv_layer_error(m, src, lo, op.get('up'), inter[l['name']]))
        except Exception as e:   # noqa
            l['max_rel_err'] = 'failed: %s' % e
    # matrix-core counters of a rocprofv3 --pmc pass of this forward (profiles/r05_unet/mfma_counters.json, written by
    # tools/unet_mfma_summary.py from SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_VALU_MFMA_MOPS_F32 / GRBM_GUI_ACTIVE), keyed by layer name:
    # a static figure of the profiled session next to the live flops / (time x peak), as the traffic figure of the gather is
    try:
        ctr = json.load(open(os.path.join(ROOT, 'profiles', 'r05_unet', 'mfma_counters.json'))).get('layers', {})
    except Exception:   # noqa
        ctr = {}
    for l in layers:
        c = ctr.get(l['name'])
        l['mfma_busy_from_counters'] = None if c is None else c['mfma_busy']
        if c is not None:
            l['counter_kernel'] = c['kernel']
    mf = [l for l in layers if l['cin'] >= 8]
    gf = sum(l.get('gflop_executed', l['gflop']) for l in mf)
    ms = sum(l['ms'] for l in mf)
    return {'config': 'BASELINE config 3: unet(16, (%d,%d,%d,1), 3, 3, nb_labels=%d, feat_mult=2, nb_conv_per_level=%d), fp32, batch 1'
                      % (size, size, size, labels, nb_conv_per_level),
            'fwd_ms': round(total, 3), 'fwd_ms_min': round(float(np.min(times)), 3), 'layers': layers,
            'mfma_layers': {'gflop': round(gf, 1), 'ms': round(ms, 3), 'tflops': round(gf / ms, 1),
                            'frac_of_fp32_mfma_peak': round(gf / ms / MFMA_F32_PEAK_TFLOPS, 4)}}


def lc3d_bench(dev, reps=10):
    """BASELINE config 5: LocallyConnected3D 3x3x3 on [1, 96, 96, 96, 16] bf16 (+ weighted CCE on the output)."""
    import neurite_amd as ne
    torch.manual_seed(6)
    x = torch.randn(1, 96, 96, 96, 16, device=dev, dtype=torch.bfloat16)
    layer = ne.layers.LocallyConnected3D(16, (3, 3, 3))
    with torch.no_grad():
        y = layer(x)
        layer.kernel.normal_(0, 1.0 / np.sqrt(432))
        lab = torch.randint(0, 16, (1, 94, 94, 94), device=dev)
        t = torch.nn.functional.one_hot(lab, 16).to(torch.bfloat16)
        w = np.linspace(0.5, 1.5, 16).astype(np.float32)
        cce = ne.metrics.WeightedCategoricalCrossentropy(label_weights=w, from_logits=True)
        for _ in range(2):
            y = layer(x)
            cce(t, y)
        e = [torch.cuda.Event(True) for _ in range(3)]
        e[0].record()
        for _ in range(reps):
            y = layer(x)
        e[1].record()
        for _ in range(reps):
            l = cce(t, y)
        e[2].record()
        torch.cuda.synchronize()
        # the same ten calls as ONE hipGraph replay: what the GPU spends on a call (one launch of the loss kernel + the division by N)
        # without the Python call path in between
        ms2_graph = None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                cce(t, y)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(reps):
                    lg = cce(t, y)
            g.replay()
            torch.cuda.synchronize()
            eg = [torch.cuda.Event(True) for _ in range(2)]
            eg[0].record()
            for _ in range(5):
                g.replay()
            eg[1].record()
            torch.cuda.synchronize()
            if abs(float(lg) - float(l)) <= 1e-6 * abs(float(l)):
                ms2_graph = eg[0].elapsed_time(eg[1]) / (5 * reps)
        except Exception as ex:   # noqa
            log('wcce graph timing failed: %s' % ex)
    ms = e[0].elapsed_time(e[1]) / reps
    ms2 = e[1].elapsed_time(e[2]) / reps
    nbytes = layer.kernel.numel() * 2 + x.numel() * 2 + y.numel() * 2 + layer.bias.numel() * 2
    return {'config': 'BASELINE config 5: LocallyConnected3D(16, 3x3x3) on [1,96,96,96,16] bf16 + weighted CCE',
            'lc3d_ms': round(ms, 4), 'algorithmic_GB': round(nbytes / 1e9, 3), 'GBs': round(nbytes / ms / 1e6, 1),
            'frac_of_hbm_peak': round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4),
            'wcce_ms': round(ms2, 4), 'wcce_GBs': round(2 * 2 * 16 * 94 ** 3 / ms2 / 1e6, 1),
            'wcce_ms_as_graph_replay': None if ms2_graph is None else round(ms2_graph, 4),
            'wcce_launches_per_call': 'one loss kernel (its last block adds the partials) + the division by N; eager calls are bound by the '
                                      'Python call path, the graph replay shows the GPU side',
            'loss': round(float(l), 5)}


def _timeit(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def registration_bench(mov, fix, trf, reps=5):
    """forward + backward of a registration step (-mean Dice of a warped one-hot map wrt the displacement field, fused
    kernels) on the bench's own volumes, so that its forward launches are the same launches as the timed ones."""
    import neurite_amd as ne
    B, size, labels = mov.shape[0], mov.shape[1], mov.shape[-1]

    def reg_step():
        f = trf.clone().requires_grad_()
        (-ne.fused.warp_dice(mov, f, fix).mean()).backward()
    reg_ms = _timeit(reg_step, reps)
    return {'what': 'fused warp+Dice forward + backward wrt the field, %d x %d^3 x %d' % (B, size, labels),
            'ms': round(reg_ms, 3), 'Mvoxels_per_s': round(B * size ** 3 / reg_ms / 1e3, 1)}


def unet_train_bench(dev, size=160, labels=32, reps=3):
    """unet segmentation training step (CCE - Dice, SGD update) at BASELINE config 3."""
    import contextlib
    import neurite_amd as ne
    with contextlib.redirect_stdout(sys.stderr):
        net = ne.models.unet(16, (size, size, size, 1), 3, 3, labels, feat_mult=2).to(dev)
    net.train()
    x = torch.randn(1, size, size, size, 1, device=dev)
    t = torch.nn.functional.one_hot(torch.randint(0, labels, (1, size, size, size), device=dev), labels).float()
    cce, dice = ne.losses.CategoricalCrossentropy(), ne.losses.Dice(check_input_limits=False)
    # neurite/tf/losses.py:225-246: the pair goes through one joint pass per direction (csrc/segloss.hip).  dice.loss is the [B, L]
    # negative Dice (losses.py:68-80); its mean is taken here so that the step holds no device->host read (mean_loss's finite check)
    seg_loss = ne.losses.multiple_losses_decorator([cce.loss, dice.loss])
    params = list(net.parameters())

    def sgd():
        with torch.no_grad():
            torch._foreach_add_(params, [p.grad for p in params], alpha=-1e-4)

    def step_with(loss_fn, update=sgd):
        def seg_step():
            for p in params:
                p.grad = None
            loss_fn(t, net(x)).mean().backward()
            update()
        return seg_step

    def sgd_per_tensor():
        with torch.no_grad():
            for p in params:
                p -= 1e-4 * p.grad
    out = {'what': 'BASELINE config 3 unet, forward + backward + SGD, batch 1; loss = multiple_losses_decorator([CCE, -Dice]).mean()'}
    out['ms'] = round(_timeit(step_with(seg_loss), reps), 3)
    out['ms_two_losses_separately'] = round(_timeit(step_with(lambda a, b: cce.loss(a, b) + dice.loss(a, b)), reps), 3)
    out['ms_round3_form'] = round(_timeit(step_with(lambda a, b: cce.loss(a, b) + dice.mean_loss(a, b), sgd_per_tensor), reps), 3)
    # the same step as ONE hipGraph launch: about 300 kernel launches of 4 us .. 0.8 ms, a third of them shorter than their own launch
    # cost, so the eager step is partly bound by the host
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step_with(seg_loss)()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for p in params:
            p.grad = None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss = seg_loss(t, net(x)).mean()
            loss.backward()
            sgd()
        graph.replay()
        first = float(loss)
        out['ms_as_one_hipgraph'] = round(_timeit(graph.replay, max(reps, 5)), 3)
        out['hipgraph_loss_first_and_last_replay'] = [round(first, 6), round(float(loss), 6)]
    except Exception as e:       # noqa
        out['ms_as_one_hipgraph'] = None
        out['hipgraph_error'] = str(e)[:200]
    return out




def prewarm(step, ms, dist=None, dev=None, sync=None):
    """`step` repeated for ~`ms` milliseconds of wall time, untimed; every pending mean is collected, the device is idle on return.
    With a process group every step carries a collective, so ALL ranks must run the same number of steps: a rank that stopped by its own
    clock while another went on for eight more would leave that one waiting in an all-reduce for ever.  The ranks therefore agree after
    every batch of steps (one all-reduce of a flag, MAX: all stop as soon as one has had its time) -- tests/test_distributed_cpu.py runs
    this with ranks that enter at different times."""
    if ms <= 0:
        return 0
    if sync is None:
        sync = torch.cuda.synchronize
    t0, n, pending = time.perf_counter(), 0, None
    while True:
        for _ in range(8):
            nxt = step(None)
            if pending is not None:
                pending.result()
            pending = nxt
            n += 1
        sync()
        done = (time.perf_counter() - t0) * 1e3 >= ms
        if dist is not None:
            flag = torch.tensor([1.0 if done else 0.0], dtype=torch.float32, **({'device': dev} if dev is not None else {}))
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            done = bool(float(flag[0]) > 0.0)
        if done:
            break
    if pending is not None:
        pending.result()
    sync()
    return n


def timed(step, steps, warmup, dist=None, dev=None, sparse_events=False):
    """
    The timed region of the bench contract: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by a barrier and a
    device synchronize on both sides; the wall time is the MAX over ranks.  `step(events)` enqueues one pass and returns a
    PendingMean (neurite_amd.distributed): the all-reduce of step k runs on RCCL's stream while step k + 1's kernels run
    on the compute stream; a step's global mean is collected (a stream-level wait, no host sync) after the next step has
    been enqueued, and every mean is complete before the closing synchronize, so all K steps' work lies inside the region.
    dev = None runs the same loop without a device (the gloo tests of this logic, tests/test_distributed_cpu.py).
    sparse_events: HIP events only around the first step and the last eight (pipelined steps: an event record is a barrier packet with a
    time stamp in its stream -- three per step cost a 0.22 ms step 25 % and a 0.86 ms step 4 %, tools/b1_pipeline_probe.py against this
    loop; the region's span needs its two ends only, k0 / k1 are then means over the last eight steps).
    Returns dict(elapsed = max over ranks [s], per_rank_s, ranks = size of the process group as the collective sees it,
    k0_ms / k1_ms = mean event intervals of a step's two kernel slots (NaN without a device), mean = last global mean).
    """
    on_gpu = dev is not None

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    # The host's garbage collector stays out of the measurement: a full collection of a process with torch loaded takes milliseconds to
    # tens of milliseconds.  Inside the timed region it leaves the device idle (one evidence session of round 6: 3.9 ms of a 16.6 ms region);
    # between the pre-warming and the timed steps it lets the clocks fall again (another session: every figure 13 % slow).  So: collect
    # FIRST, then pre-warm, warm up and time with the collector off; what the steps allocate is small and is collected afterwards.
    import gc
    gc.collect()
    gc_was_enabled = gc.isenabled()
    gc.disable()
    try:
        return _timed(step, steps, warmup, dist, dev, sparse_events, on_gpu, sync)
    finally:
        if gc_was_enabled:
            gc.enable()


def _timed(step, steps, warmup, dist, dev, sparse_events, on_gpu, sync):
    if on_gpu:
        prewarm(step, PREWARM_MS, dist, dev)
    pending, m = None, None
    for _ in range(warmup):
        nxt = step(None)
        if pending is not None:
            pending.result()
        pending = nxt
    if pending is not None:
        m = pending.result()
    pending = None
    sync()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] if (not sparse_events or k == 0 or k >= steps - 8) else None
           for k in range(steps)] if on_gpu else [None] * steps
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for k in range(steps):
        nxt = step(evs[k])
        if pending is not None:
            m = pending.result()
        pending = nxt
    if pending is not None:
        m = pending.result()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    per_rank, ranks = [elapsed], 1
    if dist is not None:
        world = dist.get_world_size()
        kw = {'device': dev} if on_gpu else {}
        mine = torch.tensor([elapsed], dtype=torch.float64, **kw)
        allt = [torch.zeros(1, dtype=torch.float64, **kw) for _ in range(world)]
        dist.all_gather(allt, mine)
        per_rank = [float(t[0]) for t in allt]
        ones = torch.ones(1, dtype=torch.float32, **kw)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)         # how many ranks the collective really spans
        ranks = int(round(float(ones[0])))
        elapsed = max(per_rank)
    if on_gpu:
        have = [e for e in evs if e is not None]
        k0 = float(np.mean([e[0].elapsed_time(e[1]) for e in have]))
        k1 = float(np.mean([e[1].elapsed_time(e[2]) for e in have]))
        # device time from the first step's first event to the LAST event of the region, per step: with steps pipelined over several
        # streams the kernels of consecutive steps overlap, k0 (a launch's own start-to-end time) then counts the shared stretches twice
        # and this -- the rate at which launches complete -- is the time a launch costs
        tail = have[-min(len(have), 8):]
        span = max(evs[0][0].elapsed_time(e[2]) for e in tail) / steps
    else:
        k0 = k1 = span = float('nan')
    return {'elapsed': elapsed, 'per_rank_s': per_rank, 'ranks': ranks, 'k0_ms': k0, 'k1_ms': k1, 'span_ms': span,
            'mean': None if m is None else float(m)}


class _Done:
    """a step result with nothing left to collect (keeps the step's output alive until the next step has been issued)"""

    def __init__(self, value=None):
        self.value = value

    def result(self):
        return None


class _OnStream:
    """a PendingMean whose collection runs on the stream its step was issued on"""

    def __init__(self, pending, stream):
        self.pending, self.stream = pending, stream

    def result(self):
        with torch.cuda.stream(self.stream):
            return self.pending.result()


_STEP_STREAMS = {}


def step_streams(nstreams, dev):
    """the HIP streams independent steps are spread over: created ONCE per process and shared by every pipelined measurement -- torch hands
    out pooled streams round-robin and the runtime maps streams onto a few hardware queues; a fresh pair per measurement ended up on ONE
    queue by the third pair (the batch-1 run lost all of its overlap: 0.277 ms per step where tools/b1_pipeline_probe.py measures 0.225)"""
    key = (str(dev), nstreams)
    if key not in _STEP_STREAMS:
        _STEP_STREAMS[key] = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    return _STEP_STREAMS[key]


def graph_pipelined(compute, nstreams, dev):
    """Small steps (one volume: ~0.22 ms of device time) are host-bound when issued call by call over several streams (~0.28 ms of
    Python per step): the compute launches of a step -- gather, Dice second stage, the [sum, count] pair -- are captured ONCE PER STREAM
    (each capture on its own stream: scratch memory is per stream, every captured launch owns its counter slot, csrc/api.hip) and the
    graphs are replayed round-robin, one replay per step.  `compute()` returns the [sum, count] device pair of a step; returns a step
    function like `pipelined` does.  The all-reduce stays outside the graphs and works on a copy of the pair."""
    from neurite_amd import distributed as nd
    streams = step_streams(nstreams, dev)
    graphs = []
    for st_ in streams:
        st_.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(st_):
            for _ in range(2):
                compute()                              # lazy allocations, workspace growth, counter-slot assignment: outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st_):
            pair = compute()
        graphs.append((g, pair))
    count = [0]

    def step(events=None):
        k = count[0] % nstreams
        count[0] += 1
        g, pair = graphs[k]
        with torch.cuda.stream(streams[k]):
            if events is not None:
                events[0].record()
            g.replay()
            if events is not None:
                events[1].record()
                events[2].record()
            return _OnStream(nd.all_reduce_mean_pair(pair.clone(), async_op=True), streams[k])
    step.graphs = graphs
    return step


def pipelined(step, nstreams, dev):
    """`step` issued round-robin on `nstreams` HIP streams (the reference's own batch loop, neurite/tf/layers.py:171
    `tf.map_fn`, is serial: this is what a data-parallel host does with INDEPENDENT steps).  Every stream first waits for the work that was
    current when the wrapper was made (the inputs), the steps themselves depend on nothing but their inputs, so consecutive steps may
    overlap: the first blocks of step k + 1's gather run on the CUs that step k's last round has left idle, and the second-stage
    launches of a step (partial rows -> sums -> dice, ~25 us of small kernels) run beside the next gather.  Scratch memory and work
    counters are per stream (neurite_amd/_lib.py: workspace; csrc/api.hip: counter slots), results are bit-identical to serial steps
    (tests/test_gpu_graph_capture.py, tools/two_stream_probe.py).  nstreams <= 1 returns `step` itself."""
    if nstreams <= 1:
        return step
    streams = step_streams(nstreams, dev)
    for st_ in streams:
        st_.wait_stream(torch.cuda.current_stream(dev))
    count = [0]

    def wrapped(events=None):
        st_ = streams[count[0] % nstreams]
        count[0] += 1
        with torch.cuda.stream(st_):
            return _OnStream(step(events), st_)
    wrapped.streams = streams
    return wrapped


def lookup_traffic(tfile, timed_kernel, B, ids=None):
    """HBM bytes per launch recorded for EXACTLY this kernel instantiation and batch (profiles/hbm_traffic.json, written by
    tools/update_hbm_traffic.py from rocprofv3 --pmc passes), or (None, why).  Counters of another instantiation are never quoted, and
    neither are counters of another BINARY: an entry carries the id of the library it was measured on and of the gather's sources, and
    `ids` = (id the loaded library reports, id of the tree's build, id of the tree's gather sources) must say that the loaded library is
    the tree's build and that the gather's sources are still the entry's (a kernel edit that keeps the name does not keep the bytes)."""
    if not timed_kernel:
        return None, 'not recorded for the unfused pipeline (see roofline_dropin; profiles/hbm_traffic.json lists interpn_zrun_c32)'
    try:
        ent = json.load(open(tfile)).get('kernels', {}).get(timed_kernel, {}).get('B%d' % B)
    except Exception:   # noqa
        ent = None
    if ent is None:
        return None, 'no counter pass recorded under profiles/hbm_traffic.json for the timed kernel %r at batch %d' % (timed_kernel, B)
    if ids is not None:
        lib_id, tree_id, gather_id = ids
        if lib_id != tree_id:
            return None, 'the loaded library (build id %s) is not the build of this tree (%s): recorded counters not quoted' % (lib_id, tree_id)
        if 'gather_sources_id' in ent:
            if ent['gather_sources_id'] != gather_id:
                return None, ('stale: profiles/hbm_traffic.json[%r][B%d] was measured on gather sources %s (library %s), this library is built from %s'
                              % (timed_kernel, B, ent['gather_sources_id'], ent.get('build_id'), gather_id))
        elif ent.get('build_id') != lib_id:
            return None, ('stale: profiles/hbm_traffic.json[%r][B%d] carries %s, the loaded library is %s'
                          % (timed_kernel, B, ('build id %s' % ent['build_id']) if ent.get('build_id') else 'no build id (recorded before round 6)', lib_id))
    return ent['bytes_per_launch'], 'static: profiles/hbm_traffic.json[%r][B%d] (%s; gather sources %s)' % (
        timed_kernel, B, ent.get('source', 'rocprofv3 --pmc pass'), ent.get('gather_sources_id', ent.get('build_id')))


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run this file under torch.distributed.run, one rank per GPU, and hand rank 0's
    JSON line on (the launcher's and RCCL's chatter stays on stderr).  The reference's only multi-device code is
    neurite/tf/utils/model.py:298-321 (Keras multi_gpu_model); here every rank is its own process with its own GPU."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, NRT_BENCH_CHILD='1', NRT_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log('bench.py: launching %d ranks: %s' % (n, ' '.join(cmd)))
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [ln for ln in p.stdout.decode(errors='replace').splitlines() if ln.strip().startswith('{')]
    if p.returncode != 0 or not lines:
        raise SystemExit('bench.py: the %d-rank launch failed (exit code %d, %d JSON lines)' % (n, p.returncode, len(lines)))
    print(lines[-1], flush=True)


CFG4_GLOBAL_BATCH = 32          # BASELINE config 4: "Batch=32 of 160^3 volumes ... sharded across 8 x MI355X"


def default_global_batch(global_batch, weak, world):
    """The global batch of a run that did not name one.  N > 1: BASELINE config 4 as SURVEY 8(d) words it -- 32 volumes in TOTAL, 32 / N per
    rank -- is the parsed line ("scaling": "strong"), so that the driver's value(N) / value(1) is the strong-scaling ratio; `--weak` (or a
    world that does not divide 32) keeps --batch-per-gpu volumes on every rank.  N = 1 keeps the 4-volume step (config 4's per-GPU share at
    N = 8, the workload `metric` is quoted on) and carries the 32-volume figure as `value_strong_b32`.  (The reference's only multi-device
    code, neurite/tf/utils/model.py:298-321, splits ONE batch over the devices: strong scaling.)"""
    if global_batch or weak or world == 1 or CFG4_GLOBAL_BATCH % world:
        return global_batch
    return CFG4_GLOBAL_BATCH


def batch_plan(global_batch, batch_per_gpu, world):
    """(volumes per rank, global batch, scaling) of a run: `--global-batch G` is BASELINE config 4 as written (G volumes in total,
    G / N per rank, "strong"); otherwise every rank holds `--batch-per-gpu` volumes ("weak").  A global batch the ranks do not divide
    is an error, never a silent truncation."""
    if global_batch:
        if global_batch % world:
            raise SystemExit('--global-batch %d is not a multiple of the %d ranks' % (global_batch, world))
        return global_batch // world, global_batch, 'strong'
    return batch_per_gpu, batch_per_gpu * world, 'weak'


def stub_main(args, rank, world):
    """--stub-step: the launch, the timed region and the one-JSON-line discipline on CPU ranks (gloo), with a stub in place of the
    kernels.  What it prints is shaped like the real line and says "data": "stub"."""
    import torch.distributed as dist
    from neurite_amd import distributed as nd
    group = None
    if world > 1 or os.environ.get('NRT_FORCE_DIST'):
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)
        group = dist
    B, global_batch, scaling = batch_plan(default_global_batch(args.global_batch, args.weak, world), args.batch_per_gpu, world)

    def step(events):
        time.sleep(0.002 * B)
        return nd.all_reduce_mean_dice(torch.full((B, 8), float(rank + 1)), async_op=True)
    r = timed(step, args.steps, args.warmup, group, None)
    if rank == 0:
        V = args.size ** 3
        print(json.dumps({'metric': 'Mvoxels/sec interpn+Dice on 160^3 x 32-label', 'value': round(world * B * V * args.steps / r['elapsed'] / 1e6, 2),
                          'unit': 'Mvoxels/s', 'n_gpus': world, 'rccl_ranks': r['ranks'], 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': round(r['elapsed'] / args.steps * 1e3, 4), 'higher_is_better': True,
                          'scaling': scaling, 'vs_baseline': None, 'dtype': 'f32', 'data': 'stub',
                          'config': {'workload': 'stub step (no kernels): plumbing self-test', 'volumes_per_gpu': B, 'global_batch': global_batch,
                                     'mean_dice': r['mean']}}), flush=True)
    if group is not None:
        dist.destroy_process_group()


def main():
    global PREWARM_MS
    args = parse()
    PREWARM_MS = 0.0 if args.stub_step else max(0.0, args.prewarm_ms)
    if ('WORLD_SIZE' not in os.environ and not os.environ.get('NRT_BENCH_CHILD')
            and (args.gpus > 1 or os.environ.get('NRT_FORCE_SPAWN'))):
        return self_launch(args.gpus)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.stub_step:
        return stub_main(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm device (MI355X); there is no CPU path.')
    # (NRT_DEVICE: every rank on one given device -- the two-ranks-on-one-GPU test of the multi-process control flow, which needs a backend
    # that accepts two ranks per device: NRT_DIST_BACKEND=gloo; a real run is one rank per GPU over nccl = RCCL)
    dev = torch.device('cuda', int(os.environ.get('NRT_DEVICE', local_rank)))
    torch.cuda.set_device(dev)
    dist = None
    real_stdout = None
    if world > 1 or (os.environ.get('NRT_FORCE_DIST') and 'RANK' in os.environ):
        # RCCL prints its version banner on the process's stdout at communicator creation: keep fd 1 for the one JSON line
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('NRT_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        log('warning: --gpus %d but WORLD_SIZE=%d; reporting n_gpus=%d' % (args.gpus, world, world))

    import neurite_amd as ne
    from neurite_amd import distributed as nd
    from neurite_amd import synth

    # strong scaling: BASELINE config 4, the global batch is fixed; weak: every GPU holds the same number of volumes
    named_batch = args.global_batch                      # what the command line named (0: the default of this world size applies)
    args.global_batch = default_global_batch(args.global_batch, args.weak, world)
    B = batch_plan(args.global_batch, args.batch_per_gpu, world)[0]
    S, L = args.size, args.labels
    V = S ** 3
    # rank r owns global batch entries [r*B, (r+1)*B): seeds follow the global entry index
    mov, fix, trf = synth.cfg2_batch(B, S, L, device=dev, seed0=100 + 3 * rank * B, rough=args.rough)
    torch.cuda.synchronize()

    if args.sweep:
        sweep(args, dev, mov, fix, trf)
        return
    if args.unet:
        del mov, fix, trf
        for ncpl in (1, 2):
            log(json.dumps(unet_bench(dev, nb_conv_per_level=ncpl)))
        log(json.dumps(lc3d_bench(dev)))
        return

    st = ne.layers.SpatialTransformer(interp_method='linear')
    st._variant, st._tune = args.variant, args.tune
    # a warped one-hot map overshoots 1.0 by an ulp, so the reference's default range assert would
    # abort the pipeline (see tests/test_gpu_dice_cce.py); min/max are still computed by the kernel
    dice = ne.metrics.Dice(check_input_limits=False)

    def make_steps(mov, fix, trf):
        def step_unfused(events=None):
            # the two reference-signature calls run EAGERLY: `warped` is written by one kernel and read back by the other
            keep = ne.deferred.enabled
            ne.deferred.enabled = False
            try:
                if events is not None:
                    events[0].record()
                warped = st([mov, trf])
                if events is not None:
                    events[1].record()
                d = dice.dice(fix, warped)                      # [B, L]
                if events is not None:
                    events[2].record()
            finally:
                ne.deferred.enabled = keep
            return nd.all_reduce_mean_dice(d, async_op=True)    # one all-reduce of 2 floats when world > 1

        def step_refsig(events=None):
            # the same two calls as a user of the reference writes them; SpatialTransformer defers the warp and Dice runs the
            # fused kernel (neurite_amd/deferred.py)
            if events is not None:
                events[0].record()
            warped = st([mov, trf])
            d = dice.dice(fix, warped)
            if events is not None:
                events[1].record()
                events[2].record()
            return nd.all_reduce_mean_dice(d, async_op=True)

        def step_fused(events=None):
            if events is not None:
                events[0].record()
            d = ne.fused.warp_dice(mov, trf, fix, _tune=args.tune)      # [B, L]; `warped` never leaves registers
            if events is not None:
                events[1].record()
                events[2].record()
            return nd.all_reduce_mean_dice(d, async_op=True)
        if not args.graph:
            return step_fused, step_unfused, step_refsig

        # --graph: the launches of a step (gather / Dice kernels, the two-level second stage, the [sum, count] pair) are
        # captured once and replayed as ONE hipGraph launch; the all-reduce stays outside the graph and works on a copy of
        # the pair, because step k + 1's replay rewrites the captured buffer while step k's collective may still read it
        def capture(compute):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    compute()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                pair = compute()

            def step(events=None):
                if events is not None:
                    events[0].record()
                g.replay()
                if events is not None:
                    events[1].record()
                    events[2].record()
                return nd.all_reduce_mean_pair(pair.clone(), async_op=True)
            step._graph = g
            return step
        def eager_pair():
            keep = ne.deferred.enabled
            ne.deferred.enabled = False
            try:
                return nd.mean_dice_pair(dice.dice(fix, st([mov, trf])))
            finally:
                ne.deferred.enabled = keep
        return (capture(lambda: nd.mean_dice_pair(ne.fused.warp_dice(mov, trf, fix, _tune=args.tune))), capture(eager_pair),
                capture(lambda: nd.mean_dice_pair(dice.dice(fix, st([mov, trf])))))

    step_fused, step_unfused, step_refsig = make_steps(mov, fix, trf)
    fused = not args.unfused
    # the timed pipeline: the reference's own two calls, SpatialTransformer -> Dice (the warp is deferred, Dice launches the fused
    # kernel); --direct times fused.warp_dice itself, --unfused the eager two-kernel form
    main_serial = (step_fused if args.direct else step_refsig) if fused else step_unfused
    # independent steps go round-robin over --streams HIP streams (a captured graph owns its buffers: replays stay on one stream)
    nstreams = 1 if (args.graph or args.unfused) else max(1, args.streams)
    main_step = pipelined(main_serial, nstreams, dev)
    r_main = timed(main_step, args.steps, args.warmup, dist, dev, sparse_events=nstreams > 1)
    elapsed, k0_ms, k1_ms, m = r_main['elapsed'], r_main['k0_ms'], r_main['k1_ms'], r_main['mean']
    o_steps = max(5, args.steps // 5)
    # the same steps strictly one after the other (what rocprofv3's per-kernel durations of an isolated launch correspond to)
    r_iso = timed(main_serial, args.steps, args.warmup, dist, dev) if nstreams > 1 else r_main
    # the other form of the same pipeline, shorter run, for the record
    r_other = timed(step_unfused if fused else step_fused, o_steps, 2, dist, dev)
    o_elapsed, o_k0, o_k1, o_m = r_other['elapsed'], r_other['k0_ms'], r_other['k1_ms'], r_other['mean']
    # the same kernel through the other entry (direct fused API when the reference-signature calls are the timed pipeline)
    r_ref = timed(step_fused if (fused and not args.direct) else step_refsig, o_steps, 2, dist, dev)
    # the stand-alone warp (the drop-in op that WRITES the warped volume) issued like the headline steps: independent launches
    # round-robin over the step streams, nothing else in between -- the rate at which interpn launches complete
    r_warp = r_warp_serial = None
    if dist is None and fused and not args.no_batch1:
        try:
            def step_warp(events=None):
                if events is not None:
                    events[0].record()
                w = ne.deferred.materialize(st([mov, trf]))
                if events is not None:
                    events[1].record()
                    events[2].record()
                return _Done(w)
            # ... and strictly one after the other with nothing else in between (the drop-in pipeline's `interpn_ms` is the same launch
            # alternating with the Dice kernels, --steps // 5 steps)
            r_warp_serial = timed(step_warp, args.steps, 2, None, dev)
            if nstreams > 1:
                r_warp = timed(pipelined(step_warp, nstreams, dev), args.steps, 2, None, dev, sparse_events=True)
            torch.cuda.empty_cache()
        except Exception as e:   # noqa
            log('stand-alone warp run failed: %s' % e)
            r_warp = r_warp_serial = None
    # BASELINE config 2 proper is batch = 1: the same two pipelines on the first volume only (N = 1 runs)
    r_b1 = None
    if dist is None and B > 1 and not args.no_batch1:
        f1, u1, _ = make_steps(mov[:1], fix[:1], trf[:1])
        b1_steps = 4 * o_steps                            # (a batch-1 step is a quarter of the headline step)
        b1_form = 'direct launches'
        if nstreams > 1:
            try:
                m1, t1, x1 = mov[:1], trf[:1], fix[:1]
                f1p = graph_pipelined(lambda: nd.mean_dice_pair(ne.fused.warp_dice(m1, t1, x1, _tune=args.tune)), nstreams, dev)
                b1_form = 'one hipGraph replay per step, one captured graph per stream'
            except Exception as e:   # noqa
                log('batch-1 graph capture failed (%s): direct launches' % e)
                f1p = pipelined(f1, nstreams, dev)
        else:
            f1p = f1
        r_b1 = (timed(f1p, b1_steps, 4, None, dev, sparse_events=nstreams > 1), timed(u1, o_steps, 2, None, dev),
                timed(f1, o_steps, 2, None, dev) if nstreams > 1 else None)
    # label maps stored as bfloat16 (exact for one-hot maps), float32 arithmetic: same Dice bit for bit, half the bytes per row
    r_bf16 = None
    if dist is None and not args.no_batch1:
        try:
            mov16, fix16 = mov.bfloat16(), fix.bfloat16()

            def step_bf16(events=None):
                if events is not None:
                    events[0].record()
                d = ne.fused.warp_dice(mov16, trf, fix16, _tune=args.tune)
                if events is not None:
                    events[1].record()
                    events[2].record()
                return nd.all_reduce_mean_dice(d, async_op=True)
            r_bf16 = timed(step_bf16, o_steps, 2, None, dev)
            d16 = ne.fused.warp_dice(mov16, trf, fix16)
            # same kernel structure (register kernel, tune bit 30): bit-identical; the float32 default (wave-cache kernel) sums in another order
            r_bf16['same_dice'] = bool(torch.equal(d16, ne.fused.warp_dice(mov, trf, fix, _tune=1 << 30)))
            r_bf16['max_abs_diff_vs_default_f32_kernel'] = float((d16 - ne.fused.warp_dice(mov, trf, fix)).abs().max())
            del mov16, fix16
        except Exception as e:   # noqa
            log('bf16-storage run failed: %s' % e)
            r_bf16 = None
    # the reference's default arguments (check_input_limits=True, metrics.py:439-444).  On one-hot maps the reference's own assert
    # fires (a tri-linear blend of ones exceeds 1.0 by an ulp: tests/test_gpu_dice_cce.py), so the default path is timed on the
    # same maps scaled by 1/2: same kernels, same bytes, plus the read-back of the four extrema the assert looks at
    r_def = r_def_eager = None
    if dist is None and not args.no_batch1:
        try:
            movh, fixh = mov * 0.5, fix * 0.5
            dice_def = ne.metrics.Dice()

            def step_default(events=None):
                if events is not None:
                    events[0].record()
                d = dice_def.dice(fixh, st([movh, trf]))
                if events is not None:
                    events[1].record()
                    events[2].record()
                return nd.all_reduce_mean_dice(d, async_op=True)
            r_def = timed(pipelined(step_default, nstreams, dev), o_steps, 2, None, dev, sparse_events=nstreams > 1)
            ne.checked.flush()               # every range assert of the run has been looked at (none may have failed)
            # the same with the assert raised at the call site (one host read-back of the extrema per step)
            keep_checked = ne.checked.enabled
            ne.checked.enabled = False
            try:
                r_def_eager = timed(step_default, o_steps, 2, None, dev)
            finally:
                ne.checked.enabled = keep_checked
            del movh, fixh
        except Exception as e:   # noqa
            log('default-argument run failed: %s' % e)
            r_def = None
    # The other scaling mode of the same run, next to the parsed line.  N = 1 (parsed line: 4 volumes): BASELINE config 4 as SURVEY 8(d)
    # words it -- B = 32 FIXED, all 32 volumes on this GPU (35 GB of its 288) -- so that value(N) / value_strong_b32 is the strong-scaling
    # figure.  N > 1 (parsed line: global batch 32, 32 / N per rank): the weak run, --batch-per-gpu volumes on every rank.
    r_strong, Bs, r_warp32 = None, 0, None
    if not args.global_batch and not args.no_strong and not args.unfused and not args.rough and CFG4_GLOBAL_BATCH % world == 0 and S == 160:
        try:
            Bs = CFG4_GLOBAL_BATCH // world
            if Bs == B:
                smov, sfix, strf = mov, fix, trf
            else:
                smov, sfix, strf = synth.cfg2_batch(Bs, S, L, device=dev, seed0=100 + 3 * rank * Bs)
            s_refsig = pipelined(make_steps(smov, sfix, strf)[2], nstreams, dev)
            r_strong = timed(s_refsig, o_steps, 2, dist, dev, sparse_events=nstreams > 1)
            if dist is None and fused and Bs != B:
                # the stand-alone warp on the same 32 volumes (serial launches): north_star's per-volume target is quoted at any batch
                del sfix

                def step_warp32(events=None):
                    if events is not None:
                        events[0].record()
                    w = ne.deferred.materialize(st([smov, strf]))
                    if events is not None:
                        events[1].record()
                        events[2].record()
                    return _Done(w)
                try:
                    r_warp32 = timed(step_warp32, o_steps, 1, None, dev)
                except Exception as e:   # noqa
                    log('stand-alone warp at batch %d failed: %s' % (Bs, e))
                sfix = None
            del smov, sfix, strf, s_refsig
            torch.cuda.empty_cache()
        except Exception as e:   # noqa
            log('cfg4_strong run failed on rank %d: %s' % (rank, e))
            r_strong = None
    r_weak, Bw = None, args.batch_per_gpu
    if args.global_batch and not named_batch and not args.no_strong and not args.unfused and not args.rough:
        try:
            if Bw == B:
                r_weak = r_main                          # (N = 8: config 4's shard IS 4 volumes per rank)
            else:
                wmov, wfix, wtrf = synth.cfg2_batch(Bw, S, L, device=dev, seed0=100 + 3 * rank * Bw)
                w_refsig = pipelined(make_steps(wmov, wfix, wtrf)[2], nstreams, dev)
                r_weak = timed(w_refsig, o_steps, 2, dist, dev, sparse_events=nstreams > 1)
                del wmov, wfix, wtrf, w_refsig
                torch.cuda.empty_cache()
        except Exception as e:   # noqa
            log('weak_per_gpu run failed on rank %d: %s' % (rank, e))
            r_weak = None
    unet_multi = None
    if dist is not None and not args.no_unet:
        # "3D UNet fwd ms at 1/2/4/8 GPU": every rank runs the config-3 forward on its own volume (data parallel inference);
        # the slowest rank's median is reported
        try:
            ms_un = unet_fwd_ms(dev)
        except Exception as e:   # noqa
            ms_un = float('inf')                 # every rank still joins the reduction below
            log('unet forward at world %d failed on rank %d: %s' % (world, rank, e))
        t_un = torch.tensor([ms_un], dtype=torch.float64, device=dev)
        dist.all_reduce(t_un, op=dist.ReduceOp.MAX)
        unet_multi = float(t_un[0]) if np.isfinite(float(t_un[0])) else None
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_vox = world * B * V * args.steps
    value = total_vox / elapsed / 1e6
    interp_bytes = INTERPN_BYTES_PER_VOXEL(L, 3) * V * B
    fused_bytes = (4 * L + 12 + 4 * L) * V * B
    if fused:
        # one kernel; it must move: moving row (4C) + loc (4D) + fixed row (4L) per voxel = 268 B at C=L=32
        kname = ('fused SpatialTransformer gather + Dice reduction (exact instantiation: `timed_kernel`), one launch per step; reached through '
                 + ('fused.warp_dice' if args.direct else 'layers.SpatialTransformer -> metrics.Dice (deferred warp)'))
        alg_bytes = fused_bytes
        # pipelined steps: a launch costs the rate at which launches complete (its own start-to-end time overlaps its neighbours')
        kms = r_main['span_ms'] if nstreams > 1 else k0_ms
    else:
        kname = 'interpn (SpatialTransformer gather), one launch per step'
        alg_bytes = interp_bytes
        kms = k0_ms
    achieved = alg_bytes / (kms * 1e-3) / 1e9
    # HBM-side bytes per launch are NOT measured in this process (PMC counters need a rocprofv3 pass of their own): the figure
    # is the FETCH_SIZE + WRITE_SIZE of the rocprofv3 --pmc pass recorded in profiles/hbm_traffic.json for this kernel and batch
    # ... and only for the kernel instantiation that was timed: the library names it (nrt_warp_dice_kernel_name), the file is keyed by
    # the names rocprofv3 printed; another instantiation's counters are not quoted (traffic stays null and says why)
    traffic, traffic_src, timed_kernel = None, None, None
    if fused:
        from neurite_amd import _lib
        timed_kernel = _lib.lib().nrt_warp_dice_kernel_name(_lib.ints([S] * 3), _lib.ints([S] * 3), L, B, _lib.LOC_SHIFT, 0, 0, 0,
                                                            int(args.tune)).decode()
    from neurite_amd import _lib, build as nbuild
    build_ids = (_lib.lib().nrt_build_id().decode(), nbuild.build_id(), nbuild.gather_sources_id())
    traffic, traffic_src = lookup_traffic(os.path.join(ROOT, 'profiles', 'hbm_traffic.json'), timed_kernel, B, build_ids)

    # drop-in (reference-signature) pipeline figures, whichever form was the timed one
    d_elapsed, d_steps, d_k0, d_k1, d_m = (o_elapsed, o_steps, o_k0, o_k1, o_m) if fused else (elapsed, args.steps, k0_ms, k1_ms, m)
    f_elapsed, f_steps, f_k0, f_m = (elapsed, args.steps, kms, m) if fused else (o_elapsed, o_steps, o_k0, o_m)
    dropin = {'what': 'reference-signature calls run eagerly (neurite_amd.deferred.enabled = False): layers.SpatialTransformer(linear) '
                      '-> metrics.Dice().dice, two kernels, `warped` written and re-read; %d steps' % d_steps,
              'value': round(world * B * V * d_steps / d_elapsed / 1e6, 2), 'unit': 'Mvoxels/s',
              'ms_per_step': round(d_elapsed / d_steps * 1e3, 4),
              'interpn_ms': round(d_k0, 4), 'interpn_GBs': round(interp_bytes / (d_k0 * 1e-3) / 1e9, 1),
              'interpn_frac_of_peak': round(interp_bytes / (d_k0 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
              'dice_ms': round(d_k1, 4), 'dice_GBs': round(DICE_BYTES_PER_VOXEL(L) * V * B / (d_k1 * 1e-3) / 1e9, 1),
              'dice_frac_of_peak': round(DICE_BYTES_PER_VOXEL(L) * V * B / (d_k1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
              'pipeline_524B_per_voxel_frac_of_peak': round(
                  (interp_bytes + DICE_BYTES_PER_VOXEL(L) * V * B) / (d_elapsed / d_steps) / 1e9 / HBM_PEAK_GBS, 4),
              'mean_dice': round(d_m, 6)}
    fusedb = {'what': ('the timed pipeline: reference-signature calls, warp deferred, fused kernel; %d steps' % f_steps)
                      if (fused and not args.direct) else
                      'fused warp+Dice kernel (neurite_amd.fused.warp_dice; `warped` never written), %d steps' % f_steps,
              'value': round(world * B * V * f_steps / f_elapsed / 1e6, 2), 'unit': 'Mvoxels/s',
              'ms_per_step': round(f_elapsed / f_steps * 1e3, 4), 'kernel_ms': round(f_k0, 4),
              'frac_of_peak_268B_per_voxel': round(fused_bytes / (f_k0 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
              'mean_dice': round(f_m, 6)}

    out = {
        'metric': 'Mvoxels/sec interpn+Dice on 160^3 x 32-label',
        'value': round(value, 2),
        'unit': 'Mvoxels/s',
        'n_gpus': world,
        'rccl_ranks': r_main['ranks'],
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 4),
        'ms_per_step_per_rank': [round(t / args.steps * 1e3, 4) for t in r_main['per_rank_s']],
        'higher_is_better': True,
        'scaling': 'strong' if args.global_batch else 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {
            'workload': 'BASELINE config 2/4: SpatialTransformer(linear)+Dice on %d^3 x %d-label one-hot fp32, '
                        '%d volumes per GPU per step (global batch %d), %s displacement field; %s'
                        % (S, L, B, B * world, 'worst-case U(-80,80)' if args.rough else 'smooth sigma=3 voxel',
                           ('reference-signature calls layers.SpatialTransformer -> metrics.Dice (warp deferred, fused kernel, warped '
                            'volume never written)' if not args.direct else 'fused kernel called directly (fused.warp_dice)') if fused
                           else 'reference-signature calls run eagerly: two kernels'),
            'volumes_per_gpu': B, 'global_batch': B * world, 'size': S, 'labels': L,
            'pipeline': ('reference_api' if not args.direct else 'fused_direct') if fused else 'unfused',
            'prewarm_ms_before_the_warmup_steps': PREWARM_MS,
            'step_launch': 'one hipGraph replay per step (--graph)' if args.graph else
                           ('direct kernel launches, independent steps round-robin on %d HIP streams' % nstreams if nstreams > 1 else 'direct kernel launches, one stream'),
            'parallelism': 'dp%d (batch-sharded, one RCCL all-reduce of 2 floats per step)' % world,
            'scaling_mode': ('strong: global batch %d%s split over the ranks' % (args.global_batch, '' if named_batch else ' (BASELINE config 4, the default of an N > 1 run)'))
                            if args.global_batch else 'weak: --batch-per-gpu %d on every rank' % B,
            'mean_dice': round(float(m), 6),
        },
        'roofline': {
            'kernel': kname,
            'bound': 'hbm',
            'achieved': round(achieved, 1),
            'peak': HBM_PEAK_GBS,
            'unit': 'GB/s',
            'frac': round(achieved / HBM_PEAK_GBS, 4),
            'traffic': traffic,
            'traffic_source': traffic_src,
            'timed_kernel': timed_kernel,
            'library_build_id': build_ids[0],
            'algorithmic_bytes_per_launch': alg_bytes,
            'avg_launch_ms': round(kms, 4),
            'avg_launch_ms_covers': (('steps pipelined over %d HIP streams: device time from the first event of the timed region to its last, '
                                      'divided by the launches -- the rate at which launches COMPLETE.  Consecutive gathers overlap (the head of '
                                      'launch k + 1 runs on the CUs launch k\'s last round has left), so a launch\'s own start-to-end time '
                                      '(`kernel_own_duration_ms`, what rocprofv3 lists per kernel) counts the shared stretches twice; '
                                      '`isolated_launch` is the same launch with the device to itself (steps strictly serial)' % nstreams)
                                     if (nstreams > 1 and fused) else
                                     'HIP events around the gather launch and, for the fused form, the two launches of its Dice second '
                                     'stage (reduce_rows + dice_soft_finalize, ~0.03 ms together): rocprofv3 lists the gather alone'),
            'steps_in_flight': nstreams,
            'kernel_own_duration_ms': round(k0_ms, 4),
            'isolated_launch': {
                'what': 'the same launch with the device to itself: steps issued strictly one after the other on one stream, HIP events around '
                        'the gather and its Dice second stage; %d steps.  This is the figure rocprofv3 --stats of a `--streams 1` run reproduces' % args.steps,
                'avg_launch_ms': round(r_iso['k0_ms'], 4),
                'achieved': round(alg_bytes / (r_iso['k0_ms'] * 1e-3) / 1e9, 1),
                'frac': round(alg_bytes / (r_iso['k0_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                'ms_per_step': round(r_iso['elapsed'] / args.steps * 1e3, 4)},
        },
        # the reference-signature path (what `SpatialTransformer` + `Dice` callers reach) next to the fused headline
        'roofline_dropin': {'kernel': 'interpn (SpatialTransformer gather, drop-in API), one launch per step', 'bound': 'hbm',
                            'achieved': dropin['interpn_GBs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                            'frac': dropin['interpn_frac_of_peak'], 'avg_launch_ms': dropin['interpn_ms'],
                            'algorithmic_bytes_per_launch': interp_bytes},
        # what a caller of the reference's own two calls gets by default: SpatialTransformer defers the warp, Dice runs the
        # fused kernel on (moving, trf, fixed) -- same kernel and numbers as `fused_pipeline`, reached through the reference API
        ('fused_direct_pipeline' if (fused and not args.direct) else 'reference_api_pipeline'): {
            'what': ('neurite_amd.fused.warp_dice(moving, trf, fixed) called directly: the same kernel without the deferred-warp '
                     'indirection; %d steps' % o_steps) if (fused and not args.direct) else
                    ('layers.SpatialTransformer(linear)([moving, trf]) -> metrics.Dice().dice(fixed, warped) as written against the '
                     'reference; the warp is deferred and Dice launches the fused kernel (neurite_amd/deferred.py); %d steps' % o_steps),
            'value': round(world * B * V * o_steps / r_ref['elapsed'] / 1e6, 2), 'unit': 'Mvoxels/s',
            'ms_per_step': round(r_ref['elapsed'] / o_steps * 1e3, 4), 'kernel_ms': round(r_ref['k0_ms'], 4),
            'frac_of_peak_268B_per_voxel': round(fused_bytes / (r_ref['k0_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            'mean_dice': round(r_ref['mean'], 6)},
        'dropin_pipeline': dropin,
        'fused_pipeline': fusedb,
        'other_pipeline': dropin if fused else fusedb,
    }
    if r_strong is not None:
        out['cfg4_strong'] = {
            'what': 'BASELINE config 4 with the global batch FIXED at 32 (SURVEY 8d): %d volumes on each of %d GPUs, the reference-signature '
                    'pipeline, one all-reduce per step; %d steps.  value(N) / value(1) is the strong-scaling ratio' % (Bs, world, o_steps),
            'scaling': 'strong', 'global_batch': 32, 'volumes_per_gpu': Bs, 'n_gpus': world, 'rccl_ranks': r_strong['ranks'],
            'value': round(32 * V * o_steps / r_strong['elapsed'] / 1e6, 2), 'unit': 'Mvoxels/s',
            'ms_per_step': round(r_strong['elapsed'] / o_steps * 1e3, 4), 'kernel_ms': round(r_strong['k0_ms'], 4),
            'mean_dice': round(r_strong['mean'], 6)}
        # N = 1: the denominator of the driver's strong-scaling ratio sits in the head of the line
        out['value_strong_b32'] = out['cfg4_strong']['value']
    if r_weak is not None:
        w_steps = args.steps if r_weak is r_main else o_steps
        out['weak_per_gpu'] = {
            'what': 'the weak-scaling run next to the parsed (strong, global batch %d) line: %d volumes on each of %d GPUs, the reference-signature '
                    'pipeline, one all-reduce per step; %d steps%s' % (args.global_batch, Bw, world, w_steps,
                                                                        ' (this world size: the same run as the parsed line)' if r_weak is r_main else ''),
            'scaling': 'weak', 'global_batch': Bw * world, 'volumes_per_gpu': Bw, 'n_gpus': world, 'rccl_ranks': r_weak['ranks'],
            'value': round(world * Bw * V * w_steps / r_weak['elapsed'] / 1e6, 2), 'unit': 'Mvoxels/s',
            'ms_per_step': round(r_weak['elapsed'] / w_steps * 1e3, 4), 'kernel_ms': round(r_weak['k0_ms'], 4),
            'mean_dice': round(r_weak['mean'], 6)}
    if r_b1 is not None:
        rf, ru, rf_iso = r_b1
        b1_bytes = (4 * L + 12 + 4 * L) * V
        b1_ms = rf['span_ms'] if nstreams > 1 else rf['k0_ms']
        out['config2_batch1'] = {
            'what': 'BASELINE config 2 as written: batch = 1, one %d^3 x %d-label volume per step, %d steps%s' % (
                S, L, b1_steps, ' round-robin on %d streams (%s)' % (nstreams, b1_form) if nstreams > 1 else ''),
            'fused': {'ms': round(rf['elapsed'] / b1_steps * 1e3, 4), 'Mvoxels_per_s': round(V * b1_steps / rf['elapsed'] / 1e6, 1),
                      'kernel_ms': round(b1_ms, 4), 'kernel_own_duration_ms': round(rf['k0_ms'], 4),
                      'frac': round(b1_bytes / (b1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            'dropin': {'ms': round(ru['elapsed'] / o_steps * 1e3, 4), 'Mvoxels_per_s': round(V * o_steps / ru['elapsed'] / 1e6, 1),
                       'interpn_ms': round(ru['k0_ms'], 4), 'dice_ms': round(ru['k1_ms'], 4),
                       'interpn_frac': round(INTERPN_BYTES_PER_VOXEL(L, 3) * V / (ru['k0_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
        if rf_iso is not None:
            out['config2_batch1']['fused_isolated'] = {
                'ms': round(rf_iso['elapsed'] / o_steps * 1e3, 4), 'kernel_ms': round(rf_iso['k0_ms'], 4),
                'frac': round(b1_bytes / (rf_iso['k0_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        # BASELINE config 2 names batch = 1: its figures also sit in the two objects a reader of the line's head keeps (`roofline`, `config`)
        out['roofline']['batch1'] = {'what': 'the same fused kernel on ONE volume per launch (BASELINE config 2 as written)'
                                             + ('; launches pipelined as in the headline, `isolated_frac` = strictly serial launches' if rf_iso is not None else ''),
                                     'avg_launch_ms': out['config2_batch1']['fused']['kernel_ms'],
                                     'frac': out['config2_batch1']['fused']['frac'],
                                     'standalone_interpn_frac': out['config2_batch1']['dropin']['interpn_frac']}
        if rf_iso is not None:
            out['roofline']['batch1']['isolated_frac'] = out['config2_batch1']['fused_isolated']['frac']
            out['roofline']['batch1']['isolated_avg_launch_ms'] = out['config2_batch1']['fused_isolated']['kernel_ms']
        out['config']['batch1_ms_per_step'] = out['config2_batch1']['fused']['ms']
        out['config']['batch1_Mvoxels_per_s'] = out['config2_batch1']['fused']['Mvoxels_per_s']
    if r_def is not None:
        out['default_args_pipeline'] = {
            'what': 'SpatialTransformer -> metrics.Dice() with the reference defaults (check_input_limits=True, metrics.py:439-444) on the bench '
                    'maps scaled by 1/2 (on one-hot maps the reference assert itself fires, see tests): the range asserts travel with the '
                    'result (neurite_amd/checked.py: extrema from the fused kernel, raised when the values reach the host or at a later call), '
                    'no host read-back per step; steps issued as in the headline; %d steps' % o_steps,
            'value': round(B * V * o_steps / r_def['elapsed'] / 1e6, 2), 'unit': 'Mvoxels/s',
            'ms_per_step': round(r_def['elapsed'] / o_steps * 1e3, 4), 'kernel_ms': round(r_def['span_ms'] if nstreams > 1 else r_def['k0_ms'], 4),
            'vs_headline': round(B * V * o_steps / r_def['elapsed'] / 1e6 / value, 4)}
        if r_def_eager is not None:
            out['default_args_pipeline']['assert_raised_at_the_call_site'] = {
                'what': 'neurite_amd.checked.enabled = False: one host read-back of the four extrema per step, steps serial',
                'value': round(B * V * o_steps / r_def_eager['elapsed'] / 1e6, 2), 'ms_per_step': round(r_def_eager['elapsed'] / o_steps * 1e3, 4)}
    if r_bf16 is not None:
        b16 = (2 * L + 12 + 2 * L) * V * B
        out['bf16_storage'] = {
            'what': 'the fused kernel on the same one-hot maps STORED as bfloat16 (float32 arithmetic on the widened rows; Dice '
                    'bit-identical to the float32 run: same_dice); not the headline -- BASELINE names fp32 volumes; %d steps' % o_steps,
            'value': round(B * V * o_steps / r_bf16['elapsed'] / 1e6, 2), 'unit': 'Mvoxels/s',
            'ms_per_step': round(r_bf16['elapsed'] / o_steps * 1e3, 4), 'kernel_ms': round(r_bf16['k0_ms'], 4),
            'algorithmic_bytes_per_voxel': 2 * L + 12 + 2 * L,
            'frac_of_peak': round(b16 / (r_bf16['k0_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'same_dice': r_bf16['same_dice'],
            'max_abs_diff_vs_default_f32_kernel': r_bf16['max_abs_diff_vs_default_f32_kernel']}
    # the stand-alone op (the drop-in `interpn` / SpatialTransformer call that WRITES the warped volume; 268 B per voxel as well)
    sa = {'frac': dropin['interpn_frac_of_peak'], 'avg_launch_ms': dropin['interpn_ms'], 'achieved': dropin['interpn_GBs'], 'volumes_per_launch': B,
          'what': 'isolated launches (the eager two-kernel pipeline, steps serial)'}
    if r_warp_serial is not None:
        w_ms = r_warp_serial['k0_ms']
        sa = {'frac': round(interp_bytes / (w_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'avg_launch_ms': round(w_ms, 4),
              'achieved': round(interp_bytes / (w_ms * 1e-3) / 1e9, 1), 'volumes_per_launch': B, 'ms_per_volume': round(w_ms / B, 4),
              'what': 'the warp alone, launches strictly one after the other on one stream, HIP events around every launch; %d launches' % args.steps,
              'ms_per_launch_wall': round(r_warp_serial['elapsed'] / args.steps * 1e3, 4),
              'in_dropin_pipeline': {'what': 'the same launch alternating with the Dice kernels of the eager two-kernel pipeline (`dropin_pipeline`)',
                                     'avg_launch_ms': dropin['interpn_ms'], 'frac': dropin['interpn_frac_of_peak']}}
    out['roofline']['standalone_interpn'] = sa
    if r_warp is not None:
        out['roofline']['standalone_interpn']['pipelined'] = {
            'what': 'the warp alone, independent launches round-robin on %d streams (as the headline steps): device time of the region / launches; %d launches' % (nstreams, args.steps),
            'avg_launch_ms': round(r_warp['span_ms'], 4),
            'achieved': round(interp_bytes / (r_warp['span_ms'] * 1e-3) / 1e9, 1),
            'frac': round(interp_bytes / (r_warp['span_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    if r_warp32 is not None:
        w32 = r_warp32['k0_ms']
        out['roofline']['standalone_interpn']['batch%d' % Bs] = {
            'what': 'the warp alone on the %d volumes of the strong-scaling run, serial launches; %d launches.  north_star: 60 %% of the roof = 0.229 ms per volume' % (Bs, o_steps),
            'avg_launch_ms': round(w32, 4), 'ms_per_volume': round(w32 / Bs, 4),
            'frac': round(INTERPN_BYTES_PER_VOXEL(L, 3) * V * Bs / (w32 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    if fused:
        out['roofline']['unfused_api_accounting_524B_per_voxel_GBs'] = round(
            (INTERPN_BYTES_PER_VOXEL(L, 3) + DICE_BYTES_PER_VOXEL(L)) * V * B / (kms * 1e-3) / 1e9, 1)
    else:
        out['roofline']['dice_kernel'] = {'avg_ms': round(k1_ms, 4),
                                          'achieved': round(DICE_BYTES_PER_VOXEL(L) * V * B / (k1_ms * 1e-3) / 1e9, 1),
                                          'unit': 'GB/s'}
    if world == 1 and not args.no_cpu_baseline:
        try:
            base, d_cpu = cpu_baseline(mov, fix, trf)
            out['cpu_baseline'] = base
            try:
                out['cpu_baseline']['cfg1_32cubed'] = cpu_cfg1()
            except Exception as e:   # noqa
                out['cpu_baseline']['cfg1_32cubed'] = {'error': str(e)}
            d_gpu = (ne.fused.warp_dice(mov[:1], trf[:1], fix[:1]) if fused
                     else dice.dice(fix[:1], ne.deferred.materialize(st([mov[:1], trf[:1]])))).cpu().numpy()
            out['config']['max_abs_dice_diff_vs_oracle'] = float(np.abs(d_gpu - d_cpu).max())
        except Exception as e:   # noqa
            out['cpu_baseline'] = {'value': None, 'unit': 'Mvoxels/s', 'cores': 0, 'kind': 'port',
                                   'sample': 'failed: %s' % e}
    if world == 1 and not args.no_unet:
        out['training'] = {}
        try:
            out['training']['registration_step'] = registration_bench(mov, fix, trf)
        except Exception as e:   # noqa
            out['training']['registration_step'] = {'error': str(e)}
        try:
            del mov, fix, trf
            torch.cuda.empty_cache()
            out['unet_fwd'] = unet_bench(dev)
        except Exception as e:   # noqa
            out['unet_fwd'] = {'error': str(e)}
        try:
            out['lc3d_wcce'] = lc3d_bench(dev)
        except Exception as e:   # noqa
            out['lc3d_wcce'] = {'error': str(e)}
        try:
            torch.cuda.empty_cache()
            out['training']['unet_train_step'] = unet_train_bench(dev, size=S, labels=L)
        except Exception as e:   # noqa
            out['training']['unet_train_step'] = {'error': str(e)}
    if unet_multi is not None:
        out['unet_fwd'] = {'config': 'BASELINE config 3, one 160^3 volume per GPU (data-parallel inference)', 'n_gpus': world,
                           'fwd_ms': round(unet_multi, 3), 'volumes_per_s': round(world / (unet_multi * 1e-3), 1)}
    if real_stdout is not None:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        os.close(real_stdout)
    print(json.dumps(out), flush=True)
    if dist is not None:
        sys.stdout.flush()
        os.dup2(2, 1)                       # anything RCCL says at teardown goes to stderr as well
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
