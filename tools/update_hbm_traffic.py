"""profiles/hbm_traffic.json from a `bash tools/gpu_session.sh pmc` run (gpurun_out/pmc_summary.json).
    python tools/update_hbm_traffic.py <label of the run, e.g. r04_final> [batch]
Entries are keyed by the EXACT kernel name rocprofv3 printed (template arguments included) and by the batch: bench.py asks the
library which instantiation it launches (nrt_warp_dice_kernel_name) and quotes traffic only for that name (the counters
of round 3 were taken on <..., 4, float> while <..., 3, float> was the timed kernel).
FETCH_SIZE (KiB per dispatch) is corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: the
factor is calibrated on dice_soft_vec, a pure streaming read whose byte count is known (expected 2.0); WRITE_SIZE is used as is.
The bench kernels must all have been launched at the headline batch (bench.py --no-batch1).
Every entry is stamped with the ids the session wrote next to the counters (gpurun_out/pmc_build_ids.json: the library that ran, the
sources of the gather kernels); bench.py quotes an entry only while both still describe the library it has loaded."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
label = sys.argv[1] if len(sys.argv) > 1 else 'unlabelled'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S, L = 160, 32
V = S ** 3
summ = json.load(open(os.path.join(ROOT, 'gpurun_out', 'pmc_summary.json')))
cal = [(k, v) for k, v in summ.items() if 'dice_soft_vec' in k and 'bwd' not in k and 'FETCH_SIZE_KiB_per_dispatch' in v]
if not cal:
    raise SystemExit('no dice_soft_vec dispatch in pmc_summary.json: the FETCH_SIZE correction cannot be calibrated')
kd, d = cal[0]
dice_bytes = 2 * 4 * L * V * B
factor = dice_bytes / (d['FETCH_SIZE_KiB_per_dispatch'] * 1024.0)
try:
    ids = json.load(open(os.path.join(ROOT, 'gpurun_out', 'pmc_build_ids.json')))
    if ids['library_build_id'] != ids['tree_build_id']:
        raise SystemExit('the session ran a library (%s) that is not the build of its tree (%s): counters not recorded' % (ids['library_build_id'], ids['tree_build_id']))
except (OSError, ValueError, KeyError):
    raise SystemExit('gpurun_out/pmc_build_ids.json missing: run the `pmc` stage of tools/gpu_session.sh (it records which binary the counters belong to)')
path = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
old = json.load(open(path)) if os.path.exists(path) else {}
out = {'note': 'HBM-side traffic per launch (bytes) of the bench kernels, keyed by exact kernel name and batch: rocprofv3 --pmc FETCH_SIZE and '
               '--pmc WRITE_SIZE (separate passes, kernel-trace only), FETCH_SIZE (KiB) x the gfx950 correction calibrated on dice_soft_vec '
               'whose byte count is known (MI355X_MICROARCH.md, HBM section: expected 2), WRITE_SIZE (KiB) as is.',
       'kernels': old.get('kernels', {}), 'history': old.get('history', {})}
for k, v in summ.items():
    if 'FETCH_SIZE_KiB_per_dispatch' not in v or not any(s in k for s in ('warp_dice', 'interpn', 'dice_soft_vec')):
        continue
    fetch = v['FETCH_SIZE_KiB_per_dispatch'] * 1024 * factor
    write = v.get('WRITE_SIZE_KiB_per_dispatch', 0) * 1024
    out['kernels'].setdefault(k, {})['B%d' % B] = {
        'bytes_per_launch': int(fetch + write), 'fetch_bytes': int(fetch), 'write_bytes': int(write), 'fetch_correction_factor': round(factor, 4),
        'calibrated_on': kd, 'build_id': ids['library_build_id'], 'gather_sources_id': ids['gather_sources_id'], 'source': 'rocprofv3 --pmc passes of %s: profiles/%s/pmc_summary.json' % (label, label)}
json.dump(out, open(path, 'w'), indent=1)
print(json.dumps(out, indent=1))
