"""profiles/hbm_traffic.json from a `bash tools/gpu_session.sh pmc` run (gpurun_out/pmc_summary.json).
    python tools/update_hbm_traffic.py <label of the run, e.g. r02_session3> [batch]
FETCH_SIZE (KiB per dispatch) is corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: the
factor is calibrated on dice_soft_vec, a pure streaming read whose byte count is known (expected 2.0); WRITE_SIZE is used as is.
The bench kernels must all have been launched at the headline batch (bench.py --no-batch1)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
label = sys.argv[1] if len(sys.argv) > 1 else 'unlabelled'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S, L = 160, 32
V = S ** 3
summ = json.load(open(os.path.join(ROOT, 'gpurun_out', 'pmc_summary.json')))
def find(sub):
    for k, v in summ.items():
        if sub in k and 'bwd' not in k:
            return k, v
    raise SystemExit('no kernel matching %r in pmc_summary.json' % sub)
kd, d = find('dice_soft_vec')
dice_bytes = 2 * 4 * L * V * B
factor = dice_bytes / (d['FETCH_SIZE_KiB_per_dispatch'] * 1024.0)
out = {'note': 'HBM-side traffic per launch (bytes) of the bench kernels at --batch-per-gpu %d: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE '
               '(separate passes, kernel-trace only), FETCH_SIZE (KiB) x the gfx950 correction calibrated on %s whose byte count is known '
               '(MI355X_MICROARCH.md, HBM section: expected 2), WRITE_SIZE (KiB) as is.' % (B, kd),
       'source': 'rocprofv3 --pmc passes of %s: profiles/%s/pmc_summary.json' % (label, label),
       'fetch_correction_factor': round(factor, 4), 'algorithmic_bytes_per_launch_B%d' % B: {
           'fused': (4 * L + 12 + 4 * L) * V * B, 'interpn': (4 * L + 12 + 4 * L) * V * B, 'dice': dice_bytes}}
for name, sub in (('fused', 'warp_dice_tile'), ('interpn', 'interpn_zrun'), ('dice', 'dice_soft_vec')):
    try:
        k, v = find(sub)
    except SystemExit:
        continue
    out['%s_bytes_per_launch_B%d' % (name, B)] = int(v['FETCH_SIZE_KiB_per_dispatch'] * 1024 * factor + v.get('WRITE_SIZE_KiB_per_dispatch', 0) * 1024)
    out['%s_kernel' % name] = k
old = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
if os.path.exists(old):
    out['history'] = json.load(open(old)).get('history', {})
json.dump(out, open(old, 'w'), indent=1)
print(json.dumps(out, indent=1))
