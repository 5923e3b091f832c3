"""Aggregate rocprofv3 --pmc counter CSVs.
    summarize_pmc.py OUT           mean counter value per dispatch, per kernel (bench runs)
    summarize_pmc.py OUT sweep     per (kernel, grid) in dispatch order (kernel-variant sweep)
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced
stream (MI355X_MICROARCH.md, HBM section) -- calibrated here on the Dice kernel whose byte count is known.
"""
import csv
import glob
import json
import os
import sys
from collections import OrderedDict, defaultdict

out_dir = sys.argv[1]
mode = sys.argv[2] if len(sys.argv) > 2 else 'bench'


def short(k):
    k = k.replace('(anonymous namespace)::', '').replace('void ', '')
    return k.split('(')[0][:70]


if mode == 'sweep':
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        files = glob.glob(os.path.join(out_dir, 'pmcsweep_' + counter, '**', '*counter_collection.csv'), recursive=True)
        groups = OrderedDict()
        for f in files:
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if row.get('Counter_Name') != counter:
                        continue
                    name = short(row.get('Kernel_Name', '?'))
                    if 'interpn' not in name and 'dice' not in name:
                        continue
                    key = (name, row.get('Grid_Size', ''), row.get('Workgroup_Size', ''))
                    groups.setdefault(key, []).append(float(row.get('Counter_Value', 0)))
        res = []
        for (name, grid, wg), v in groups.items():
            res.append({'kernel': name, 'grid': grid, 'wg': wg, 'n': len(v), counter + '_MB': round(sum(v) / len(v) / 1024, 1)})
            print(json.dumps(res[-1]))
        with open(os.path.join(out_dir, 'pmcsweep_%s.json' % counter), 'w') as f:
            json.dump(res, f, indent=1)
    sys.exit(0)

summary = {}
for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
    files = glob.glob(os.path.join(out_dir, 'pmc_' + counter, '**', '*counter_collection.csv'), recursive=True)
    acc = defaultdict(lambda: [0.0, 0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') != counter:
                    continue
                k = short(row.get('Kernel_Name', '?'))
                if 'interpn' not in k and 'dice' not in k and 'wcce' not in k and 'reduce_rows' not in k:
                    continue
                acc[k][0] += float(row.get('Counter_Value', 0))
                acc[k][1] += 1
    for k, (s, n) in acc.items():
        summary.setdefault(k, {})[counter + '_KiB_per_dispatch'] = round(s / max(n, 1), 1)
        summary[k]['dispatches_' + counter] = n
print(json.dumps(summary, indent=1))
with open(os.path.join(out_dir, 'pmc_summary.json'), 'w') as f:
    json.dump(summary, f, indent=1)
