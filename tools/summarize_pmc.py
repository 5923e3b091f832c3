"""Aggregate rocprofv3 --pmc counter CSVs per kernel name: mean counter value per dispatch."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir = sys.argv[1]
summary = {}
for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
    files = glob.glob(os.path.join(out_dir, 'pmc_' + counter, '**', '*counter_collection.csv'), recursive=True)
    acc = defaultdict(lambda: [0.0, 0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') != counter:
                    continue
                k = row.get('Kernel_Name', '?')
                acc[k][0] += float(row.get('Counter_Value', 0))
                acc[k][1] += 1
    for k, (s, n) in acc.items():
        short = k.split('(')[0][-60:]
        summary.setdefault(short, {})[counter + '_KB_per_dispatch'] = s / max(n, 1)
        summary[short]['dispatches_' + counter] = n
print(json.dumps(summary, indent=1))
with open(os.path.join(out_dir, 'pmc_summary.json'), 'w') as f:
    json.dump(summary, f, indent=1)
