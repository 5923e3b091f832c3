#!/usr/bin/env python3
"""What a launch costs when consecutive launches overlap: from a rocprofv3 --kernel-trace CSV (one row per dispatch with start / end time
stamps), for every kernel whose name contains <substring>:
    dispatches, mean own duration (end - start: what `--stats` averages), and the dispatches grouped into RUNS of launches that follow
    each other without the device going idle in between (gap < 50 us) -- per run: launches, span (last end - first start), service time =
    span / launches, and how much of the span two or more of them were running at once.
With steps pipelined over several streams (bench.py --streams N) the service time is what `roofline.avg_launch_ms` reports and the own
duration is `kernel_own_duration_ms`; with --streams 1 the two coincide.
    python tools/trace_service_time.py <dir or csv> <substring> [grid_size]"""
import csv, glob, json, os, sys

path, match = sys.argv[1], sys.argv[2]
grid = sys.argv[3] if len(sys.argv) > 3 else None
files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r.get('Kernel_Name', '')
            if match not in name:
                continue
            if grid and r.get('Grid_Size', r.get('Grid_Size_X', '')) not in (grid,):
                continue
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]))
rows.sort()
out = {'kernel_match': match, 'dispatches': len(rows)}
if rows:
    out['kernel'] = rows[0][2]
    out['mean_own_duration_ms'] = round(sum(e - s for s, e, _ in rows) / len(rows) / 1e6, 4)
    runs, cur = [], [rows[0]]
    for r in rows[1:]:
        if r[0] - max(e for _, e, _ in cur) < 50000:
            cur.append(r)
        else:
            runs.append(cur); cur = [r]
    runs.append(cur)
    summary = []
    for run in runs:
        if len(run) < 8:
            continue
        s0, e1 = run[0][0], max(e for _, e, _ in run)
        # time with >= 2 of the run's launches in flight (sweep over the end points)
        ev = sorted([(s, 1) for s, _, _ in run] + [(e, -1) for _, e, _ in run])
        depth, last, both = 0, s0, 0
        for tstamp, d in ev:
            if depth >= 2:
                both += tstamp - last
            depth += d; last = tstamp
        summary.append({'launches': len(run), 'span_ms': round((e1 - s0) / 1e6, 4), 'service_ms_per_launch': round((e1 - s0) / len(run) / 1e6, 4),
                        'mean_own_duration_ms': round(sum(e - s for s, e, _ in run) / len(run) / 1e6, 4),
                        'fraction_of_span_with_two_or_more_in_flight': round(both / (e1 - s0), 3)})
    summary.sort(key=lambda r: -r['launches'])
    out['number_of_runs'] = len(summary)
    out['longest_runs_of_back_to_back_launches'] = summary[:4]
print(json.dumps(out, indent=1))
