#!/usr/bin/env python3
"""profiles/r05_unet/mfma_counters.json from the counter pass of the unet forward (gpurun_out/pmc_unet_r05.json, written by
`PMC_ONLY="mfma mfma2" bash tools/pmc_cmd.sh unet_r05 conv python tools/unet_small.py 5`) and its kernel stats
(gpurun_out/unet_kernel_stats.csv, rocprofv3 --kernel-trace --stats of tools/unet_small.py):
    executed flops = SQ_INSTS_VALU_MFMA_MOPS_F32 x 512;  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
    python tools/unet_mfma_summary.py [label]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
label = sys.argv[1] if len(sys.argv) > 1 else 'r05_unet'
pmc = json.load(open(os.path.join(ROOT, 'gpurun_out', 'pmc_unet_r05.json')))
stats = {}
for r in csv.DictReader(open(os.path.join(ROOT, 'gpurun_out', 'unet_kernel_stats.csv'))):
    k = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    stats[k] = float(r['AverageNs']) * 1e-6
# BASELINE config 3, one convolution per level: which kernel runs which layer
LAYERS = {'unet_conv_downarm_0_0': 'conv3d_c1_mfma<1, true>', 'unet_conv_downarm_1_0': 'conv3d_p27_mfma<2>',
          'unet_conv_downarm_2_0': 'conv3d_mfma<2, true, false>', 'unet_conv_uparm_3_0': 'conv3d_up2_mfma<2, 0>',
          'unet_conv_uparm_4_0': 'conv3d_up2_mfma<1, 2>'}
out = {'note': 'matrix-core counters of the config-3 unet forward, per kernel: rocprofv3 --pmc passes (tools/pmc_cmd.sh groups mfma, mfma2) and '
               '--kernel-trace --stats of tools/unet_small.py, one session (%s).  conv3d_up2_mfma<1, 2> is the last decoder convolution WITH the '
               'folded head (the bench times that layer without it)' % label,
       'peak_tflops': 157.3, 'layers': {}}
for layer, kern in LAYERS.items():
    c = pmc.get(kern)
    if c is None:
        continue
    gflop = c['SQ_INSTS_VALU_MFMA_MOPS_F32'] * 512 / 1e9
    busy = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * c['GRBM_GUI_ACTIVE'] / 8.0)
    ms = stats.get(kern)
    out['layers'][layer] = {'kernel': kern, 'executed_gflop': round(gflop, 2), 'mfma_busy': round(busy, 4), 'ms_under_rocprofv3': None if ms is None else round(ms, 4),
                            'flops_over_time_peak': None if ms is None else round(gflop / ms / 157.3, 4)}
os.makedirs(os.path.join(ROOT, 'profiles', label), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'profiles', label, 'mfma_counters.json'), 'w'), indent=1)
print(json.dumps(out, indent=1))
