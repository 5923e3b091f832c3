#!/usr/bin/env python3
"""unet training step (BASELINE config 3, forward + backward + SGD) with the loss pair evaluated jointly (losses.multiple_losses_decorator,
csrc/segloss.hip) and separately; one JSON line.   python tools/train_step_probe.py [reps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench            # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
print(json.dumps(bench.unet_train_bench(torch.device('cuda:0'), reps=reps)))
from neurite_amd import metrics      # noqa: E402
print(json.dumps({'joint_applications': metrics.JointSegLoss.applications, 'through_softmax': metrics.JointSegLoss.through_softmax}))
