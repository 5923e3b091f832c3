import json, sys, torch
sys.path.insert(0, '.')
import bench
print(json.dumps(bench.lc3d_bench(torch.device('cuda:0'))))
