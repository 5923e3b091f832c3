import json, os, sys, time
import torch
sys.path.insert(0, os.getcwd())
import bench
import neurite_amd as ne
from neurite_amd import synth, distributed as nd
dev = torch.device('cuda:0')
mov, fix, trf = synth.cfg2_batch(4, 160, 32, device=dev)
m1, t1, x1 = mov[:1], trf[:1], fix[:1]
comp = lambda: nd.mean_dice_pair(ne.fused.warp_dice(m1, t1, x1))
out = {}
for rep in range(2):
    stp = bench.graph_pipelined(comp, 2, dev)
    r = bench.timed(stp, 40, 4, None, dev, sparse_events=True)
    out['A_bench_graph_step_%d' % rep] = round(r['elapsed'] / 40 * 1e3, 4)
    # B: no events at all, same step
    def loop(n, stp=stp):
        p = None
        for k in range(n):
            nx = stp(None)
            if p is not None: p.result()
            p = nx
        p.result()
    loop(8); torch.cuda.synchronize(); t0 = time.perf_counter(); loop(96); torch.cuda.synchronize()
    out['B_no_events_%d' % rep] = round((time.perf_counter() - t0) / 96 * 1e3, 4)
    # C: replay only (no clone / pending)
    graphs = stp.graphs
    streams = [torch.cuda.Stream() for _ in range(2)]
    def loop2(n):
        for k in range(n):
            with torch.cuda.stream(streams[k % 2]):
                graphs[k % 2][0].replay()
    loop2(8); torch.cuda.synchronize(); t0 = time.perf_counter(); loop2(96); torch.cuda.synchronize()
    out['C_replay_only_%d' % rep] = round((time.perf_counter() - t0) / 96 * 1e3, 4)
    # D: replay + clone
    def loop3(n):
        for k in range(n):
            with torch.cuda.stream(streams[k % 2]):
                graphs[k % 2][0].replay(); c = graphs[k % 2][1].clone()
    loop3(8); torch.cuda.synchronize(); t0 = time.perf_counter(); loop3(96); torch.cuda.synchronize()
    out['D_replay_clone_%d' % rep] = round((time.perf_counter() - t0) / 96 * 1e3, 4)
    # E: replay + clone + division
    def loop4(n):
        for k in range(n):
            with torch.cuda.stream(streams[k % 2]):
                graphs[k % 2][0].replay(); c = graphs[k % 2][1].clone(); q = c[0] / c[1]
    loop4(8); torch.cuda.synchronize(); t0 = time.perf_counter(); loop4(96); torch.cuda.synchronize()
    out['E_replay_clone_div_%d' % rep] = round((time.perf_counter() - t0) / 96 * 1e3, 4)
    # F: direct calls on the B=4 views, 2 streams, with all_reduce_mean_dice
    def loop5(n):
        for k in range(n):
            with torch.cuda.stream(streams[k % 2]):
                nd.all_reduce_mean_dice(ne.fused.warp_dice(m1, t1, x1), async_op=True)
    loop5(8); torch.cuda.synchronize(); t0 = time.perf_counter(); loop5(96); torch.cuda.synchronize()
    out['F_direct_views_%d' % rep] = round((time.perf_counter() - t0) / 96 * 1e3, 4)
print(json.dumps(out))
