"""TA/L1 hit bandwidth for the interpn lane pattern (8 lanes x 16 B per row): bytes per clock per CU."""
import json, torch
import neurite_amd as ne
lib = ne._lib.lib()
dev = torch.device('cuda:0')
sink = torch.zeros(256, device=dev)
CUS, CLK = 256, 2.4e9
for blocks in (256, 512, 1024, 2048):
    for rows in (32, 64, 128, 1024):      # window = rows * 128 B: 4, 8, 16 KB (L1) and 128 KB (L2)
        src = torch.rand(blocks * rows * 32, device=dev)
        for pattern in (0, 1, 2):
            iters = 2000
            for _ in range(2):
                lib.nrt_membench_l1_f32(ne._lib.ptr(src), ne._lib.ptr(sink), rows, iters, pattern, blocks, ne._lib.stream_ptr(dev))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            lib.nrt_membench_l1_f32(ne._lib.ptr(src), ne._lib.ptr(sink), rows, iters, pattern, blocks, ne._lib.stream_ptr(dev))
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            nbytes = blocks * 256 * 16 * 8 * iters
            print(json.dumps({'blocks': blocks, 'window_KB': rows * 128 // 1024, 'pattern': pattern, 'ms': round(ms, 3),
                              'TBps': round(nbytes / ms / 1e9, 2), 'B_per_clk_per_CU_at_2.4GHz': round(nbytes / (ms * 1e-3) / CUS / CLK, 1)}))
