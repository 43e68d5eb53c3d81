"""development aid: direct launches vs hipGraph replay for the multi-kernel dwarfs at small sizes"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops

def wall(fn, iters=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6

for lg in (14, 16, 18, 20):
    n = 1 << lg
    keys = ops.gen_uniform_u32(n, 1, 0, 2**32 - 1)
    plan = ops.RadixSort(n, 8)
    direct = wall(lambda: plan.launch(keys))
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): plan.launch(keys)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): plan.launch(keys)
    graph = wall(g.replay)
    b = ops.gen_uniform_u32(n, 2, 0, n - 1); p = ops.gen_uniform_u32(n, 3, 0, n - 1)
    j = ops.HashJoin(n, n)
    def both(): j.build(b); j.probe(p)
    jd = wall(both)
    with torch.cuda.stream(side): both()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2): both()
    jg = wall(g2.replay)
    print(f"n=2^{lg}: sort direct {direct:.1f} us  graph {graph:.1f} us | join direct {jd:.1f} us  graph {jg:.1f} us", flush=True)
