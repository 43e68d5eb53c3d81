import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ab import times, dropmax
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 27
n = 1 << lg
build = ops.gen_uniform_u32(n, 42, 0, n - 1)
probe = ops.gen_uniform_u32(n, 43, 0, n - 1)
ids = torch.arange(n, device="cuda", dtype=torch.int64).to(torch.int32)
rj = ops.RadixJoin(n, n)
for name, rid in (("row ids generated", None), ("row ids as a column", ids)):
    pb = dropmax(times(lambda: rj.partition_build(build, rid), 7, warm=1))
    pp = dropmax(times(lambda: rj.partition_probe(probe, rid), 7, warm=1))
    m = dropmax(times(rj.match, 7, warm=1))
    print(f"2^{lg} {name:22s}: partition build side {pb:8.1f} us, probe side {pp:8.1f} us, match {m:8.1f} us", flush=True)
