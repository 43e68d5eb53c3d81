"""development aid: 2^LG-key sort time (HIP events, drop-max-mean of 9, refresh copy subtracted) for DBHIP_LIB, 8- and 4-bit"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << lg
keys0 = ops.gen_uniform_u32(n, 42, 0, 2**32 - 1)
keys = keys0.clone()
def ev(fn, k=9):
    out = []
    for _ in range(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3)
    out.sort()
    return sum(out[:-1]) / (len(out) - 1)
res = []
ref = torch.sort(keys0.to(torch.int64) & 0xFFFFFFFF).values
for bits in (8, 4):
    plan = ops.RadixSort(n, bits)
    def run():
        keys.copy_(keys0); plan.launch(keys)
    run(); run()
    t = ev(run) - ev(lambda: keys.copy_(keys0))
    run(); torch.cuda.synchronize()
    ok = bool(torch.equal(keys.to(torch.int64) & 0xFFFFFFFF, ref))
    res.append(f"{bits}-bit {t:7.1f} us {'ok' if ok else 'WRONG'}")
print(f"{os.environ.get('DBHIP_LIB', 'default').split('libdbhip_')[-1]:24s} 2^{lg}: " + "   ".join(res), flush=True)
