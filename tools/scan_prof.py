import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dwarf_bench_amd import ops
n = 1 << 28
src = ops.gen_uniform_u32(n, 42, 1, 10000)
plan = ops.CopyIfLt(n)
for filt in (5, 5001):
    for it in range(3):
        plan.launch(src, filt)
        torch.cuda.synchronize()
        hdr = plan.ws[:256].view(torch.int32).cpu().tolist()
        pad = hdr[4:]
        st = pad[8:15]
        print(f"filter={filt} it={it}: us per phase (WG owning tile 100) idle={st[0]/100:.1f} datawait+count={st[1]/100:.1f} control={st[2]/100:.1f} prefetchissue+flush={st[3]/100:.1f} ticketwait={st[5]/100:.1f} resolve_p2={st[6]/100:.1f} total={sum(st)/100:.1f} | chipwide evals: incomplete={pad[16]} no-inclusive={pad[17]} resolved={pad[18]}")
