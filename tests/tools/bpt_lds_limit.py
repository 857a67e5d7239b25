import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import master_amd as ma
from master_amd import scenegen as sb
for n in (30, 60, 90, 100):
    s = sb.random_soup(n, seed=3)
    row = []
    for k in (ma.KERNEL_AUTO, ma.KERNEL_MEGA_GLOBAL):
        pt = ma.PathTracing(s, beta=2.0); pt.set_kernel(k)
        pt.bpt_render_rgbn(512, 512, spp=2, seed=1)
        best = 1e9
        for _ in range(2):
            pt.bpt_render_rgbn(512, 512, spp=16, seed=1); best = min(best, pt.last_stats.trace_ms)
        row.append(best)
    print("soup %d: BPT auto %.1f ms, HBM kernels %.1f ms (PT auto kernel %d)" % (s.indices.shape[0], row[0], row[1], ma.PathTracing(s).get_kernel()), flush=True)
