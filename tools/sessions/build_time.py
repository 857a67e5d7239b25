import os, sys
sys.path.insert(0, "/root/repo")
import master_amd as ma
from master_amd import scenegen as sb
for spec in ("atrium", "atrium:2000000"):
    s = sb.load(spec)
    for pairs in ("0", "1"):
        os.environ["MI_PT_PAIRS"] = pairs
        pt = ma.PathTracing(s)
        pt2 = ma.PathTracing(s)
        print(spec, "pairs", pairs, "build_ms %.2f" % pt2.bvh_info().build_ms, flush=True)
