import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.environ["MI_PT_LIB"] = os.path.join(ROOT, "master_amd", "libmi_pt_phase.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, master_amd as ma
names = ["regeneration", "closest traversal", "querySurface+logic", "NEE set-up", "shadow traversal", "BSDF sample", "commit", "loop overhead"]
for spec, (W, H, spp, mp) in (("CornellBoxDiffuse", (512, 512, 64, 8)), ("CornellBoxSpecular", (512, 512, 32, ma.PTRDIFF_MAX))):
    s = ma.Scene.load(os.path.join(ROOT, "scenes", spec + ".miscene"))
    pt = ma.PathTracing(s, max_path=mp)
    pt.render_rgbn(W, H, spp=spp, seed=1); st = pt.last_stats
    tot = float(sum(st.phase_cycles))
    print(spec, "trace %.2f ms, %.0f Msamples/s (diagnostic build)" % (st.trace_ms, st.num_basic_rays / st.trace_ms / 1e3))
    for k in range(8): print("   %-20s %5.1f %%" % (names[k], 100.0 * st.phase_cycles[k] / tot))
