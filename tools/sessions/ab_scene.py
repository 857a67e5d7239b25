#!/usr/bin/env python3
"""A/B library variants on arbitrary scenes in one GPU session: tools/ab_scene.py lib1.so lib2.so -- scene W H spp [scene W H spp ...]"""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
args = sys.argv[1:]; i = args.index("--"); libs, rest = args[:i], args[i + 1:]
code = r'''
import os, sys
sys.path.insert(0, %r)
import master_amd as ma
from master_amd import scenegen
spec, W, H, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
s = scenegen.load(spec) if spec.split(":")[0] in scenegen.SCENES else ma.Scene.load(os.path.join(%r, "scenes", spec + ".miscene"))
pt = ma.PathTracing(s)
pt.render_rgbn(W, H, spp=2, seed=1)
best = 0
for r in range(3):
    pt.render_rgbn(W, H, spp=spp, seed=1); st = pt.last_stats
    best = max(best, st.num_basic_rays / st.trace_ms / 1e3)
print("%%-24s %%-28s %%8.1f Msamples/s (stack %%d LDS entries, depth %%d)" %% (spec, os.path.basename(os.environ["MI_PT_LIB"]), best, pt.bvh_info().stack_entries, pt.bvh_info().max_depth))
''' % (ROOT, ROOT)
for k in range(0, len(rest), 4):
    for lib in libs:
        env = dict(os.environ, MI_PT_LIB=os.path.join(ROOT, "master_amd", lib))
        r = subprocess.run([sys.executable, "-c", code] + rest[k:k + 4], env=env, capture_output=True, text=True)
        print(r.stdout.strip() or ("FAILED " + lib + " " + r.stderr[-300:]), flush=True)
