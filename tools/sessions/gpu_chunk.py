import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
code = r'''
import os, sys
sys.path.insert(0, %r)
import master_amd as ma
s = ma.Scene.load(os.path.join(%r, "scenes", "CornellBoxDiffuse.miscene"))
pt = ma.PathTracing(s, max_path=8)
pt.render_rgbn(512, 512, spp=64, seed=1)
best = 0
for r in range(3):
    pt.render_rgbn(512, 512, spp=1024, seed=1); st = pt.last_stats
    best = max(best, st.num_basic_rays / st.trace_ms / 1e3)
print("chunk_spp %%4s  %%8.1f Msamples/s  gpu_ms %%.2f" %% (os.environ.get("MI_PT_CHUNK_SPP", "auto"), best, st.gpu_ms))
''' % (ROOT, ROOT)
for c in ("auto", "8", "16", "32", "64", "256", "1024"):
    env = dict(os.environ)
    if c != "auto": env["MI_PT_CHUNK_SPP"] = c
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-300:], flush=True)
