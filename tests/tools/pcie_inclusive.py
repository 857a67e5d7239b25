import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import master_amd as ma
s = ma.Scene.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scenes", "CornellBoxDiffuse.miscene"))
pt = ma.PathTracing(s, max_path=8)
pt.render_rgbn(512, 512, spp=64, seed=1)
best = 1e9
for k in range(3):
    t = time.perf_counter(); pt.render_rgbn(512, 512, spp=1024, seed=0x5EED, sample_offset=k * 1024); dt = time.perf_counter() - t
    st = pt.last_stats
    best = min(best, dt)
print("host-buffer call: %.2f ms wall (device %.2f ms), %.0f Msamples/s PCIe-inclusive" % (best * 1e3, st.gpu_ms, st.num_basic_rays / best / 1e6))
