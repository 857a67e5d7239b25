#!/usr/bin/env python3
"""First-contact diagnostics on a GPU box: prints parity numbers instead of asserting."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import master_amd as ma
import oracle

def rand_rays(scene, n, rng):
    lo = scene.positions.min(0); hi = scene.positions.max(0)
    o = np.zeros(n, ma.SURFACE_DTYPE)
    o["position"] = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    g = rng.normal(size=(n, 3)); g /= np.linalg.norm(g, axis=1, keepdims=True)
    o["gnormal"] = g.astype(np.float32)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)

def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "CornellBoxDiffuse"
    max_path = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    s = ma.Scene.load(os.path.join(ROOT, "scenes", name + ".miscene"))
    t = time.time(); pt = ma.PathTracing(s, max_path=max_path); print("create %.3fs" % (time.time() - t))
    orc = oracle.Oracle(s, max_path=max_path)
    gi, oi = pt.bvh_info(), orc.bvh_info()
    print("bvh gpu: nodes %d depth %d stack %d build %.3f ms | oracle: nodes %d depth %d" % (gi.n_nodes, gi.max_depth, gi.stack_entries, gi.build_ms, oi.n_nodes, oi.max_depth))
    print("scene bounds equal:", list(gi.scene_lo) == list(oi.scene_lo), list(gi.scene_hi) == list(oi.scene_hi))
    gn, gs, gm = pt.bvh(); on, os_, om = orc.bvh()
    print("morton equal", np.array_equal(gm, om), "order equal", np.array_equal(gs, os_), "nodes equal", gn.tobytes() == on.tobytes())
    if gn.tobytes() != on.tobytes():
        for f in gn.dtype.names:
            if f == "reserved": continue
            bad = np.nonzero(np.any(np.atleast_2d(gn[f] != on[f]).reshape(len(gn), -1), axis=1))[0]
            print("  field", f, "mismatch nodes", bad[:10], len(bad))
    rng = np.random.default_rng(0)
    o, d = rand_rays(s, 200000, rng)
    gh, gt, gp = pt.intersect(o, d); oh, ot, op = orc.intersect(o, d)
    print("intersect: prim equal %.6f  t equal %.6f  hits bytes equal %s  hit frac %.3f" % ((gp == op).mean(), (gt == ot).mean(), gh.tobytes() == oh.tobytes(), (gp != 0xFFFFFFFF).mean()))
    if gh.tobytes() != oh.tobytes():
        for f in gh.dtype.names:
            a, b = gh[f].reshape(len(gh), -1), oh[f].reshape(len(oh), -1)
            bad = np.any(a != b, axis=1) & ~(np.all(np.isnan(a.astype(np.float64)), axis=1))
            if f != "material_id":
                print("  field", f, "mismatch", bad.sum(), "max abs diff", np.nanmax(np.abs(a.astype(np.float64) - b)))
            else:
                print("  field", f, "mismatch", bad.sum())
    tg = np.zeros(len(o), ma.SURFACE_DTYPE); o2, _ = rand_rays(s, len(o), rng); tg["position"] = o2["position"]; tg["gnormal"] = o2["gnormal"]
    gv, ov = pt.occluded(o, tg), orc.occluded(o, tg)
    print("occluded equal %.6f visible frac %.3f" % ((gv == ov).mean(), gv.mean()))
    W = H = 64
    xy = np.stack(np.meshgrid(np.arange(W), np.arange(H)), -1).reshape(-1, 2).astype(np.uint32)
    xy = np.tile(xy, (8, 1)); si = np.repeat(np.arange(8, dtype=np.uint64), W * H)
    gr, gc = pt.trace_paths(W, H, xy, si, seed=7); orr, oc = orc.trace_paths(W, H, xy, si, seed=7)
    fin = np.isfinite(orr).all(1)
    exact = (gr.view(np.uint32) == orr.view(np.uint32)).all(1)
    close = np.isclose(gr, orr, rtol=1e-4, atol=1e-6).all(1)
    print("trace_paths n=%d exact %.6f close %.6f counts equal %.6f mean gpu %s mean orc %s" % (len(xy), exact.mean(), close.mean(), (gc == oc).all(1).mean(), gr[fin].mean(0), orr[fin].mean(0)))
    bad = np.nonzero(~exact)[0][:5]
    for i in bad: print("   path", i, xy[i], si[i], gr[i], orr[i], gc[i], oc[i])
    for kern in (ma.KERNEL_MEGA_LDS, ma.KERNEL_MEGA_GLOBAL):
        try: pt.set_kernel(kern)
        except ma.MiError as e: print("kernel", kern, "unavailable:", e); continue
        img = pt.render_rgbn(128, 128, spp=16, seed=1); st = pt.last_stats
        ref = orc.render_rgbn(128, 128, spp=16, seed=1); so = orc.last_stats
        print("kernel %d render 128x128x16: equal bytes %s max abs diff %.3g | gpu basic %d shadow %d err %d paths %d | orc basic %d shadow %d err %d | %.3f ms trace %.3f ms" % (
            kern, img.tobytes() == ref.tobytes(), np.abs(img - ref).max(), st.num_basic_rays, st.num_shadow_rays, st.numeric_errors, st.num_paths, so.num_basic_rays, so.num_shadow_rays, so.numeric_errors, st.gpu_ms, st.trace_ms))
        for spp in (64, 1024):
            img = pt.render_rgbn(512, 512, spp=spp, seed=3); st = pt.last_stats
            print("   512x512x%d: %.2f ms (trace %.2f) basic %d shadow %d -> %.1f Msamples/s, %.1f Mrays/s" % (spp, st.gpu_ms, st.trace_ms, st.num_basic_rays, st.num_shadow_rays, st.num_basic_rays / st.trace_ms / 1e3, (st.num_basic_rays + st.num_shadow_rays) / st.trace_ms / 1e3))

if __name__ == "__main__":
    main()
