import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, master_amd as ma, oracle
oracle.build = lambda: None
s = ma.Scene.load(os.path.join(ROOT, "scenes", "TestCaseFurnace.miscene"))
W = H = 64
xy = np.stack(np.meshgrid(np.arange(W), np.arange(H)), -1).reshape(-1, 2).astype(np.uint32); si = np.zeros(len(xy), np.uint64)
pt = ma.PathTracing(s, max_path=2, lights=0.0); orc = oracle.Oracle(s, max_path=2, lights=0.0)
g, gc = pt.trace_paths(W, H, xy, si, seed=7); o, oc = orc.trace_paths(W, H, xy, si, seed=7)
for k, nm in enumerate(["org.y", "org.z", "t1"]):
    print(nm, "exact frac", (g[:, k].view(np.uint32) == o[:, k].view(np.uint32)).mean())
bad = np.nonzero((g.view(np.uint32) != o.view(np.uint32)).any(1))[0][:6]
for i in bad: print(i, [float(x).hex() for x in g[i]], [float(x).hex() for x in o[i]])
