import os, sys
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+"/tests")
import numpy as np, master_amd as ma, oracle
s=ma.Scene.load(ROOT+"/scenes/MirrorAndAreaLight.miscene")
rng=np.random.default_rng(5); n=5000
xy=np.stack([rng.integers(0,64,n),rng.integers(0,48,n)],1).astype(np.uint32); si=rng.integers(0,32,n).astype(np.uint64)
os.environ["MI_BPT_STAGED"]="1"; a=ma.PathTracing(s,beta=2.0); ra,sa,ca=a.bpt_trace_paths(64,48,xy,si,seed=7)
os.environ["MI_BPT_STAGED"]="0"; b=ma.PathTracing(s,beta=2.0); rb,sb,cb=b.bpt_trace_paths(64,48,xy,si,seed=7)
orr,os_,oc=oracle.Oracle(s,beta=2.0).bpt_trace_paths(64,48,xy,si,seed=7)
bad=np.nonzero(~((ra.view(np.uint32)==rb.view(np.uint32)).all(1)))[0]
print("staged vs one-kernel radiance mismatches", len(bad), "splat mismatches", (~(sa.view(np.uint32)==sb.view(np.uint32)).all(1)).sum(), "count mismatches", (~(ca==cb).all(1)).sum())
print("one-kernel vs oracle close", np.isclose(rb,orr,rtol=5e-5,atol=1e-6).all(1).mean())
for i in bad[:8]: print(i, "staged", ra[i], ca[i], "one", rb[i], cb[i], "oracle", orr[i], oc[i])
