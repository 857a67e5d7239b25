#!/usr/bin/env python3
"""Wave-level cost model of the megakernel's trip structure, driven by real traversal event sequences from the oracle.

Compares (a) the current scheme: every trip = [closest traversal of all lanes][shade][shadow traversal][sample], each
traversal running until its slowest lane is done, with (b) dynamic phase scheduling inside the wave: traversal state
persists, the wave keeps stepping while at least `theta` lanes traverse, finished lanes wait and are shaded together.
Costs are in node-body units (leaf body 1.6, shading phases from the phase-stamp profile).

    python tests/lab/wave_sim.py [scene] [max_path]
"""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, HERE)
import numpy as np  # noqa: E402
import run_lab  # noqa: E402

C_NODE, C_LEAF, C_SHADE, C_SAMPLE, C_REGEN = 1.0, 1.6, 14.0, 20.0, 4.6


def record(scene_name, max_path, n_paths=6000):
    run_lab.build()
    import oracle
    import master_amd as ma
    oracle.ORACLE_LIB = run_lab.SO; oracle.build = lambda: run_lab.SO
    L = oracle.lib()
    L.lab_log.argtypes = [C.c_void_p, C.c_uint64]; L.lab_log_size.restype = C.c_uint64
    s = ma.Scene.load(os.path.join(ROOT, "scenes", scene_name + ".miscene"))
    o = oracle.Oracle(s, max_path=max_path)
    rng = np.random.default_rng(1)
    W, H = 64, 64
    buf = np.zeros(64 << 20, np.uint8)
    L.lab_log(buf.ctypes.data_as(C.c_void_p), buf.size)
    paths = []
    # 8x8 tiles like the kernel: consecutive paths of one wave come from one tile
    for t in range(n_paths // 64):
        tx, ty = rng.integers(0, W // 8), rng.integers(0, H // 8)
        xy = np.array([[tx * 8 + i % 8, ty * 8 + i // 8] for i in range(64)], np.uint32)
        si = np.full(64, t, np.uint64)
        for i in range(64):
            L.lab_log_mark()
            o.trace_paths(W, H, xy[i:i + 1], si[i:i + 1], seed=3)
    n = L.lab_log_size()
    ev = buf[:n].tobytes().decode()
    L.lab_log(None, 0)
    for p in ev.split("P")[1:]:
        # segments: each 'C' starts a segment; an optional 'S' ray follows inside the same segment
        segs = []
        for c in p.split("C")[1:]:
            parts = c.split("S")
            segs.append((runs(parts[0]), runs(parts[1]) if len(parts) > 1 else None))
        paths.append(segs)
    return paths


def runs(s):
    """event string -> list of node-run lengths, each followed by one leaf test except possibly the last."""
    out = []; k = 0
    for ch in s:
        if ch == "n": k += 1
        elif ch == "l": out.append((k, 1)); k = 0
    if k: out.append((k, 0))
    return out


def trav_cost_static(lanes_runs):
    """while-while loop over a set of rays started together: per outer round the node loop runs max(run) bodies and
    one leaf body if any lane has a leaf."""
    cost = 0.0; useful = 0.0
    depth = max((len(r) for r in lanes_runs), default=0)
    for k in range(depth):
        act = [r[k] for r in lanes_runs if len(r) > k]
        cost += max(a[0] for a in act) * C_NODE + (C_LEAF if any(a[1] for a in act) else 0.0)
        useful += sum(a[0] * C_NODE + a[1] * C_LEAF for a in act)
    return cost, useful


def simulate_static(paths, n_waves=40):
    it = iter(paths); total = 0.0; segs = 0; useful = 0.0; trav = 0.0
    for _ in range(n_waves):
        lanes = []
        for _ in range(64):
            try: lanes.append(list(next(it)))
            except StopIteration: break
        pool = [list(next(it, [])) for _ in range(64 * 3)]  # regeneration pool of the chunk
        while any(lanes):
            cur = [l[0] for l in lanes if l]
            c, u = trav_cost_static([s[0] for s in cur]); total += c; useful += u; trav += c
            total += C_SHADE
            sh = [s[1] for s in cur if s[1] is not None]
            if sh:
                c, u = trav_cost_static(sh); total += c; useful += u; trav += c
            total += C_SAMPLE + C_REGEN
            segs += len(cur)
            for i, l in enumerate(lanes):
                if l:
                    l.pop(0)
                    if not l and pool: lanes[i] = pool.pop()
    return total / segs, trav / segs, useful / segs


def simulate_dynamic(paths, theta, n_waves=40):
    """lane states: 0 idle/regen, 1 traversing (closest), 2 waiting for shade, 3 traversing (shadow), 4 waiting for sample."""
    it = iter(paths); total = 0.0; segs = 0; trav = 0.0
    for _ in range(n_waves):
        pool = []
        for _ in range(64 * 4):
            p = next(it, None)
            if p: pool.append(list(p))
        lane = [None] * 64
        def start(i):
            if pool:
                lane[i] = {"path": pool.pop(), "st": 1, "k": 0}
                lane[i]["runs"] = list(lane[i]["path"][0][0])
                if not lane[i]["runs"]: lane[i]["st"] = 2
            else: lane[i] = None
        for i in range(64): start(i)
        total += C_REGEN
        while any(l is not None for l in lane):
            tr = [l for l in lane if l and l["st"] in (1, 3)]
            w2 = [l for l in lane if l and l["st"] == 2]
            w4 = [l for l in lane if l and l["st"] == 4]
            if tr and (len(tr) >= theta or (not w2 and not w4)):
                # one outer round of the traversal loop for every traversing lane
                heads = [l["runs"][0] for l in tr]
                c = max(h[0] for h in heads) * C_NODE + (C_LEAF if any(h[1] for h in heads) else 0.0)
                total += c; trav += c
                for l in tr:
                    l["runs"].pop(0)
                    if not l["runs"]: l["st"] = 2 if l["st"] == 1 else 4
                continue
            # shade whichever waiting set is larger (ties: the earlier phase)
            if w2 and len(w2) >= len(w4):
                total += C_SHADE
                for l in w2:
                    sh = l["path"][0][1]
                    if sh: l["runs"] = list(sh); l["st"] = 3
                    else: l["st"] = 4
                    if l["st"] == 3 and not l["runs"]: l["st"] = 4
            else:
                total += C_SAMPLE
                regen = False
                for i, l in enumerate(lane):
                    if l and l["st"] == 4 and l in w4:
                        segs += 1
                        l["path"].pop(0)
                        if l["path"]:
                            l["runs"] = list(l["path"][0][0]); l["st"] = 1
                            if not l["runs"]: l["st"] = 2
                        else:
                            start(i); regen = True
                if regen: total += C_REGEN
    return total / segs, trav / segs


if __name__ == "__main__":
    scene = sys.argv[1] if len(sys.argv) > 1 else "CornellBoxDiffuse"
    mp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    paths = record(scene, mp)
    nseg = sum(len(p) for p in paths)
    print(scene, "paths", len(paths), "segments", nseg)
    c, t, u = simulate_static(paths)
    print("static  : cost/segment %.2f  traversal %.2f  (useful lane-work / 64 = %.2f)" % (c, t, u / 64))
    for theta in (8, 16, 24, 32, 40, 48):
        c2, t2 = simulate_dynamic(paths, theta)
        print("dynamic theta=%2d: cost/segment %.2f  traversal %.2f  -> speed-up %.2fx" % (theta, c2, t2, c / c2))


def simulate_tail(paths, K, min_ready=32, n_waves=40):
    """Static trip structure, but a traversal loop stops once at most K lanes are still walking (and at least min_ready
    lanes have a result): the stragglers keep their traversal state, sit out the following phases and continue in the same
    phase of the next trip.  Per lane the sequence of operations is unchanged (results stay bit-identical)."""
    it = iter(paths); total = 0.0; segs = 0; trav = 0.0

    def loop(parts):
        nonlocal total, trav
        n0 = len(parts)
        done = [l for l in parts if not l["runs"]]
        act = [l for l in parts if l["runs"]]
        while act:
            if len(act) <= K and len(done) >= min(min_ready, n0 - K) and len(done) > 0:
                break
            heads = [l["runs"][0] for l in act]
            c = max(h[0] for h in heads) * C_NODE + (C_LEAF if any(h[1] for h in heads) else 0.0)
            total += c; trav += c
            for l in act:
                l["runs"].pop(0)
            done += [l for l in act if not l["runs"]]
            act = [l for l in act if l["runs"]]
        return done

    for _ in range(n_waves):
        pool = []
        for _ in range(64 * 4):
            p = next(it, None)
            if p: pool.append(list(p))
        lane = [None] * 64

        def start(i):
            if pool:
                lane[i] = {"path": pool.pop(), "st": "C"}
                lane[i]["runs"] = list(lane[i]["path"][0][0])
            else:
                lane[i] = None
        for i in range(64): start(i)
        while any(l is not None for l in lane):
            cs = [l for l in lane if l and l["st"] == "C"]
            if cs:
                for l in loop(cs): l["st"] = "H"
            hs = [l for l in lane if l and l["st"] == "H"]
            if hs:
                total += C_SHADE
                for l in hs:
                    sh = l["path"][0][1]
                    if sh: l["runs"] = list(sh); l["st"] = "S"
                    else: l["st"] = "B"
            ss = [l for l in lane if l and l["st"] == "S"]
            if ss:
                for l in loop(ss): l["st"] = "B"
            bs = [i for i, l in enumerate(lane) if l and l["st"] == "B"]
            if bs:
                total += C_SAMPLE + C_REGEN
                for i in bs:
                    l = lane[i]; segs += 1
                    l["path"].pop(0)
                    if l["path"]: l["runs"] = list(l["path"][0][0]); l["st"] = "C"
                    else: start(i)
    return total / segs, trav / segs


if __name__ == "__main__" and os.environ.get("WAVE_SIM_TAIL", "1") != "0":
    for K in (0, 2, 4, 8, 12, 16):
        c3, t3 = simulate_tail(paths, K)
        print("tail K=%2d: cost/segment %.2f  traversal %.2f  -> speed-up %.2fx" % (K, c3, t3, c / c3))
