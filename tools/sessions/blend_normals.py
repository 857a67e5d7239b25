#!/usr/bin/env python3
"""Per mesh of a .blend: how do the stored vertex normals (MVert.no, what the importer hands the loader) relate to the faces?
   python tools/blend_normals.py models/TestCase35.blend ...   (r03: why TestCase35/37/38/39/41 miss the normalisation)"""
import struct, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from blend_dump import Blend, idname

def mesh_report(bl, b):
    o = b[5]; g = lambda f: bl.get(o, 'Mesh', f)
    tv, tp, tl = g('totvert'), g('totpoly'), g('totloop')
    def arr(ptr, sname, n):
        blk = bl.by_ptr.get(ptr)
        if not blk: return None, 0
        return blk[5], bl.type_len[sname]
    mv, szv = arr(g('mvert'), 'MVert', tv); ml, szl = arr(g('mloop'), 'MLoop', tl); mp, szp = arr(g('mpoly'), 'MPoly', tp)
    if mv is None or ml is None or mp is None: return None
    co = np.zeros((tv, 3)); no = np.zeros((tv, 3))
    oco = bl.offset('MVert', 'co')[0]; ono = bl.offset('MVert', 'no')[0]
    for i in range(tv):
        co[i] = struct.unpack(bl.e + '3f', bl.data[mv + i * szv + oco: mv + i * szv + oco + 12])
        no[i] = np.array(struct.unpack(bl.e + '3h', bl.data[mv + i * szv + ono: mv + i * szv + ono + 6])) / 32767.0
    ov = bl.offset('MLoop', 'v')[0]
    loops = np.array([struct.unpack(bl.e + 'i', bl.data[ml + i * szl + ov: ml + i * szl + ov + 4])[0] for i in range(tl)])
    ols, otl, ofl = bl.offset('MPoly', 'loopstart')[0], bl.offset('MPoly', 'totloop')[0], bl.offset('MPoly', 'flag')[0]
    smooth = 0; cosines = []; flipped = 0; zero_no = int((np.linalg.norm(no, axis=1) < 0.5).sum()); ngon = {}
    used = np.zeros(tv, bool)
    for p in range(tp):
        ls, n = struct.unpack(bl.e + 'ii', bl.data[mp + p * szp + ols: mp + p * szp + ols + 8])
        fl = bl.data[mp + p * szp + ofl]
        smooth += fl & 1
        ngon[n] = ngon.get(n, 0) + 1
        vs = loops[ls:ls + n]; used[vs] = True
        P = co[vs]
        fn = np.zeros(3)
        for k in range(n):  # Newell
            a, c = P[k], P[(k + 1) % n]
            fn += np.array([(a[1] - c[1]) * (a[2] + c[2]), (a[2] - c[2]) * (a[0] + c[0]), (a[0] - c[0]) * (a[1] + c[1])])
        ln = np.linalg.norm(fn)
        if ln == 0: continue
        fn /= ln
        for v in vs:
            l = np.linalg.norm(no[v])
            if l > 0:
                c = float(no[v] @ fn / l); cosines.append(c); flipped += c < 0
    cs = np.array(cosines) if cosines else np.zeros(1)
    return dict(name=idname(bl, b), verts=tv, polys=tp, loops=tl, ngons=ngon, smooth_polys=smooth, unused_verts=int((~used).sum()), zero_normals=zero_no,
                cos_vertex_vs_face_normal=dict(min=float(cs.min()), mean=float(cs.mean()), frac_below_0_99=float((cs < 0.99).mean()), flipped=int(flipped)))

for path in sys.argv[1:]:
    bl = Blend(path)
    print("==", os.path.basename(path))
    for b in bl.blocks_of(b'OB\0\0'):
        o = b[5]
        if bl.get(o, 'Object', 'type') == 1:
            m = np.array(bl.get(o, 'Object', 'obmat')).reshape(4, 4)
            print("  object %-16s det(obmat3) %+.4g  parent %s" % (idname(bl, b), np.linalg.det(m[:3, :3]), "yes" if bl.get(o, 'Object', 'parent') else "no"))
    for b in bl.blocks_of(b'ME\0\0'):
        r = mesh_report(bl, b)
        print("  ", r)
