#!/usr/bin/env python3
"""Independent .blend (Blender 2.7x) SDNA walker used to cross-check the C++ reader
(master_amd/csrc/blend_reader.cpp).  Not part of the product path.

Usage: blend_dump.py file.blend            -> prints objects / meshes / materials / lamps / cameras
"""
import struct, sys, json

class Blend:
    def __init__(self, path):
        d = open(path, 'rb').read()
        assert d[:7] == b'BLENDER', 'not a .blend (maybe gzip-compressed)'
        self.psz = 8 if d[7:8] == b'-' else 4
        self.le = d[8:9] == b'v'
        self.ver = int(d[9:12])
        e = '<' if self.le else '>'
        self.e = e
        self.data = d
        off = 12
        self.blocks = []
        while True:
            code = d[off:off+4]
            if self.psz == 8:
                size, old, sdna, count = struct.unpack(e+'iQii', d[off+4:off+24]); hs = 24
            else:
                size, old, sdna, count = struct.unpack(e+'iIii', d[off+4:off+20]); hs = 20
            self.blocks.append((code, size, old, sdna, count, off+hs))
            if code == b'ENDB': break
            off += hs + size
        self.by_ptr = {b[2]: b for b in self.blocks if b[2]}
        self._parse_dna()

    def _parse_dna(self):
        b = [b for b in self.blocks if b[0] == b'DNA1'][0]
        d = self.data; o = b[5]; e = self.e
        assert d[o:o+4] == b'SDNA'; o += 4
        def strs(tag, o):
            assert d[o:o+4] == tag, (tag, d[o:o+4]); o += 4
            n, = struct.unpack(e+'i', d[o:o+4]); o += 4
            out = []
            for _ in range(n):
                z = d.index(b'\0', o); out.append(d[o:z].decode()); o = z+1
            return out, (o+3) & ~3
        self.names, o = strs(b'NAME', o)
        self.types, o = strs(b'TYPE', o)
        assert d[o:o+4] == b'TLEN'; o += 4
        self.tlen = list(struct.unpack(e+'%dh' % len(self.types), d[o:o+2*len(self.types)])); o += 2*len(self.types); o = (o+3)&~3
        assert d[o:o+4] == b'STRC'; o += 4
        ns, = struct.unpack(e+'i', d[o:o+4]); o += 4
        self.structs = []; self.struct_by_name = {}
        for i in range(ns):
            t, nf = struct.unpack(e+'hh', d[o:o+4]); o += 4
            fields = []
            for _ in range(nf):
                ft, fn = struct.unpack(e+'hh', d[o:o+4]); o += 4
                fields.append((self.types[ft], self.names[fn]))
            self.structs.append((self.types[t], fields))
            self.struct_by_name[self.types[t]] = i
        self.type_len = dict(zip(self.types, self.tlen))

    def field_size(self, ftype, fname):
        n = 1
        base = fname
        while base.endswith(']'):
            i = base.rindex('['); n *= int(base[i+1:-1]); base = base[:i]
        if base.startswith('*') or base.startswith('(*'):
            return self.psz * n, n
        return self.type_len[ftype] * n, n

    def offset(self, sname, fname):
        """byte offset + (type, full name) of field whose bare name is fname."""
        off = 0
        for ft, fn in self.structs[self.struct_by_name[sname]][1]:
            bare = fn.lstrip('*').split('[')[0]
            if fn.startswith('(*'): bare = fn[2:fn.index(')')]
            sz, n = self.field_size(ft, fn)
            if bare == fname:
                return off, ft, fn, sz
            off += sz
        raise KeyError((sname, fname))

    def get(self, base, sname, fname, fmt=None):
        off, ft, fn, sz = self.offset(sname, fname)
        raw = self.data[base+off: base+off+sz]
        if fn.startswith('*'):
            return struct.unpack(self.e + ('Q' if self.psz == 8 else 'I'), raw[:self.psz])[0]
        code = {'float': 'f', 'int': 'i', 'short': 'h', 'char': 'b', 'double': 'd', 'uchar':'B','ushort':'H'}.get(ft)
        if code is None:
            return raw
        n = sz // struct.calcsize(code)
        v = struct.unpack(self.e + '%d%s' % (n, code), raw)
        return v[0] if n == 1 else list(v)

    def blocks_of(self, code):
        return [b for b in self.blocks if b[0] == code]

    def sname(self, b):
        return self.structs[b[3]][0]

def idname(bl, b):
    raw = bl.get(b[5], 'ID', 'name')
    return bytes((x & 0xff) for x in raw).split(b'\0')[0].decode()

def dump(path):
    bl = Blend(path)
    out = {'version': bl.ver, 'psz': bl.psz}
    objs = []
    for b in bl.blocks_of(b'OB\0\0'):
        o = b[5]
        objs.append(dict(name=idname(bl, b), type=bl.get(o, 'Object', 'type'), data=bl.get(o, 'Object', 'data'),
                         obmat=bl.get(o, 'Object', 'obmat'), parent=bl.get(o,'Object','parent'),
                         totcol=bl.get(o,'Object','totcol'), mat=bl.get(o,'Object','mat'), matbits=bl.get(o,'Object','matbits')))
    out['objects'] = objs
    mats = []
    for b in bl.blocks_of(b'MA\0\0'):
        o = b[5]
        g = lambda f: bl.get(o, 'Material', f)
        mats.append(dict(name=idname(bl, b), ptr=b[2], rgb=[g('r'), g('g'), g('b')], spec_rgb=[g('specr'), g('specg'), g('specb')],
                         mir=[g('mirr'), g('mirg'), g('mirb')], ref=g('ref'), spec=g('spec'), har=g('har'), mode=g('mode'),
                         ang=g('ang'), ray_mirror=g('ray_mirror'), alpha=g('alpha'), emit=g('emit')))
    out['materials'] = mats
    lamps = []
    for b in bl.blocks_of(b'LA\0\0'):
        o = b[5]; g = lambda f: bl.get(o, 'Lamp', f)
        lamps.append(dict(name=idname(bl, b), ptr=b[2], type=g('type'), rgb=[g('r'), g('g'), g('b')], energy=g('energy'),
                          area_shape=g('area_shape'), area_size=g('area_size'), area_sizey=g('area_sizey'), mode=g('mode'), dist=g('dist')))
    out['lamps'] = lamps
    cams = []
    for b in bl.blocks_of(b'CA\0\0'):
        o = b[5]; g = lambda f: bl.get(o, 'Camera', f)
        cams.append(dict(name=idname(bl, b), ptr=b[2], type=g('type'), lens=g('lens'), sensor_x=g('sensor_x'), sensor_y=g('sensor_y'),
                         sensor_fit=g('sensor_fit'), clipsta=g('clipsta'), clipend=g('clipend')))
    out['cameras'] = cams
    meshes = []
    for b in bl.blocks_of(b'ME\0\0'):
        o = b[5]; g = lambda f: bl.get(o, 'Mesh', f)
        meshes.append(dict(name=idname(bl, b), ptr=b[2], totvert=g('totvert'), totpoly=g('totpoly'), totloop=g('totloop'),
                           totface=g('totface'), totcol=g('totcol'), mat=g('mat')))
    out['meshes'] = meshes
    return bl, out

if __name__ == '__main__':
    bl, out = dump(sys.argv[1])
    for k in ('objects', 'materials', 'lamps', 'cameras', 'meshes'):
        print('==', k)
        for x in out[k]:
            print('  ', {kk: (['%.4g' % v for v in vv] if isinstance(vv, list) and vv and isinstance(vv[0], float) else vv) for kk, vv in x.items()})
