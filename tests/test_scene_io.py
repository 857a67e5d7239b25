"""Host-side data formats either side of the hot path: .miscene container, EXR (R,G,B,denom), the
.blend reader against an independent SDNA walker (tools/blend_dump.py) and the committed fixtures."""
import os
import sys

import numpy as np
import pytest

import master_amd as ma
from conftest import REFERENCE, ROOT, load_scene, scene_path

sys.path.insert(0, os.path.join(ROOT, "tools"))
needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "models")), reason="reference tree not present (GPU box)")


def scenes_equal(a, b):
    return (np.array_equal(a.positions, b.positions) and np.array_equal(a.tangents, b.tangents) and np.array_equal(a.indices, b.indices)
            and np.array_equal(a.mesh_tri_offset, b.mesh_tri_offset) and np.array_equal(a.mesh_material_id, b.mesh_material_id)
            and [bytes(m) for m in a.materials] == [bytes(m) for m in b.materials] and [bytes(l) for l in a.lights] == [bytes(l) for l in b.lights]
            and [bytes(c) for c in a.cameras] == [bytes(c) for c in b.cameras] and a.material_names == b.material_names)


def test_miscene_round_trip(tmp_path, cornell):
    p = str(tmp_path / "c.miscene")
    cornell.save(p)
    assert scenes_equal(cornell, ma.Scene.load(p))
    open(p, "r+b").truncate(200)
    with pytest.raises(ma.MiError):
        ma.Scene.load(p)


def test_cornell_scene_facts(cornell):
    """30 triangles + 2 for the light quad, 7 diffuse materials, one lamp, one camera."""
    s = cornell
    assert s.n_triangles == 32 and len(s.lights) == 1 and len(s.cameras) == 1
    assert [m.type for m in s.materials] == [ma.BSDF_CAMERA] + [ma.BSDF_DIFFUSE] * 7 + [ma.BSDF_LIGHT]
    assert sorted(s.material_names[1:8]) == sorted(["backWall", "ceiling", "floor", "leftWall", "rightWall", "shortBox", "tallBox"])
    l = s.lights[0]
    np.testing.assert_allclose(list(l.exitance), [0.67 * 40, 0.48 * 40, 0.16 * 40], rtol=1e-6)   # rgb * energy
    np.testing.assert_allclose(list(l.size), [0.5, 0.5])                                           # square lamp: area_size both ways
    np.testing.assert_allclose(list(l.tangent[3:6]), [0, 0, -1], atol=1e-6)                        # emits along -Z
    np.testing.assert_allclose(list(l.position), [-0.005, 0.03, 1.98], atol=1e-6)
    assert (s.mesh_material_id[-1] & 3) == ma.ENTITY_LIGHT and np.all((s.mesh_material_id[:-1] & 3) == ma.ENTITY_MESH)
    # material table order: cameras, scene materials, light BSDFs (loader.cpp:304-305,375-378,451-453)
    assert l.material_id >> 2 == len(s.materials) - 1 and s.materials[-1].light_id == 0
    # per-corner frames (loader.cpp:332-339): col1 = normal, col0 along the first edge, col2 = cross(n, col0)
    T = s.tangents.reshape(-1, 3, 3)
    tri0 = s.positions[s.indices[0]]
    e = tri0[1] - tri0[0]
    np.testing.assert_allclose(T[s.indices[0, 0], 0], e / np.linalg.norm(e), atol=1e-5)
    np.testing.assert_allclose(np.cross(T[:, 1], T[:, 0]), T[:, 2], atol=1e-5)


@needs_reference
@pytest.mark.parametrize("name", ["CornellBoxDiffuse", "CornellBoxSpecular", "TestCaseFurnace", "TestCase0"])
def test_committed_fixture_is_what_the_reader_produces(name):
    fresh = ma.Scene.load_blend(os.path.join(REFERENCE, "models", name + ".blend"))
    assert scenes_equal(fresh, load_scene(name))


@needs_reference
@pytest.mark.parametrize("name", ["CornellBoxDiffuse", "CornellBoxSpecular", "CornellBoxPhong", "LivingRoom"])
def test_blend_reader_against_independent_sdna_walker(name):
    import blend_dump

    path = os.path.join(REFERENCE, "models", name + ".blend")
    bl, d = blend_dump.dump(path)
    s = ma.Scene.load_blend(path)
    n_area = sum(1 for l in d["lamps"] if l["type"] == 4)
    tris_of = {m["ptr"]: m["totloop"] - 2 * m["totpoly"] for m in d["meshes"]}
    tri_expected = sum(tris_of.get(o["data"], 0) for o in d["objects"] if o["type"] == 1)  # a mesh datablock may be instanced by several objects
    assert len(s.lights) == n_area and len(s.cameras) == sum(1 for o in d["objects"] if o["type"] == 11)
    assert s.n_triangles == tri_expected + 2 * n_area
    by_name = {m["name"][2:]: m for m in d["materials"]}
    for i, m in enumerate(s.materials):
        src = by_name.get(s.material_names[i])
        if m.type in (ma.BSDF_CAMERA, ma.BSDF_LIGHT, ma.BSDF_SUN) or src is None:
            continue
        np.testing.assert_allclose(list(m.diffuse), src["rgb"], rtol=1e-6)
        if src["mode"] & 0x10000:
            assert m.type == ma.BSDF_TRANSMISSION and m.ior_internal == pytest.approx(src["ang"])
        elif src["mode"] & 0x40000:
            assert m.type == ma.BSDF_REFLECTION
        elif not any(src["spec_rgb"]):
            assert m.type == ma.BSDF_DIFFUSE
        else:
            assert m.type == ma.BSDF_PHONG and m.power == src["har"]
    cam = [c for c in d["cameras"]][0]
    assert s.cameras[0].fovx == pytest.approx(2 * np.arctan2(cam["sensor_x"], 2 * cam["lens"]), rel=1e-6)


def test_specular_scene_has_the_delta_bsdfs():
    s = load_scene("CornellBoxSpecular")
    kinds = {m.type for m in s.materials}
    assert ma.BSDF_REFLECTION in kinds and ma.BSDF_TRANSMISSION in kinds
    glass = [m for m in s.materials if m.type == ma.BSDF_TRANSMISSION][0]
    assert glass.ior_internal == pytest.approx(2.0) and glass.ior_external == 1.0   # SURVEY App. B, loader.cpp:381-382


def test_exr_round_trip_and_layout(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 5, (13, 17, 4)).astype(np.float32)
    p = str(tmp_path / "a.exr")
    ma.save_exr(p, img, {"technique": "PT", "num_samples": 3, "max_path": 8})
    back = ma.load_exr(p)
    assert np.array_equal(back, img)
    raw = open(p, "rb").read()
    assert raw[:4] == b"\x76\x2f\x31\x01" and b"denom\x00" in raw and b"technique\x00string\x00" in raw and b"PT" in raw
    # channels are stored alphabetically, rows top-down: first stored float of the first scan line = B of OUR top row
    with pytest.raises(ma.MiError):
        ma.load_exr(str(tmp_path / "missing.exr"))
    open(p, "r+b").truncate(len(raw) // 2)
    with pytest.raises(ma.MiError):
        ma.load_exr(p)


def write_exr_like_openexr(path, img, compression, long_key=None):
    """An EXR scan-line file written the way OpenEXR writes one — independent of master_amd/csrc/exr_io.cpp: channels B, G, R, denom (FLOAT),
    increasing-y line order, `compression` in {0 none, 1 RLE, 2 ZIPS, 3 ZIP}: per chunk the lines' channel rows are split into even / odd
    bytes, delta-predicted (d[i] = t[i] - t[i-1] + 128) and deflated (ZIP: 16 lines per chunk) or run-length coded."""
    import struct
    import zlib
    h, w = img.shape[:2]
    top_down = img[::-1]  # EXR line 0 = top; our row 0 = bottom
    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data
    ch = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", 2, 0, 0, 0, 0, 1, 1) for n in ("B", "G", "R", "denom")) + b"\0"
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    flags = 0x04 if long_key else 0
    hdr = b"\x76\x2f\x31\x01" + bytes([2, flags, 0, 0])
    hdr += attr("channels", "chlist", ch) + attr("compression", "compression", bytes([compression])) + attr("dataWindow", "box2i", box)
    hdr += attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    if long_key:
        hdr += attr(long_key, "string", b"value")
    hdr += b"\0"
    per = 16 if compression == 3 else 1
    def rle(b):
        out, i = bytearray(), 0
        while i < len(b):
            j = i
            while j + 1 < len(b) and b[j + 1] == b[i] and j - i < 126:
                j += 1
            if j - i >= 2:
                out += struct.pack("b", j - i) + b[i:i + 1]; i = j + 1
            else:
                k = i
                while k < len(b) and k - i < 127 and not (k + 2 < len(b) and b[k] == b[k + 1] == b[k + 2]):
                    k += 1
                out += struct.pack("b", -(k - i)) + b[i:k]; i = k
        return bytes(out)
    chunks = []
    for y0 in range(0, h, per):
        raw = b"".join(top_down[y, :, c].astype("<f4").tobytes() for y in range(y0, min(h, y0 + per)) for c in (2, 1, 0, 3))
        if compression == 0:
            data = raw
        else:
            t = np.frombuffer(raw, np.uint8)
            t = np.concatenate([t[0::2], t[1::2]]).astype(np.int32)
            d = t.copy(); d[1:] = (t[1:] - t[:-1] + 128 + 256) % 256
            enc = d.astype(np.uint8).tobytes()
            data = rle(enc) if compression == 1 else zlib.compress(enc)
            if len(data) >= len(raw):
                data = raw  # OpenEXR stores a chunk raw when compression does not shrink it
        chunks.append(struct.pack("<ii", y0, len(data)) + data)
    table_at = len(hdr)
    off, table = table_at + 8 * len(chunks), b""
    for c in chunks:
        table += struct.pack("<Q", off); off += len(c)
    open(path, "wb").write(hdr + table + b"".join(chunks))


@pytest.mark.parametrize("compression", [0, 1, 2, 3])
def test_exr_reader_takes_the_compressions_the_reference_writes(tmp_path, compression):
    """save_exr of the reference uses OpenEXR's default header = ZIP compression (exr.cpp:177-232): `master continue / merge / errors` inputs and
    the baked reference images come in that form.  Files from an independent encoder (smooth and noisy content, a chunk that does not shrink and is
    stored raw, a last partial ZIP block, a long attribute name) must load bit for bit."""
    rng = np.random.default_rng(compression)
    yy, xx = np.meshgrid(np.arange(37), np.arange(29), indexing="ij")
    smooth = np.stack([np.sin(xx * 0.1) + yy * 0.01, np.full_like(xx, 2.5, dtype=np.float64), xx * 0.0, np.full_like(xx, 64.0, dtype=np.float64)], -1).astype(np.float32)
    noisy = rng.uniform(0, 5, (37, 29, 4)).astype(np.float32)
    for k, img in enumerate((smooth, noisy)):
        p = str(tmp_path / ("c%d_%d.exr" % (compression, k)))
        write_exr_like_openexr(p, img, compression, long_key="an_attribute_name_longer_than_thirty_one_bytes" if k else None)
        assert np.array_equal(ma.load_exr(p).view(np.uint32), img.view(np.uint32))
    raw = open(p, "rb").read()
    for cut in (len(raw) // 3, len(raw) - 5):
        open(p, "wb").write(raw[:cut])
        with pytest.raises(ma.MiError):
            ma.load_exr(p)


def read_exr_like_openexr(path):
    """An OpenEXR scan-line reader written from the file-format specification, independent of master_amd/csrc/exr_io.cpp (the image holds no OpenEXR
    library): magic, version field, attribute list (name, type, size, value), the eight attributes every file must carry, the line-offset table,
    then one chunk per scan line (y, byte count, channels in alphabetical order).  Returns (header dict, {channel: [H][W] float32 top-down})."""
    import struct
    d = open(path, "rb").read()
    assert struct.unpack("<I", d[:4])[0] == 20000630, "magic"
    version, flags = d[4], struct.unpack("<I", d[4:8])[0] >> 8
    assert version == 2 and not flags & 0x2 and not flags & 0x10  # scan lines, single part
    max_name = 255 if flags & 0x4 else 31
    o, hdr = 8, {}
    def cstr(o):
        e = d.index(b"\0", o)
        return d[o:e].decode(), e + 1
    while d[o] != 0:
        name, o = cstr(o); typ, o = cstr(o)
        assert 1 <= len(name) <= max_name and 1 <= len(typ) <= max_name
        size = struct.unpack("<i", d[o:o + 4])[0]; o += 4
        hdr[name] = (typ, d[o:o + size]); o += size
    o += 1
    required = {"channels": "chlist", "compression": "compression", "dataWindow": "box2i", "displayWindow": "box2i", "lineOrder": "lineOrder",
                "pixelAspectRatio": "float", "screenWindowCenter": "v2f", "screenWindowWidth": "float"}
    for k, t in required.items():
        assert k in hdr and hdr[k][0] == t, k
    assert hdr["compression"][1] == b"\0" and hdr["lineOrder"][1] == b"\0"  # uncompressed, increasing y
    x0, y0, x1, y1 = struct.unpack("<4i", hdr["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    chans, c = [], hdr["channels"][1]
    p = 0
    while c[p] != 0:
        e = c.index(b"\0", p); nm = c[p:e].decode(); p = e + 1
        ptype, plinear, xs, ys = struct.unpack("<iB3xii", c[p:p + 16]); p += 16
        assert ptype == 2 and xs == 1 and ys == 1  # FLOAT, no subsampling
        chans.append(nm)
    assert chans == sorted(chans), "channels must be stored in alphabetical order"
    offsets = struct.unpack("<%dQ" % h, d[o:o + 8 * h])
    out = {nm: np.zeros((h, w), np.float32) for nm in chans}
    for y in range(h):
        off = offsets[y]
        yy, nbytes = struct.unpack("<ii", d[off:off + 8])
        assert yy == y0 + y and nbytes == w * 4 * len(chans)
        row = np.frombuffer(d[off + 8:off + 8 + nbytes], "<f4").reshape(len(chans), w)
        for k, nm in enumerate(chans):
            out[nm][y] = row[k]
    assert offsets[-1] + 8 + w * 4 * len(chans) == len(d)
    return hdr, out


def test_exr_writer_output_parses_by_the_file_format_specification(tmp_path):
    """What mi_exr_save_rgbn writes, opened by an independent reader: required attributes and their types, channel list B, G, R,
    denom as FLOAT, offset table, top-down rows, string metadata — the layout save_exr gives its files (exr.cpp:177-232)."""
    rng = np.random.default_rng(3)
    img = rng.uniform(0, 9, (21, 34, 4)).astype(np.float32)
    p = str(tmp_path / "w.exr")
    ma.save_exr(p, img, {"technique": "PT", "statistics.num_samples": "17", "records[0].frame_duration_of_a_long_key_name": "0.5"})
    hdr, ch = read_exr_like_openexr(p)
    assert sorted(ch) == ["B", "G", "R", "denom"]
    for k, nm in enumerate(("R", "G", "B", "denom")):
        assert np.array_equal(ch[nm][::-1], img[..., k])  # EXR line 0 = top of the image, our row 0 = bottom
    assert hdr["technique"] == ("string", b"PT") and hdr["statistics.num_samples"] == ("string", b"17")
    assert hdr["records[0].frame_duration_of_a_long_key_name"] == ("string", b"0.5")
    import struct
    assert struct.unpack("<4i", hdr["displayWindow"][1]) == (0, 0, 33, 20) and struct.unpack("<f", hdr["pixelAspectRatio"][1])[0] == 1.0


def test_exr_writer_sets_the_long_name_flag(tmp_path):
    img = np.ones((3, 4, 4), np.float32)
    p = str(tmp_path / "l.exr")
    key = "records[123456].frame_duration_of_the_frame"  # 43 bytes > 31
    ma.save_exr(p, img, {key: "1.0", "short": "x"})
    raw = open(p, "rb").read()
    assert raw[5] & 0x04 and key.encode() + b"\0string\0" in raw and np.array_equal(ma.load_exr(p), img)
    ma.save_exr(p, img, {"short": "x"})
    assert not open(p, "rb").read()[5] & 0x04
    with pytest.raises(ma.MiError):
        ma.save_exr(p, img, {"k" * 300: "v"})


def test_exr_vertical_flip(tmp_path):
    img = np.zeros((4, 3, 4), np.float32)
    img[0, :, 0] = 7.0  # our row 0 = bottom of the image
    p = str(tmp_path / "f.exr")
    ma.save_exr(p, img)
    raw = np.frombuffer(open(p, "rb").read(), np.uint8)
    # last scan line in the file (EXR y = 3 = bottom) must hold the 7s in its R block
    line_bytes = 8 + 3 * 16
    last = raw[-line_bytes:]
    vals = last[8:].view(np.float32).reshape(4, 3)  # B, G, R, denom
    assert np.all(vals[2] == 7.0) and not vals[0].any()


def test_host_parsers_survive_damaged_files(tmp_path):
    """The .blend / .miscene / EXR readers parse files from outside: under AddressSanitizer + UBSan (CPU build; GPU sanitizers are not
    available) every fixture and a few hundred truncated / bit-damaged copies must end in success or an error code, never in a fault."""
    import shutil
    import subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "fuzz")
    src = [os.path.join(root, "tests", "tools", "fuzz_host_parsers.cpp")] + [os.path.join(root, "master_amd", "csrc", f) for f in ("scene_host.cpp", "blend_reader.cpp", "exr_io.cpp")]
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                    "-I", os.path.join(root, "include")] + src + ["-o", exe, "-lz"], check=True)
    small = tmp_path / "in"; small.mkdir()
    rng = np.random.default_rng(5)
    for comp in (1, 2, 3):  # compressed EXR inputs from the independent encoder above: RLE, ZIPS, ZIP decoders under mutation
        write_exr_like_openexr(str(small / ("z%d.exr" % comp)), rng.uniform(0, 2, (21, 19, 4)).astype(np.float32) * (comp != 2) + 1.0, comp)
    for n in ("CornellBoxDiffuse", "TestCase0", "TestCase10", "DoubleLight", "CornellBoxSpecular"):
        shutil.copy(os.path.join(root, "scenes", n + ".miscene"), small)
    dirs = [str(small)]
    ref = "/root/reference/models"
    if os.path.isdir(ref):  # the reference's own .blend files, where they exist (not on the GPU box)
        few = tmp_path / "blend"; few.mkdir()
        for n in ("CornellBoxDiffuse", "TestCase0", "TestCase33", "CornellBoxSpecular"):
            if os.path.exists(os.path.join(ref, n + ".blend")):
                os.symlink(os.path.join(ref, n + ".blend"), few / (n + ".blend"))
        dirs.append(str(few))
    r = subprocess.run([exe, str(tmp_path), "36", "600"] + dirs, capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and "rejected" in r.stdout, (r.stdout[-400:], r.stderr[-2000:])
