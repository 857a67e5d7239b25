// blend_reader.cpp — .blend (Blender 2.5x–2.7x, uncompressed) -> mi::SceneData.
//
// Stands in for loadScene (loader.cpp:458-487), whose importer — a fork of assimp
// (github.com/ciechowoj/assimp, pin unknown) — is not available.  What the reference asks of
// the importer and how it consumes the result is followed from loader.cpp:
//   flags Triangulate | GenNormals | JoinIdenticalVertices | PreTransformVertices  (loader.cpp:461-462)
//   cameras   loader.cpp:293-307   (fovx = 2 * aiCamera::mHorizontalFOV)
//   materials loader.cpp:371-400   ($mat.blend.transparency.use -> Transmission, $mat.blend.mirror.use ->
//                                   Reflection, specular == 0 -> Diffuse, else Phong)
//   meshes    loader.cpp:309-369   (no tangents from the importer => de-indexed vertices, per-corner frames)
//   lights    loader.cpp:434-456   (area lamps only; light quad appended as a mesh; LightBSDF / sun)
//   bounding sphere loader.cpp:408-432
// What the importer itself does with the Blender structs is NOT pinned by the reference tree;
// this reader follows the published behaviour of assimp 4.x's Blender importer (diffuse =
// Material.r/g/b, specular = specr/g/b, shininess = har, lamp colour = rgb * energy, area size
// = (area_size, area_size) for square lamps, camera half-angle = atan2(sensor_x, 2 lens)) and
// exposes the uncertain scale factors as mi_blend_options.
//
// All struct offsets are resolved through the file's own DNA1 block.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "scene_host.hpp"

namespace mi {
namespace {

struct Field { std::string type, name; size_t offset, size; bool is_ptr; };
struct Struct { std::string name; std::vector<Field> fields; size_t size; };
struct Block { char code[5]; uint32_t size; uint64_t old_ptr; uint32_t sdna, count; size_t data; };

class BlendFile {
 public:
  std::vector<uint8_t> buf;
  size_t psz = 8;
  std::vector<Block> blocks;
  std::map<uint64_t, size_t> by_ptr;
  std::vector<Struct> structs;
  std::map<std::string, size_t> struct_by_name;
  std::string error;

  bool load(const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) { error = "cannot open file"; return false; }
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    buf.resize(n > 0 ? size_t(n) : 0);
    bool ok = n > 0 && std::fread(buf.data(), 1, size_t(n), f) == size_t(n);
    std::fclose(f);
    if (!ok) { error = "cannot read file"; return false; }
    if (buf.size() < 12 || std::memcmp(buf.data(), "BLENDER", 7) != 0) {
      error = (buf.size() > 2 && buf[0] == 0x1f && buf[1] == 0x8b) ? "gzip-compressed .blend is not supported" : "not a .blend file";
      return false;
    }
    psz = buf[7] == '-' ? 8 : 4;
    if (buf[8] != 'v') { error = "big-endian .blend is not supported"; return false; }
    size_t off = 12;
    const size_t hs = psz == 8 ? 24 : 20;
    for (;;) {
      if (off + hs > buf.size()) { error = "truncated block header"; return false; }
      Block b;
      std::memcpy(b.code, &buf[off], 4); b.code[4] = 0;
      b.size = rd32(off + 4);
      if (psz == 8) { b.old_ptr = rd64(off + 8); b.sdna = rd32(off + 16); b.count = rd32(off + 20); }
      else { b.old_ptr = rd32(off + 8); b.sdna = rd32(off + 12); b.count = rd32(off + 16); }
      b.data = off + hs;
      if (std::memcmp(b.code, "ENDB", 4) == 0) break;
      if (b.data + b.size > buf.size()) { error = "truncated block"; return false; }
      if (b.old_ptr) by_ptr[b.old_ptr] = blocks.size();
      blocks.push_back(b);
      off = b.data + b.size;
    }
    return parse_dna();
  }

  // every read is bounds-checked (a damaged file must end in an error message, not in a fault): out of range reads as 0
  bool in(size_t o, size_t n) const { return o <= buf.size() && n <= buf.size() - o; }
  uint32_t rd32(size_t o) const { uint32_t v = 0; if (in(o, 4)) std::memcpy(&v, &buf[o], 4); return v; }
  uint64_t rd64(size_t o) const { uint64_t v = 0; if (in(o, 8)) std::memcpy(&v, &buf[o], 8); return v; }
  uint64_t rdptr(size_t o) const { return psz == 8 ? rd64(o) : rd32(o); }
  float rdf(size_t o) const { float v = 0.0f; if (in(o, 4)) std::memcpy(&v, &buf[o], 4); return v; }
  int16_t rd16(size_t o) const { int16_t v = 0; if (in(o, 2)) std::memcpy(&v, &buf[o], 2); return v; }

  const Field* field(const std::string& sname, const std::string& fname) const {
    auto it = struct_by_name.find(sname);
    if (it == struct_by_name.end()) return nullptr;
    for (const Field& f : structs[it->second].fields)
      if (f.name == fname) return &f;
    return nullptr;
  }
  size_t struct_size(const std::string& sname) const {
    auto it = struct_by_name.find(sname);
    return it == struct_by_name.end() ? 0 : structs[it->second].size;
  }
  // Resolve an old pointer to (block, byte offset of the pointee in buf); 0 if dangling.
  const Block* resolve(uint64_t ptr, size_t* data = nullptr) const {
    if (!ptr) return nullptr;
    auto it = by_ptr.find(ptr);
    if (it == by_ptr.end()) return nullptr;
    if (data) *data = blocks[it->second].data;
    return &blocks[it->second];
  }

 private:
  bool parse_dna() {
    const Block* dna = nullptr;
    for (const Block& b : blocks) if (std::memcmp(b.code, "DNA1", 4) == 0) dna = &b;
    if (!dna) { error = "no DNA1 block"; return false; }
    size_t o = dna->data, end = dna->data + dna->size;
    auto tag = [&](const char* t) { bool ok = o + 4 <= end && std::memcmp(&buf[o], t, 4) == 0; o += 4; return ok; };
    auto strs = [&](std::vector<std::string>& out) {
      if (o + 4 > end) { o = end; return; }
      uint32_t n = rd32(o); o += 4;
      for (uint32_t i = 0; i < n && o < end; ++i) {
        const char* s = reinterpret_cast<const char*>(&buf[o]);
        size_t len = strnlen(s, end - o);
        out.emplace_back(s, len);
        o += len + 1;
      }
      o = (o + 3) & ~size_t(3);
    };
    std::vector<std::string> names, types;
    if (!tag("SDNA") || !tag("NAME")) { error = "bad DNA1"; return false; }
    strs(names);
    if (!tag("TYPE")) { error = "bad DNA1 (TYPE)"; return false; }
    strs(types);
    if (!tag("TLEN")) { error = "bad DNA1 (TLEN)"; return false; }
    std::vector<uint16_t> tlen(types.size());
    if (o > end || 2 * types.size() > end - o) { error = "bad DNA1 (TLEN)"; return false; }
    std::memcpy(tlen.data(), &buf[o], 2 * types.size());
    o += 2 * types.size(); o = (o + 3) & ~size_t(3);
    if (!tag("STRC")) { error = "bad DNA1 (STRC)"; return false; }
    if (o + 4 > end) { error = "bad DNA1 (STRC)"; return false; }
    uint32_t ns = rd32(o); o += 4;
    for (uint32_t i = 0; i < ns; ++i) {
      if (o + 4 > end) { error = "bad DNA1 (STRC)"; return false; }
      uint16_t t = uint16_t(rd16(o)), nf = uint16_t(rd16(o + 2)); o += 4;
      if (t >= types.size() || o > end || size_t(nf) * 4 > end - o) { error = "bad DNA1 (STRC)"; return false; }
      Struct st; st.name = types[t]; st.size = tlen[t];
      size_t off = 0;
      for (uint16_t k = 0; k < nf; ++k) {
        uint16_t ft = uint16_t(rd16(o)), fn = uint16_t(rd16(o + 2)); o += 4;
        if (ft >= types.size() || fn >= names.size() || names[fn].empty()) { error = "bad DNA1 (STRC)"; return false; }
        const std::string& full = names[fn];
        Field f; f.type = types[ft]; f.offset = off;
        f.is_ptr = full[0] == '*' || full[0] == '(';
        size_t count = 1;
        std::string bare = full;
        for (size_t p = bare.find('['); p != std::string::npos; p = bare.find('[', p + 1)) {
          const int dim = std::atoi(bare.c_str() + p + 1);
          count *= size_t(dim > 0 && dim < (1 << 20) ? dim : 1);
          if (count > (size_t(1) << 28)) count = size_t(1) << 28;
        }
        if (bare.find('[') != std::string::npos) bare = bare.substr(0, bare.find('['));
        if (!bare.empty() && bare[0] == '(') { const size_t q = bare.find(')'); bare = (bare.size() > 2 && q != std::string::npos && q > 2) ? bare.substr(2, q - 2) : std::string(); }
        while (!bare.empty() && bare[0] == '*') bare = bare.substr(1);
        f.name = bare;
        f.size = (f.is_ptr ? psz : tlen[ft]) * count;
        off += f.size;
        st.fields.push_back(f);
      }
      struct_by_name[st.name] = structs.size();
      structs.push_back(st);
    }
    return true;
  }
};

struct F3 { float x, y, z; };
inline F3 operator+(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline F3 operator-(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline F3 operator*(F3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline F3 cross(F3 a, F3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline F3 normalize(F3 a) { float s = 1.0f / std::sqrt(dot(a, a)); return a * s; }

struct Mat4 {  // column-major, Blender obmat[4][4]: m[c][r]
  float m[4][4];
  F3 point(F3 p) const { return {m[0][0] * p.x + m[1][0] * p.y + m[2][0] * p.z + m[3][0], m[0][1] * p.x + m[1][1] * p.y + m[2][1] * p.z + m[3][1], m[0][2] * p.x + m[1][2] * p.y + m[2][2] * p.z + m[3][2]}; }
  F3 dir(F3 p) const { return {m[0][0] * p.x + m[1][0] * p.y + m[2][0] * p.z, m[0][1] * p.x + m[1][1] * p.y + m[2][1] * p.z, m[0][2] * p.x + m[1][2] * p.y + m[2][2] * p.z}; }
  // inverse-transpose of the upper 3x3 applied to a normal (what PreTransformVertices does)
  F3 normal(F3 n) const {
    float a = m[0][0], b = m[0][1], c = m[0][2], d = m[1][0], e = m[1][1], f = m[1][2], g = m[2][0], h = m[2][1], i = m[2][2];
    // cofactor matrix (= det * inverse-transpose), scale is irrelevant after normalisation
    F3 r = {(e * i - h * f) * n.x + -(b * i - h * c) * n.y + (b * f - e * c) * n.z,
            -(d * i - g * f) * n.x + (a * i - g * c) * n.y + -(a * f - d * c) * n.z,
            (d * h - g * e) * n.x + -(a * h - g * b) * n.y + (a * e - d * b) * n.z};
    float det = a * (e * i - h * f) - d * (b * i - h * c) + g * (b * f - e * c);
    if (det < 0) r = r * -1.0f;
    return r;
  }
};

struct MeshTri { F3 p[3]; F3 n[3]; };

constexpr int OB_MESH = 1, OB_LAMP = 10, OB_CAMERA = 11;
constexpr int LA_AREA = 4;
constexpr int MA_TRANSP = 0x10000, MA_RAYMIRROR = 0x40000;
constexpr int LA_NO_DIFF = 0x800;

std::string id_name(const BlendFile& bf, size_t base) {
  const Field* f = bf.field("ID", "name");
  if (!f) return "";
  if (!bf.in(base + f->offset, f->size)) return "";
  const char* s = reinterpret_cast<const char*>(&bf.buf[base + f->offset]);
  std::string n(s, strnlen(s, f->size));
  return n.size() > 2 ? n.substr(2) : n;
}

}  // namespace

int load_blend(const char* path, const mi_blend_options* opts_in, SceneData& out) {
  mi_blend_options opts = {0.0f, 0.0f, 1.0f, 0};
  if (opts_in) opts = *opts_in;
  if (!(opts.lamp_energy_scale > 0.0f)) opts.lamp_energy_scale = 1.0f;

  BlendFile bf;
  if (!bf.load(path)) return fail(MI_ERR_IO, std::string("Cannot load \"") + path + "\" scene: " + bf.error + ".");

#define NEED(s, f) const Field* s##_##f = bf.field(#s, #f); if (!s##_##f) return fail(MI_ERR_IO, std::string("Cannot load \"") + path + "\" scene: DNA has no " #s "." #f ".")
  NEED(Object, type); NEED(Object, data); NEED(Object, obmat);
  NEED(Mesh, mat); NEED(Mesh, mpoly); NEED(Mesh, mloop); NEED(Mesh, mvert); NEED(Mesh, totvert); NEED(Mesh, totpoly);
  NEED(Mesh, totloop); NEED(Mesh, totcol);
  NEED(MVert, co); NEED(MVert, no); NEED(MLoop, v); NEED(MPoly, loopstart); NEED(MPoly, totloop); NEED(MPoly, mat_nr);
  NEED(Material, r); NEED(Material, g); NEED(Material, b); NEED(Material, specr); NEED(Material, specg); NEED(Material, specb);
  NEED(Material, ref); NEED(Material, spec); NEED(Material, har); NEED(Material, mode); NEED(Material, ang);
  NEED(Lamp, type); NEED(Lamp, r); NEED(Lamp, g); NEED(Lamp, b); NEED(Lamp, energy); NEED(Lamp, area_shape);
  NEED(Lamp, area_size); NEED(Lamp, area_sizey); NEED(Lamp, mode);
  NEED(Camera, lens); NEED(Camera, sensor_x); NEED(Camera, clipsta); NEED(Camera, clipend);
#undef NEED
  const size_t sz_mvert = bf.struct_size("MVert"), sz_mloop = bf.struct_size("MLoop"), sz_mpoly = bf.struct_size("MPoly");

  // Objects in scene order: Scene.base list when resolvable, else file order.
  std::vector<size_t> objects;  // byte offsets of Object structs
  {
    const Field* sc_base = bf.field("Scene", "base");
    const Field* base_next = bf.field("Base", "next");
    const Field* base_obj = bf.field("Base", "object");
    for (const Block& b : bf.blocks) {
      if (std::memcmp(b.code, "SC\0\0", 4) != 0 || !sc_base || !base_next || !base_obj) continue;
      uint64_t p = bf.rdptr(b.data + sc_base->offset);  // ListBase.first
      size_t guard = 0;
      while (p && guard++ < 1000000) {
        size_t bd;
        if (!bf.resolve(p, &bd)) break;
        size_t od;
        if (bf.resolve(bf.rdptr(bd + base_obj->offset), &od)) objects.push_back(od);
        p = bf.rdptr(bd + base_next->offset);
      }
      break;
    }
    if (objects.empty())
      for (const Block& b : bf.blocks)
        if (std::memcmp(b.code, "OB\0\0", 4) == 0) objects.push_back(b.data);
  }

  auto obmat_of = [&](size_t ob) {
    Mat4 m;
    std::memcpy(m.m, &bf.buf[ob + Object_obmat->offset], sizeof m.m);
    return m;
  };

  // ---- cameras (loader.cpp:293-307) --------------------------------------------------------
  out = SceneData();
  for (size_t ob : objects) {
    if (bf.rd16(ob + Object_type->offset) != OB_CAMERA) continue;
    size_t cd;
    if (!bf.resolve(bf.rdptr(ob + Object_data->offset), &cd)) continue;
    Mat4 M = obmat_of(ob);
    float lens = bf.rdf(cd + Camera_lens->offset), sensor = bf.rdf(cd + Camera_sensor_x->offset);
    mi_camera c;
    F3 pos = M.point({0, 0, 0});
    F3 look = normalize(M.dir({0, 0, -1}));  // aiCamera::mLookAt, normalised at loader.cpp:299
    F3 up = normalize(M.dir({0, 1, 0}));     // aiCamera::mUp, loader.cpp:300
    c.position[0] = pos.x; c.position[1] = pos.y; c.position[2] = pos.z;
    c.direction[0] = look.x; c.direction[1] = look.y; c.direction[2] = look.z;
    c.up[0] = up.x; c.up[1] = up.y; c.up[2] = up.z;
    float half = std::atan2(sensor, 2.0f * lens);  // importer's mHorizontalFOV (half angle)
    c.fovx = half * 2.0f;                          // loader.cpp:301
    out.cameras.push_back(c);
    mi_material m; std::memset(&m, 0, sizeof m);
    m.type = MI_BSDF_CAMERA;                       // loader.cpp:304-305
    out.materials.push_back(m);
    out.material_names.push_back("camera");
  }
  const uint32_t materials_base = uint32_t(out.materials.size());  // loader.cpp:373

  // ---- meshes: collect triangles per Blender material (PreTransformVertices joins per material)
  std::vector<uint64_t> material_ptrs;             // importer material order = first use
  std::vector<std::vector<MeshTri>> tris_by_material;
  auto material_slot = [&](uint64_t ptr) {
    for (size_t i = 0; i < material_ptrs.size(); ++i) if (material_ptrs[i] == ptr) return i;
    material_ptrs.push_back(ptr);
    tris_by_material.emplace_back();
    return material_ptrs.size() - 1;
  };
  for (size_t ob : objects) {
    if (bf.rd16(ob + Object_type->offset) != OB_MESH) continue;
    size_t me;
    if (!bf.resolve(bf.rdptr(ob + Object_data->offset), &me)) continue;
    Mat4 M = obmat_of(ob);
    int totvert = int(bf.rd32(me + Mesh_totvert->offset)), totpoly = int(bf.rd32(me + Mesh_totpoly->offset));
    int totloop = int(bf.rd32(me + Mesh_totloop->offset)), totcol = bf.rd16(me + Mesh_totcol->offset);
    size_t mv, ml, mp, mm = 0;
    const Block* bv = bf.resolve(bf.rdptr(me + Mesh_mvert->offset), &mv);
    const Block* bl = bf.resolve(bf.rdptr(me + Mesh_mloop->offset), &ml);
    const Block* bp = bf.resolve(bf.rdptr(me + Mesh_mpoly->offset), &mp);
    if (!bv || !bl || !bp || totpoly <= 0) continue;
    if (size_t(totvert) * sz_mvert > bv->size || size_t(totloop) * sz_mloop > bl->size || size_t(totpoly) * sz_mpoly > bp->size)
      return fail(MI_ERR_IO, std::string("Cannot load \"") + path + "\" scene: mesh arrays are truncated.");
    const Block* bm = bf.resolve(bf.rdptr(me + Mesh_mat->offset), &mm);
    for (int p = 0; p < totpoly; ++p) {
      size_t po = mp + size_t(p) * sz_mpoly;
      int ls = int(bf.rd32(po + MPoly_loopstart->offset)), tl = int(bf.rd32(po + MPoly_totloop->offset));
      int mat_nr = bf.rd16(po + MPoly_mat_nr->offset);
      if (tl < 3 || ls < 0 || ls + tl > totloop) continue;
      uint64_t mptr = 0;
      if (bm && mat_nr >= 0 && mat_nr < totcol && size_t(mat_nr + 1) * bf.psz <= bm->size) mptr = bf.rdptr(mm + size_t(mat_nr) * bf.psz);
      std::vector<MeshTri>& dst = tris_by_material[material_slot(mptr)];
      auto corner = [&](int k, F3& P, F3& N) {
        int v = int(bf.rd32(ml + size_t(ls + k) * sz_mloop + MLoop_v->offset));
        if (v < 0 || v >= totvert) v = 0;
        size_t vo = mv + size_t(v) * sz_mvert;
        F3 co = {bf.rdf(vo + MVert_co->offset), bf.rdf(vo + MVert_co->offset + 4), bf.rdf(vo + MVert_co->offset + 8)};
        F3 no = {bf.rd16(vo + MVert_no->offset) / 32767.0f, bf.rd16(vo + MVert_no->offset + 2) / 32767.0f, bf.rd16(vo + MVert_no->offset + 4) / 32767.0f};
        P = M.point(co);
        N = normalize(M.normal(no));
      };
      for (int k = 1; k + 1 < tl; ++k) {  // aiProcess_Triangulate: (0, k, k+1)
        MeshTri t;
        corner(0, t.p[0], t.n[0]); corner(k, t.p[1], t.n[1]); corner(k + 1, t.p[2], t.n[2]);
        dst.push_back(t);
      }
    }
  }

  // ---- materials (loader.cpp:375-400) ------------------------------------------------------
  for (size_t i = 0; i < material_ptrs.size(); ++i) {
    mi_material m; std::memset(&m, 0, sizeof m);
    std::string name = "DefaultMaterial";
    size_t md;
    if (bf.resolve(material_ptrs[i], &md)) {
      name = id_name(bf, md);
      float ref = bf.rdf(md + Material_ref->offset), spec = bf.rdf(md + Material_spec->offset);
      float ds = opts.diffuse_scale_by_ref != 0.0f ? ref : 1.0f, ss = opts.specular_scale_by_spec != 0.0f ? spec : 1.0f;
      m.diffuse[0] = bf.rdf(md + Material_r->offset) * ds; m.diffuse[1] = bf.rdf(md + Material_g->offset) * ds; m.diffuse[2] = bf.rdf(md + Material_b->offset) * ds;
      m.specular[0] = bf.rdf(md + Material_specr->offset) * ss; m.specular[1] = bf.rdf(md + Material_specg->offset) * ss; m.specular[2] = bf.rdf(md + Material_specb->offset) * ss;
      m.power = float(bf.rd16(md + Material_har->offset));
      int mode = int(bf.rd32(md + Material_mode->offset));
      if (mode & MA_TRANSP) {               // "$mat.blend.transparency.use"
        m.type = MI_BSDF_TRANSMISSION;
        m.ior_internal = bf.rdf(md + Material_ang->offset);  // "$mat.blend.transparency.ior"
        m.ior_external = 1.0f;              // loader.cpp:382
      } else if (mode & MA_RAYMIRROR) {     // "$mat.blend.mirror.use"
        m.type = MI_BSDF_REFLECTION;
      } else if (m.specular[0] == 0.0f && m.specular[1] == 0.0f && m.specular[2] == 0.0f) {
        m.type = MI_BSDF_DIFFUSE;           // loader.cpp:386-388
      } else {
        m.type = MI_BSDF_PHONG;             // loader.cpp:397-398
      }
    } else {
      m.type = MI_BSDF_DIFFUSE;             // assimp's default material: grey 0.6
      m.diffuse[0] = m.diffuse[1] = m.diffuse[2] = 0.6f;
    }
    out.materials.push_back(m);
    out.material_names.push_back(name);
  }

  // ---- meshes (loader.cpp:309-369, "no tangents" branch) -----------------------------------
  out.mesh_tri_offset.push_back(0);
  auto push_frame = [&](F3 c0, F3 c1, F3 c2) {
    const float t[9] = {c0.x, c0.y, c0.z, c1.x, c1.y, c1.z, c2.x, c2.y, c2.z};
    out.tangents.insert(out.tangents.end(), t, t + 9);
  };
  for (size_t i = 0; i < tris_by_material.size(); ++i) {
    if (tris_by_material[i].empty()) continue;
    for (const MeshTri& t : tris_by_material[i]) {
      uint32_t base = uint32_t(out.positions.size() / 3);
      F3 edge = t.p[1] - t.p[0];                                     // loader.cpp:332
      for (int k = 0; k < 3; ++k) {
        out.positions.push_back(t.p[k].x); out.positions.push_back(t.p[k].y); out.positions.push_back(t.p[k].z);
        F3 normal = t.n[k];
        F3 tangent = normalize(edge - normal * dot(normal, edge));   // loader.cpp:336
        F3 bitangent = normalize(cross(normal, tangent));            // loader.cpp:337
        push_frame(tangent, normal, bitangent);                      // [0]=tangent, [1]=normal, [2]=bitangent
        out.indices.push_back(base + uint32_t(k));
      }
    }
    out.mesh_tri_offset.push_back(uint32_t(out.indices.size() / 3));
    out.mesh_material_id.push_back(((materials_base + uint32_t(i)) << 2) | MI_ENTITY_MESH);  // loader.cpp:365-366
    out.mesh_names.push_back(out.material_names[materials_base + i]);
  }
  if (out.indices.empty()) return fail(MI_ERR_IO, std::string("Cannot load \"") + path + "\" scene: no mesh geometry.");

  // ---- bounding sphere over the surface meshes (loader.cpp:408-432) ------------------------
  {
    size_t nv = out.positions.size() / 3;
    F3 c = {0, 0, 0};
    for (size_t v = 0; v < nv; ++v) c = c + F3{out.positions[3 * v], out.positions[3 * v + 1], out.positions[3 * v + 2]};
    c = c * (1.0f / float(nv));
    float r2 = 0.0f;
    for (size_t v = 0; v < nv; ++v) {
      F3 d = c - F3{out.positions[3 * v], out.positions[3 * v + 1], out.positions[3 * v + 2]};
      r2 = std::fmax(r2, dot(d, d));
    }
    out.bounding_sphere[0] = c.x; out.bounding_sphere[1] = c.y; out.bounding_sphere[2] = c.z;
    out.bounding_sphere[3] = std::sqrt(r2);
  }

  // ---- lights (loader.cpp:434-456, AreaLights.cpp:38-97) -----------------------------------
  for (size_t ob : objects) {
    if (bf.rd16(ob + Object_type->offset) != OB_LAMP) continue;
    size_t la;
    if (!bf.resolve(bf.rdptr(ob + Object_data->offset), &la)) continue;
    if (bf.rd16(la + Lamp_type->offset) != LA_AREA) continue;       // loader.cpp:439
    Mat4 M = obmat_of(ob);
    float energy = bf.rdf(la + Lamp_energy->offset) * opts.lamp_energy_scale;
    float sx = bf.rdf(la + Lamp_area_size->offset);
    float sy = bf.rd16(la + Lamp_area_shape->offset) == 0 ? sx : bf.rdf(la + Lamp_area_sizey->offset);
    F3 position = M.point({0, 0, 0});
    F3 direction = normalize(M.dir({0, 0, -1}));  // Blender area lamps emit along local -Z
    F3 up = normalize(M.dir({0, 1, 0}));
    const uint32_t material_index = uint32_t(out.materials.size());  // loader.cpp:443
    const uint32_t light_id = uint32_t(out.lights.size());
    mi_light l; std::memset(&l, 0, sizeof l);
    F3 t0 = normalize(cross(up, direction));      // AreaLights.cpp:80
    const float T[9] = {t0.x, t0.y, t0.z, direction.x, direction.y, direction.z, up.x, up.y, up.z};
    std::memcpy(l.tangent, T, sizeof T);
    l.position[0] = position.x; l.position[1] = position.y; l.position[2] = position.z;
    l.size[0] = sx; l.size[1] = sy;
    l.exitance[0] = bf.rdf(la + Lamp_r->offset) * energy; l.exitance[1] = bf.rdf(la + Lamp_g->offset) * energy; l.exitance[2] = bf.rdf(la + Lamp_b->offset) * energy;
    l.diffuse = (int(bf.rd32(la + Lamp_mode->offset)) & LA_NO_DIFF) ? 0u : 1u;  // aiLight::mDiffuse is fork-only; unpinned
    l.material_id = (material_index << 2) | MI_ENTITY_LIGHT;       // AreaLights.cpp:86
    out.lights.push_back(l);
    // AreaLight::create_mesh (AreaLights.cpp:38-60)
    F3 left = t0 * 0.5f, upv = up * 0.5f;
    F3 q[4] = {position - left * sx - upv * sy, position + left * sx - upv * sy, position + left * sx + upv * sy, position - left * sx + upv * sy};
    uint32_t base = uint32_t(out.positions.size() / 3);
    for (int k = 0; k < 4; ++k) {
      out.positions.push_back(q[k].x); out.positions.push_back(q[k].y); out.positions.push_back(q[k].z);
      push_frame(t0, direction, up);
    }
    const uint32_t idx[6] = {0, 1, 2, 2, 3, 0};
    for (uint32_t k : idx) out.indices.push_back(base + k);
    out.mesh_tri_offset.push_back(uint32_t(out.indices.size() / 3));
    out.mesh_material_id.push_back(l.material_id);
    std::string name = id_name(bf, ob);
    out.mesh_names.push_back(name);
    mi_material m; std::memset(&m, 0, sizeof m);
    m.type = l.diffuse ? MI_BSDF_LIGHT : MI_BSDF_SUN;              // AreaLights.cpp:26-36
    m.light_id = light_id;
    out.materials.push_back(m);
    out.material_names.push_back(name);
  }
  return MI_OK;
}

}  // namespace mi
