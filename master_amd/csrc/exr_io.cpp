// exr_io.cpp — minimal OpenEXR scan-line writer / reader in the reference's channel layout.
// Replaces save_exr / load_exr (exr.cpp:177-232, 245-297), which need libIlmImf (absent):
// float channels R, G, B, denom; rows flipped so EXR line 0 is the top of the image
// (exr.cpp:207-214); run metadata as string attributes (exr.cpp:196-198).
// Written files are uncompressed (compression = NO_COMPRESSION), single part, version 2 (long-name flag when a metadata key needs it).
// The reader also takes what the reference itself writes: save_exr uses OpenEXR's default header, i.e. ZIP_COMPRESSION (16 scan lines per
// chunk, zlib + byte predictor + even/odd interleave); ZIPS (one line per chunk) and RLE come with it.  So `master continue / merge /
// errors` inputs and the reference's baked images can be loaded.
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "scene_host.hpp"

namespace {

void put_bytes(std::vector<uint8_t>& o, const void* p, size_t n) { const uint8_t* b = static_cast<const uint8_t*>(p); o.insert(o.end(), b, b + n); }
void put_str(std::vector<uint8_t>& o, const char* s) { put_bytes(o, s, std::strlen(s) + 1); }
void put_i32(std::vector<uint8_t>& o, int32_t v) { put_bytes(o, &v, 4); }
void put_f32(std::vector<uint8_t>& o, float v) { put_bytes(o, &v, 4); }
void put_attr(std::vector<uint8_t>& o, const char* name, const char* type, const void* data, int32_t size) {
  put_str(o, name); put_str(o, type); put_i32(o, size); put_bytes(o, data, size_t(size));
}

float half_to_float(uint16_t h) {
  uint32_t s = uint32_t(h >> 15) << 31, e = (h >> 10) & 31u, m = h & 1023u, bits;
  if (e == 0) {
    if (m == 0) bits = s;
    else { int sh = 0; while (!(m & 1024u)) { m <<= 1; ++sh; } m &= 1023u; bits = s | ((113u - uint32_t(sh)) << 23) | (m << 13); }
  } else if (e == 31) bits = s | 0x7F800000u | (m << 13);
  else bits = s | ((e + 112u) << 23) | (m << 13);
  float f; std::memcpy(&f, &bits, 4); return f;
}

// OpenEXR's post-processing of a compressed chunk (ImfZip / ImfRle): undo the byte predictor, then merge the two halves (even / odd bytes)
void unpredict_and_interleave(std::vector<uint8_t>& t, uint8_t* out) {
  const size_t n = t.size();
  for (size_t i = 1; i < n; ++i) t[i] = uint8_t(int(t[i - 1]) + int(t[i]) - 128);
  const size_t half = (n + 1) / 2;
  for (size_t i = 0; i < half; ++i) { out[2 * i] = t[i]; if (2 * i + 1 < n) out[2 * i + 1] = t[half + i]; }
}

// one chunk -> `expect` raw bytes.  compression: 0 none, 1 RLE, 2 ZIPS, 3 ZIP.  A chunk whose stored size equals `expect` is stored raw.
bool decode_chunk(int compression, const uint8_t* src, size_t size, size_t expect, std::vector<uint8_t>& out) {
  out.resize(expect);
  if (compression == 0 || size == expect) { if (size != expect) return false; std::memcpy(out.data(), src, expect); return true; }
  std::vector<uint8_t> tmp(expect);
  if (compression == 2 || compression == 3) {
    uLongf got = uLongf(expect);
    if (uncompress(tmp.data(), &got, src, uLong(size)) != Z_OK || got != expect) return false;
  } else if (compression == 1) {  // RLE: signed count byte: n >= 0 -> repeat next byte n + 1 times; n < 0 -> copy -n literal bytes
    size_t i = 0, o = 0;
    while (i < size) {
      const int c = int8_t(src[i++]);
      if (c < 0) { const size_t k = size_t(-c); if (i + k > size || o + k > expect) return false; std::memcpy(&tmp[o], &src[i], k); i += k; o += k; }
      else { const size_t k = size_t(c) + 1; if (i >= size || o + k > expect) return false; std::memset(&tmp[o], src[i++], k); o += k; }
    }
    if (o != expect) return false;
  } else {
    return false;
  }
  unpredict_and_interleave(tmp, out.data());
  return true;
}

}  // namespace

extern "C" {

int mi_exr_save_rgbn(const char* path, uint32_t width, uint32_t height, const float* rgbn, uint32_t n_meta,
                     const char* const* meta_keys, const char* const* meta_values) {
  if (!path || !rgbn || width == 0 || height == 0 || (n_meta && (!meta_keys || !meta_values)))
    return mi::fail(MI_ERR_INVALID_ARGUMENT, "mi_exr_save_rgbn: bad argument");
  std::vector<uint8_t> o;
  uint8_t magic[8] = {0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0};
  for (uint32_t i = 0; i < n_meta; ++i) {  // attribute names: up to 31 bytes, up to 255 with the long-name flag (version field bit 10)
    const size_t len = meta_keys[i] ? std::strlen(meta_keys[i]) : 0;
    if (len > 255) return mi::fail(MI_ERR_INVALID_ARGUMENT, "mi_exr_save_rgbn: a metadata key is longer than 255 bytes");
    if (len > 31) magic[5] |= 0x04;
  }
  put_bytes(o, magic, 8);
  {  // chlist: channels sorted by name: B, G, R, denom
    std::vector<uint8_t> ch;
    for (const char* name : {"B", "G", "R", "denom"}) {
      put_str(ch, name); put_i32(ch, 2 /*FLOAT*/);
      const uint8_t lin[4] = {0, 0, 0, 0}; put_bytes(ch, lin, 4);
      put_i32(ch, 1); put_i32(ch, 1);
    }
    ch.push_back(0);
    put_attr(o, "channels", "chlist", ch.data(), int32_t(ch.size()));
  }
  const uint8_t zero = 0;
  put_attr(o, "compression", "compression", &zero, 1);
  const int32_t win[4] = {0, 0, int32_t(width) - 1, int32_t(height) - 1};
  put_attr(o, "dataWindow", "box2i", win, 16);
  put_attr(o, "displayWindow", "box2i", win, 16);
  put_attr(o, "lineOrder", "lineOrder", &zero, 1);
  const float one = 1.0f, center[2] = {0.0f, 0.0f};
  put_attr(o, "pixelAspectRatio", "float", &one, 4);
  put_attr(o, "screenWindowCenter", "v2f", center, 8);
  put_attr(o, "screenWindowWidth", "float", &one, 4);
  for (uint32_t i = 0; i < n_meta; ++i)
    if (meta_keys[i] && meta_values[i] && meta_keys[i][0])
      put_attr(o, meta_keys[i], "string", meta_values[i], int32_t(std::strlen(meta_values[i])));
  o.push_back(0);
  const size_t table = o.size();
  const size_t line_bytes = 8 + size_t(width) * 16;
  o.resize(table + size_t(height) * 8 + size_t(height) * line_bytes);
  for (uint32_t y = 0; y < height; ++y) {
    const uint64_t off = table + size_t(height) * 8 + size_t(y) * line_bytes;
    std::memcpy(&o[table + size_t(y) * 8], &off, 8);
    const int32_t yy = int32_t(y), sz = int32_t(width * 16);
    std::memcpy(&o[off], &yy, 4); std::memcpy(&o[off + 4], &sz, 4);
    const float* row = rgbn + size_t(height - 1 - y) * width * 4;  // vertical flip
    uint8_t* dst = &o[off + 8];  // not 4-byte aligned in general: the header length is arbitrary
    const int src_of_channel[4] = {2, 1, 0, 3};  // B, G, R, denom
    for (int c = 0; c < 4; ++c)
      for (uint32_t x = 0; x < width; ++x) std::memcpy(dst + (size_t(c) * width + x) * 4, &row[size_t(x) * 4 + src_of_channel[c]], 4);
  }
  // save through a temporary + rename like Options.cpp:1271-1278 does when the output exists
  const std::string tmp = std::string(path) + ".tmp";
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return mi::fail(MI_ERR_IO, std::string("Cannot write \"") + path + "\".");
  const bool ok = std::fwrite(o.data(), 1, o.size(), f) == o.size();
  std::fclose(f);
  if (!ok || std::rename(tmp.c_str(), path) != 0) { std::remove(tmp.c_str()); return mi::fail(MI_ERR_IO, std::string("Cannot write \"") + path + "\"."); }
  return MI_OK;
}

int mi_exr_load_rgbn(const char* path, uint32_t* width, uint32_t* height, float** rgbn) {
  if (!path || !width || !height || !rgbn) return mi::fail(MI_ERR_INVALID_ARGUMENT, "mi_exr_load_rgbn: null argument");
  FILE* f = std::fopen(path, "rb");
  if (!f) return mi::fail(MI_ERR_IO, std::string("Cannot load \"") + path + "\".");
  std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> d(n > 0 ? size_t(n) : 0);
  const bool rd_ok = n > 0 && std::fread(d.data(), 1, d.size(), f) == d.size();
  std::fclose(f);
  const std::string bad = std::string("Cannot load \"") + path + "\": ";
  if (!rd_ok || d.size() < 8 || d[0] != 0x76 || d[1] != 0x2f || d[2] != 0x31 || d[3] != 0x01) return mi::fail(MI_ERR_IO, bad + "not an OpenEXR file.");
  if (d[5] & 0x12) return mi::fail(MI_ERR_UNSUPPORTED, bad + "tiled / multi-part files are not supported.");
  size_t o = 8;
  struct Chan { std::string name; int type; };
  std::vector<Chan> chans;
  int32_t win[4] = {0, 0, -1, -1};
  int compression = 0;
  auto cstr = [&](std::string& s) { size_t b = o; while (o < d.size() && d[o]) ++o; if (o >= d.size()) return false; s.assign(reinterpret_cast<char*>(&d[b]), o - b); ++o; return true; };
  for (;;) {
    if (o >= d.size()) return mi::fail(MI_ERR_IO, bad + "truncated header.");
    if (d[o] == 0) { ++o; break; }
    std::string name, type;
    int32_t size;
    if (!cstr(name) || !cstr(type) || o + 4 > d.size()) return mi::fail(MI_ERR_IO, bad + "truncated header.");
    std::memcpy(&size, &d[o], 4); o += 4;
    if (size < 0 || o + size_t(size) > d.size()) return mi::fail(MI_ERR_IO, bad + "truncated header.");
    if (name == "channels") {
      size_t p = o;
      const size_t attr_end = o + size_t(size);
      while (p < attr_end && d[p]) {
        Chan c; size_t b = p; while (p < attr_end && d[p]) ++p;
        if (p + 1 + 16 > attr_end) return mi::fail(MI_ERR_IO, bad + "truncated channel list.");
        c.name.assign(reinterpret_cast<char*>(&d[b]), p - b); ++p;
        int32_t t; std::memcpy(&t, &d[p], 4); c.type = t; p += 16;
        chans.push_back(c);
      }
    } else if (name == "dataWindow" && size == 16) std::memcpy(win, &d[o], 16);
    else if (name == "compression" && size == 1) compression = d[o];
    o += size_t(size);
  }
  if (compression < 0 || compression > 3) return mi::fail(MI_ERR_UNSUPPORTED, bad + "only uncompressed, RLE, ZIPS and ZIP scan-line files are supported.");
  const int64_t w = int64_t(win[2]) - win[0] + 1, h = int64_t(win[3]) - win[1] + 1;
  if (w <= 0 || h <= 0 || w > 65536 || h > 65536 || chans.empty()) return mi::fail(MI_ERR_IO, bad + "bad data window.");
  int dst_of[64]; size_t line_size = 0;
  if (chans.size() > 64) return mi::fail(MI_ERR_UNSUPPORTED, bad + "too many channels.");
  for (size_t c = 0; c < chans.size(); ++c) {
    const std::string& nm = chans[c].name;
    dst_of[c] = nm == "R" ? 0 : nm == "G" ? 1 : nm == "B" ? 2 : nm == "denom" ? 3 : -1;
    if (chans[c].type != 1 && chans[c].type != 2 && chans[c].type != 0) return mi::fail(MI_ERR_UNSUPPORTED, bad + "bad channel type.");
    line_size += size_t(w) * (chans[c].type == 1 ? 2 : 4);
  }
  const int64_t lines_per_chunk = compression == 3 ? 16 : 1, n_chunks = (h + lines_per_chunk - 1) / lines_per_chunk;
  if (line_size == 0 || (compression == 0 && size_t(h) > d.size() / line_size) || size_t(n_chunks) > d.size() / 8)
    return mi::fail(MI_ERR_IO, bad + "truncated pixel data.");  // also bounds the allocation by the file size (a compressed line needs >= 8 bytes of chunk header)
  if (size_t(w) * size_t(h) > (size_t(1) << 31)) return mi::fail(MI_ERR_UNSUPPORTED, bad + "image too large.");
  // compressed files: zlib / RLE expand by at most ~1032 : 1 / 128 : 1, so the pixel data cannot be larger than that multiple of the file — a
  // crafted 32 KB header must not reach a 32 GiB calloc before its first chunk is looked at
  if (compression != 0 && size_t(h) * line_size / 1100 > d.size()) return mi::fail(MI_ERR_IO, bad + "truncated pixel data.");
  float* out = static_cast<float*>(std::calloc(size_t(w) * size_t(h) * 4, sizeof(float)));
  if (!out) return mi::fail(MI_ERR_OUT_OF_MEMORY, "out of memory");
  bool has_denom = false;
  for (const Chan& c : chans) has_denom |= c.name == "denom";
  std::vector<uint8_t> raw;
  std::vector<uint8_t> seen(size_t(n_chunks), 0);
  for (int64_t k = 0; k < n_chunks; ++k) {
    uint64_t off;
    if (o + size_t(k) * 8 + 8 > d.size()) { std::free(out); return mi::fail(MI_ERR_IO, bad + "truncated offset table."); }
    std::memcpy(&off, &d[o + size_t(k) * 8], 8);
    if (off > d.size() || 8 > d.size() - off) { std::free(out); return mi::fail(MI_ERR_IO, bad + "truncated pixel data."); }
    int32_t yy, stored; std::memcpy(&yy, &d[off], 4); std::memcpy(&stored, &d[off + 4], 4);
    const int64_t first = int64_t(yy) - win[1];
    if (first < 0 || first >= h || stored < 0 || size_t(stored) > d.size() - off - 8) { std::free(out); return mi::fail(MI_ERR_IO, bad + "bad scan line."); }
    // every chunk starts on a chunk boundary and is seen exactly once: rows a malformed table left uncovered would stay 0 silently
    if (first % lines_per_chunk != 0 || seen[size_t(first / lines_per_chunk)]) { std::free(out); return mi::fail(MI_ERR_IO, bad + "misplaced or repeated scan-line chunk."); }
    seen[size_t(first / lines_per_chunk)] = 1;
    const int64_t lines = h - first < lines_per_chunk ? h - first : lines_per_chunk;
    if (!decode_chunk(compression, &d[off + 8], size_t(stored), size_t(lines) * line_size, raw)) { std::free(out); return mi::fail(MI_ERR_IO, bad + "corrupt scan-line chunk."); }
    size_t p = 0;
    for (int64_t l = 0; l < lines; ++l) {
      const int64_t row = (h - 1) - (first + l);  // flip back: EXR line 0 = top
      for (size_t c = 0; c < chans.size(); ++c) {
        for (int64_t x = 0; x < w; ++x) {
          float v;
          if (chans[c].type == 1) { uint16_t hv; std::memcpy(&hv, &raw[p], 2); p += 2; v = half_to_float(hv); }
          else if (chans[c].type == 2) { std::memcpy(&v, &raw[p], 4); p += 4; }
          else { uint32_t uv; std::memcpy(&uv, &raw[p], 4); p += 4; v = float(uv); }
          if (dst_of[c] >= 0) out[(size_t(row) * size_t(w) + size_t(x)) * 4 + size_t(dst_of[c])] = v;
        }
      }
    }
  }
  if (!has_denom)  // plain RGB image: one sample per pixel (load_exr of a 3-channel file, exr.cpp:245-297)
    for (size_t i = 0; i < size_t(w) * size_t(h); ++i) out[i * 4 + 3] = 1.0f;
  *width = uint32_t(w); *height = uint32_t(h); *rgbn = out;
  return MI_OK;
}

}  // extern "C"
