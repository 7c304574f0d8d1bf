// scene_io.cpp — SURVEY.md §8f rows 1 and 2, the data formats either side of
// the trace loop: triangle meshes in (Wavefront OBJ text), resolved frames out
// (binary PPM).  Host-only; nothing here touches the GPU.
//
// No reference counterpart exists ("3D models" is a plan item,
// notes/plan.md:21); conventions followed: whole-file reads and writes like
// os_read_entire_file / os_write_entire_file_or_panic
// (code/os/filesystem.hxx:4-9), POD arrays out, colours 0xRRGGBBAA
// (code/gfx/gfx.hpp:4-6), image rows bottom-up (code/math/mat.cxx:104-129).
#include "rtgpu_internal.h"

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct V { float x, y, z; };

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r'; }

// One OBJ index token "v", "v/vt", "v//vn" or "v/vt/vn": returns the position
// index (1-based, or negative = relative to the vertices read so far) and skips
// the token.  0 = malformed.
long parse_index(const char *&p, const char *end) {
  char *q = nullptr;
  errno = 0;
  long v = std::strtol(p, &q, 10);
  if (q == p || errno) return 0;
  p = q;
  while (p < end && !is_space(*p) && *p != '\n') p++;  // the /vt/vn tail
  return v;
}

int fail(const std::string &msg) {
  rtgpu::set_global_error(msg);
  return 1;
}

}  // namespace

// text need not be NUL-terminated.  Records: `v x y z [w]` (w ignored), `f i j k
// ...` (polygons are fanned around their first vertex), `usemtl NAME` (every new
// name gets the next material slot), `#` comments; everything else (vt, vn, o,
// g, s, mtllib, l, p) is skipped.  Output: triangles only; materials[0] is a
// grey lambertian used before the first usemtl, materials[1..] one grey
// lambertian per usemtl name in order of first appearance - the caller owns the
// arrays and edits albedo / type in place.
extern "C" int rtgpu_scene_parse_obj(const char *text, uint64_t n_bytes, rtgpu_scene_desc *out) {
  if (!out) return fail("obj: output descriptor is NULL");
  std::memset(out, 0, sizeof(*out));
  if (!text && n_bytes) return fail("obj: text is NULL");
  std::vector<V> verts;
  std::vector<rtgpu_triangle> tris;
  std::vector<std::string> mtl_names;
  uint32_t cur_mat = 0;
  const char *p = text, *end = text + n_bytes;
  uint64_t line_no = 0;
  std::vector<long> face;
  std::string line;
  while (p < end) {
    const char *eol = (const char *)std::memchr(p, '\n', (size_t)(end - p));
    if (!eol) eol = end;
    line.assign(p, eol);  // NUL-terminated copy: strtof/strtol stop at the line's end
    p = eol < end ? eol + 1 : end;
    line_no++;
    const char *s = line.c_str(), *le = s + line.size();
    while (s < le && is_space(*s)) s++;
    if (s == le || *s == '#') continue;
    if (s[0] == 'v' && s + 1 < le && is_space(s[1])) {
      V v;
      char *q = nullptr;
      const char *c = s + 1;
      float xyz[3];
      for (int k = 0; k < 3; k++) {
        xyz[k] = std::strtof(c, &q);
        if (q == c) return fail("obj: line " + std::to_string(line_no) + ": vertex needs three numbers");
        c = q;
      }
      v.x = xyz[0]; v.y = xyz[1]; v.z = xyz[2];
      if (!std::isfinite(v.x) || !std::isfinite(v.y) || !std::isfinite(v.z))
        return fail("obj: line " + std::to_string(line_no) + ": non-finite vertex");
      verts.push_back(v);
    } else if (s[0] == 'f' && s + 1 < le && is_space(s[1])) {
      face.clear();
      const char *c = s + 1;
      for (;;) {
        while (c < le && is_space(*c)) c++;
        if (c >= le) break;
        long idx = parse_index(c, le);
        long n = (long)verts.size();
        long at = idx > 0 ? idx - 1 : n + idx;  // negative: relative to the end
        if (idx == 0 || at < 0 || at >= n)
          return fail("obj: line " + std::to_string(line_no) + ": vertex index out of range");
        face.push_back(at);
      }
      if (face.size() < 3) return fail("obj: line " + std::to_string(line_no) + ": face with fewer than 3 vertices");
      for (size_t k = 1; k + 1 < face.size(); k++) {
        rtgpu_triangle t;
        const V &a = verts[(size_t)face[0]], &b = verts[(size_t)face[k]], &c2 = verts[(size_t)face[k + 1]];
        t.v0[0] = a.x;  t.v0[1] = a.y;  t.v0[2] = a.z;
        t.v1[0] = b.x;  t.v1[1] = b.y;  t.v1[2] = b.z;
        t.v2[0] = c2.x; t.v2[1] = c2.y; t.v2[2] = c2.z;
        t.material = cur_mat;
        tris.push_back(t);
      }
    } else if (line.compare((size_t)(s - line.c_str()), 6, "usemtl") == 0 && s + 6 < le && is_space(s[6])) {
      const char *c = s + 6;
      while (c < le && is_space(*c)) c++;
      const char *ne = le;
      while (ne > c && is_space(ne[-1])) ne--;
      std::string name(c, ne);
      uint32_t found = 0;
      for (size_t k = 0; k < mtl_names.size(); k++)
        if (mtl_names[k] == name) found = (uint32_t)k + 1u;
      if (!found) { mtl_names.push_back(name); found = (uint32_t)mtl_names.size(); }
      cur_mat = found;
    }
  }
  if (tris.size() > RTGPU_MAX_PRIMS_PER_TYPE) return fail("obj: too many triangles");
  size_t n_mat = 1 + mtl_names.size();
  rtgpu_triangle *t_out = (rtgpu_triangle *)std::calloc(tris.size() ? tris.size() : 1, sizeof(rtgpu_triangle));
  rtgpu_material *m_out = (rtgpu_material *)std::calloc(n_mat, sizeof(rtgpu_material));
  if (!t_out || !m_out) {
    std::free(t_out);
    std::free(m_out);
    return fail("obj: out of host memory");
  }
  if (!tris.empty()) std::memcpy(t_out, tris.data(), tris.size() * sizeof(rtgpu_triangle));
  for (size_t k = 0; k < n_mat; k++) {
    m_out[k].albedo[0] = m_out[k].albedo[1] = m_out[k].albedo[2] = 0.5f;
    m_out[k].param = 0.0f;
    m_out[k].type = RTGPU_MAT_LAMBERTIAN;
  }
  out->triangles = t_out;  out->n_triangles = tris.size();
  out->materials = m_out;  out->n_materials = n_mat;
  return 0;
}

extern "C" int rtgpu_scene_load_obj(const char *path, rtgpu_scene_desc *out) {
  if (out) std::memset(out, 0, sizeof(*out));
  if (!path || !out) return fail("obj: path or output descriptor is NULL");
  std::FILE *f = std::fopen(path, "rb");
  if (!f) return fail(std::string("obj: cannot open ") + path + ": " + std::strerror(errno));
  std::string text;
  char buf[1 << 16];
  size_t got;
  while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, got);
  bool bad = std::ferror(f) != 0;
  std::fclose(f);
  if (bad) return fail(std::string("obj: read error on ") + path);
  return rtgpu_scene_parse_obj(text.data(), text.size(), out);
}

// Binary PPM (P6) of a resolved frame.  rgba8 holds 0xRRGGBBAA per pixel with
// row 0 at the BOTTOM (the layout of the accumulation buffer); PPM rows run top
// to bottom, so rows are written in reverse.  Alpha is dropped.
extern "C" int rtgpu_write_ppm(const char *path, const uint32_t *rgba8, uint32_t width, uint32_t height) {
  if (!path || !rgba8 || width == 0 || height == 0) return fail("ppm: bad arguments");
  std::FILE *f = std::fopen(path, "wb");
  if (!f) return fail(std::string("ppm: cannot open ") + path + ": " + std::strerror(errno));
  std::fprintf(f, "P6\n%u %u\n255\n", width, height);
  std::vector<unsigned char> row((size_t)width * 3);
  for (uint32_t y = height; y-- > 0;) {
    const uint32_t *src = rgba8 + (size_t)y * width;
    for (uint32_t x = 0; x < width; x++) {
      row[3 * (size_t)x + 0] = (unsigned char)(src[x] >> 24);
      row[3 * (size_t)x + 1] = (unsigned char)(src[x] >> 16);
      row[3 * (size_t)x + 2] = (unsigned char)(src[x] >> 8);
    }
    if (std::fwrite(row.data(), 1, row.size(), f) != row.size()) {
      std::fclose(f);
      return fail(std::string("ppm: write error on ") + path);
    }
  }
  if (std::fclose(f) != 0) return fail(std::string("ppm: write error on ") + path);
  return 0;
}

// PNG (8-bit RGBA, no interlace) with the image data in stored deflate blocks: no
// compression library is needed and every viewer reads it.  Rows flipped to top-down as
// in the PPM; the alpha byte is kept.
namespace {
struct Crc32 {
  uint32_t table[256];
  Crc32() {
    for (uint32_t n = 0; n < 256; n++) {
      uint32_t c = n;
      for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[n] = c;
    }
  }
  uint32_t update(uint32_t crc, const unsigned char *p, size_t n) const {
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return crc;
  }
};
void put_be32(unsigned char *p, uint32_t v) { p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v; }
bool write_chunk(std::FILE *f, const Crc32 &crc, const char type[4], const unsigned char *data, size_t n) {
  unsigned char head[8], tail[4];
  put_be32(head, (uint32_t)n);
  std::memcpy(head + 4, type, 4);
  uint32_t c = crc.update(0xFFFFFFFFu, head + 4, 4);
  c = crc.update(c, data, n);
  put_be32(tail, c ^ 0xFFFFFFFFu);
  return std::fwrite(head, 1, 8, f) == 8 && (n == 0 || std::fwrite(data, 1, n, f) == n) && std::fwrite(tail, 1, 4, f) == 4;
}
}  // namespace

extern "C" int rtgpu_write_png(const char *path, const uint32_t *rgba8, uint32_t width, uint32_t height) {
  if (!path || !rgba8 || width == 0 || height == 0) return fail("png: bad arguments");
  const size_t row_bytes = (size_t)width * 4 + 1;  // filter byte + pixels
  const size_t raw = row_bytes * height;
  if (raw >= (1ull << 31)) return fail("png: image too large for one IDAT chunk");
  // zlib stream: header, stored blocks of at most 65535 bytes, Adler-32 of the raw data
  const size_t n_blocks = (raw + 65534) / 65535;
  std::vector<unsigned char> z;
  z.reserve(2 + raw + 5 * n_blocks + 4);
  z.push_back(0x78);
  z.push_back(0x01);
  std::vector<unsigned char> data(raw);
  for (uint32_t y = 0; y < height; y++) {
    const uint32_t *src = rgba8 + (size_t)(height - 1 - y) * width;
    unsigned char *dst = &data[y * row_bytes];
    dst[0] = 0;  // filter: none
    for (uint32_t x = 0; x < width; x++) put_be32(dst + 1 + 4 * (size_t)x, src[x]);  // 0xRRGGBBAA -> R G B A
  }
  uint32_t a = 1, b = 0;
  for (size_t i = 0; i < raw;) {  // Adler-32, sums reduced every 5552 bytes
    const size_t n = std::min<size_t>(5552, raw - i);
    for (size_t k = 0; k < n; k++) { a += data[i + k]; b += a; }
    a %= 65521u; b %= 65521u;
    i += n;
  }
  for (size_t i = 0; i < raw; i += 65535) {
    const size_t n = std::min<size_t>(65535, raw - i);
    z.push_back(i + n == raw ? 1 : 0);  // BFINAL, BTYPE = 00
    z.push_back((unsigned char)(n & 0xFF)); z.push_back((unsigned char)(n >> 8));
    z.push_back((unsigned char)(~n & 0xFF)); z.push_back((unsigned char)((~n >> 8) & 0xFF));
    z.insert(z.end(), data.begin() + (std::ptrdiff_t)i, data.begin() + (std::ptrdiff_t)(i + n));
  }
  unsigned char ad[4];
  put_be32(ad, (b << 16) | a);
  z.insert(z.end(), ad, ad + 4);
  std::FILE *f = std::fopen(path, "wb");
  if (!f) return fail(std::string("png: cannot open ") + path + ": " + std::strerror(errno));
  static const Crc32 crc;
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  unsigned char ihdr[13];
  put_be32(ihdr, width);
  put_be32(ihdr + 4, height);
  ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;  // 8 bit, RGBA, deflate, adaptive, no interlace
  bool ok = std::fwrite(sig, 1, 8, f) == 8 && write_chunk(f, crc, "IHDR", ihdr, 13) &&
            write_chunk(f, crc, "IDAT", z.data(), z.size()) && write_chunk(f, crc, "IEND", nullptr, 0);
  if (std::fclose(f) != 0) ok = false;
  if (!ok) return fail(std::string("png: write error on ") + path);
  return 0;
}

