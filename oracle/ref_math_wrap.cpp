// ref_math_wrap.cpp — extern "C" shim around the REFERENCE's own math module,
// compiled from /root/reference/code where it lies (oracle/Makefile, target
// `ref`).  Test infrastructure only; output goes to oracle/_ref/ (git-ignored).
//
// Nothing of the reference is copied or replaced: first.hpp, math/math.hpp and
// math/math.cpp are the reference's files, included by path.  first.hpp refuses
// any compiler but 64-bit MSVC by macro test alone (code/first.hpp:1-11); the
// three macros below are defined AFTER the system headers so that only the
// reference's own test sees them.  The system headers are the ones
// code/first.cpp:1-6 includes.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cfloat>
#include <cmath>
#include <stdarg.h>

#define _WIN32 1
#define _WIN64 1
#define _MSC_VER 1930

#include "first.hpp"
#include "math/math.hpp"
#include "math/math.cpp"

namespace {
struct Init {
  Init() { rt::gLog_File = stderr; }  // rt::logf tees to this file (first.hpp:39-51)
} g_init;

rt::Vec3 load3(const float *p) {
  rt::Vec3 v;
  v.x = p[0]; v.y = p[1]; v.z = p[2];
  return v;
}
void store3(rt::Vec3 v, float *p) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
}  // namespace

extern "C" {
float ref_dot3(const float *a, const float *b) { return rt::dot(load3(a), load3(b)); }
float ref_len3(const float *a) { return rt::len(load3(a)); }
void ref_normalize3(const float *a, float *out) { store3(rt::normalized(load3(a)), out); }
void ref_add3(const float *a, const float *b, float *out) { store3(load3(a) + load3(b), out); }
void ref_sub3(const float *a, const float *b, float *out) { store3(load3(a) - load3(b), out); }
void ref_mul3(const float *a, const float *b, float *out) { store3(load3(a) * load3(b), out); }
void ref_scale3(const float *a, float s, float *out) { store3(load3(a) * s, out); }
void ref_transform_point4(const float *p, const float *m, float *out) {
  rt::Vec4 v;
  v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3];
  rt::Mat4x4 M;
  std::memcpy(M.v, m, sizeof(M.v));
  rt::Vec4 r = rt::transformed_point(v, M);
  out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
}
