// isx_device.hpp — gfx950 device code of the integrating-sphere tracer (product code).
//
// Numeric contract: every value that can reach a branch or an output is
// computed with IEEE binary64 +,-,*,/,sqrt and EXPLICIT fma() in exactly the expression
// order of the specification; the translation unit is built with -ffp-contract=off so
// hipcc adds no fused operation of its own.  Anything marked "cull" is conservative
// pre-selection and never decides a result.
//
// Reference behaviour replaced (see include/isx.h for the full list):
//   AOpticsManager::TraceNonSequential  fluxAtObserverOptimize.C:254,295 (ROBAST, behaviour inferred)
//   port test                            fluxAtObserver.C:162-166
//   Detector::checkIntersection          fluxAtObserver.C:70-107
//   BRDF                                 nonLambertianFlux.C:147-208
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// gfx950 (MI355X, CDNA4) is the ONLY target: the device code uses v_bitop3_b32 (xor3 below), CDNA4's 160 KiB of LDS per
// workgroup and its packed-f32 rates.  Another --offload-arch is refused here, at the top, instead of failing somewhere in Philox.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libisx is written for gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif

namespace isx {

struct V3 { double x, y, z; };

// GLOBAL-typed views of device buffers.  A pointer that reaches a kernel through the LDS copy of a parameter block (or through
// a compiler barrier) is a generic pointer to hipcc, and an access through it is a flat_load / flat_atomic: flat operations
// complete out of order, so the wave waits with s_waitcnt vmcnt(0) lgkmcnt(0) -- a detector-table read then also drains the
// wave's LDS queue.  Through these types the same access is a global_load / global_atomic (tools/isa_stats.py --check).
typedef double isx_d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const isx_d2 GlbD2;
typedef __attribute__((address_space(1))) const double GlbF64;
typedef __attribute__((address_space(1))) unsigned long long GlbU64;
// six consecutive doubles (an exit line: last point + direction; a detector or disc entry) from global memory
__device__ __forceinline__ void load6(const double* p6, double (&o)[6]) {
  const GlbF64* s = (const GlbF64*)p6;
  o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3]; o[4] = s[4]; o[5] = s[5];
}
__device__ __forceinline__ void load_line(const double* p6, V3& P, V3& V) {   // (16-byte aligned: three 16-byte loads)
  const GlbD2* s = (const GlbD2*)p6;
  const isx_d2 a = s[0], b = s[1], c = s[2];
  P.x = a.x; P.y = a.y; P.z = b.x; V.x = b.y; V.y = c.x; V.z = c.y;
}
// += on a 64-bit global accumulator (device scope, relaxed: the histogram flush and the census)
__device__ __forceinline__ void global_add_u64(unsigned long long* p, unsigned long long v) {
  __hip_atomic_fetch_add((GlbU64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

enum : int { K_NONE = 0, K_INNER = 1, K_OUTER = 2, K_CONE = 3, K_BOX = 4 };
enum : int { ST_EXITED = 1, ST_ABSORBED = 2, ST_SUSPENDED = 3 };

// Kernel-argument block: prepared geometry (host computes it with the spec's expressions).
struct Geom {
  double rin2, rout2, zcut_in, zcut_out, k2, ninv_rin, inv_rout, H, rho, sigma;
  double src[3], dir0[3];
  double brdf_theta_scale;  // rough * M_PI / 6   (nonLambertianFlux.C:178)
  double brdf_spec;         // specular/(specular+diffuse) (:157-159)
  int lambertian, limit, source_model, surface_model;
  int sched_mask, sched_min;
  int chord, pad2;          // ISX_TRACE_CHORD; sched_*: generic-search batching, flush when (iter & mask) == mask or >= min lanes parked
  double r_in;
  unsigned long long rho_thr;  // absorb test on the raw Philox word: (w + 0.5) 2^-32 < rho  <=>  w < rho_thr (exact, see prepare_geom)
  double inv_thr;              // 1 / rho_thr: the surviving word b, rescaled, is the azimuth's uniform u2 = (b + 1/2) / rho_thr
  double psi_k1, psi_k0;       // circle_point_psi's angle straight from the word: psi = (u2 - 1/2) pi/2 = fma(b, psi_k1, psi_k0)
  // The pencil source starts every ray at the same point in the same direction, so the first boundary is one point for the
  // whole launch: the persistent kernels find it once per workgroup (next_hit_s1<true> on the source itself, the arithmetic
  // every ray would repeat) and keep it in their LDS copy of this block.  q0_ok = 0: rule S1 does not apply to the source
  // (it lies outside the ball, or its first hit falls into the port opening): fresh rays then take the generic search.
  double q0[3];
  int q0_ok, pad3;
};

// The handful of constants the hot loop needs; kept in SGPRs.  Everything else of Geom is read
// on demand from an LDS copy (a `const volatile Geom&`), so it never occupies scalar registers
// across the loop (SGPR spills were >10 % of the issued instructions before this split).
struct Hot {
  double zcut_in, ninv_rin, r_in, psi_k1, psi_k0;
  unsigned long long rho_thr;
  int lambertian, limit, source_model, surface_model, chord;
};
__device__ __forceinline__ Hot make_hot(const Geom& g) {
  Hot h;
  h.zcut_in = g.zcut_in; h.ninv_rin = g.ninv_rin; h.rho_thr = g.rho_thr; h.psi_k1 = g.psi_k1; h.psi_k0 = g.psi_k0;
  h.lambertian = g.lambertian; h.limit = g.limit; h.source_model = g.source_model; h.surface_model = g.surface_model;
  h.r_in = g.r_in; h.chord = g.chord;
  return h;
}

__device__ __forceinline__ double dot3(const V3& a, const V3& b) { return fma(a.x, b.x, fma(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ V3 axpy(double t, const V3& v, const V3& p) {
  V3 q;
  q.x = fma(t, v.x, p.x); q.y = fma(t, v.y, p.y); q.z = fma(t, v.z, p.z);
  return q;
}

// ---------------------------------------------------------------- Philox4x32-10
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return (uint32_t)__builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t w[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    // (the two three-way XORs of a round as ONE v_bitop3_b32 each -- truth table 0x96 = a ^ b ^ c; hipcc leaves them as two v_xor_b32
    //  when the key is a scalar: 40 -> 20 instructions per block, 13 % -> 7 % of the trace kernel's VALU stream)
    const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  w[0] = c0; w[1] = c1; w[2] = c2; w[3] = c3;
}
__device__ __forceinline__ void draw_block(uint64_t seed, uint64_t ray, uint32_t block, uint32_t stream, uint32_t w[4]) {
  philox4x32_10((uint32_t)ray, (uint32_t)(ray >> 32), block, stream, (uint32_t)seed, (uint32_t)(seed >> 32), w);
}
// The two random words (a, b) of interaction j of a trace: block j/2 of the trace's stream serves two interactions,
// words (0,1) the even one, (2,3) the odd one.  a -> polar angle; b -> absorption (survive iff b < rho_thr) and, rescaled,
// the azimuth.  PH_DIRECT computes the block on the spot.  The persistent kernels run an even number of synchronous
// bounce steps per loop trip and keep the last block in `cw`: on even steps (PH_EVEN) every lane computes the block its
// NEXT even interaction needs, (j+1)/2 -- its current one if j is even, one ahead if j is odd, in which case words (2,3)
// of the block it already holds serve this bounce; on odd steps (PH_ODD) nothing is computed.  A lane reaches an odd
// step only from the even step before it, and re-enters (refill, generic-search flush, re-scatter) only on step 0, so
// `cw` always holds block j/2 when it is read.  One Philox block per two bounces; which words a ray sees never depends
// on the schedule.
enum : int { PH_DIRECT = 0, PH_EVEN = 1, PH_ODD = 2 };
template <int PH>
__device__ __forceinline__ void bounce_words(uint64_t seed, uint64_t ray, uint32_t j, uint32_t stream, uint32_t (&cw)[4],
                                             uint32_t& wa, uint32_t& wb) {
  const bool odd = (j & 1u) != 0u;
  if (PH == PH_ODD) {
    wa = odd ? cw[2] : cw[0];
    wb = odd ? cw[3] : cw[1];
  } else if (PH == PH_EVEN) {
    const uint32_t ha = cw[2], hb = cw[3];
    draw_block(seed, ray, (j + 1u) >> 1, stream, cw);
    wa = odd ? ha : cw[0];
    wb = odd ? hb : cw[1];
  } else {
    uint32_t w[4];
    draw_block(seed, ray, j >> 1, stream, w);
    wa = odd ? w[2] : w[0];
    wb = odd ? w[3] : w[1];
  }
}
// (w + 0.5) * 2^-32: both operations are exact (33 significant bits), and both constants are inline operands
__device__ __forceinline__ double u01(uint32_t w) { return ((double)w + 0.5) * 0x1.0p-32; }

// ---------------------------------------------------------------- elementary functions
// IEEE sqrt and 1/x for operands KNOWN to be normal and far from the exponent limits (1 - z^2 in [2^-31, 1]; the squared
// length of an emitted direction n + s, in [~1e-32, 4]).  The arithmetic is the compiler's own f64 expansion (v_rsq/v_rcp seed, Goldschmidt/Newton
// steps in fma, final residual correction) without the range scaling and special-case fix-ups those operands
// never need, so the results are the correctly rounded ones -- checked against the CPU over the whole input
// family in tests/test_gpu_parity.py::test_unit_range_sqrt_rcp_are_ieee.
__device__ __forceinline__ double sqrt_unit(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  const double d0 = fma(-g, g, x);
  g = fma(d0, h, g);
  const double d1 = fma(-g, g, x);
  return fma(d1, h, g);
}
__device__ __forceinline__ double neg_rcp_unit(double d) {   // -1.0 / d
  double y = __builtin_amdgcn_rcp(d);
  y = fma(fma(-d, y, 1.0), y, y);
  y = fma(fma(-d, y, 1.0), y, y);
  const double q = -y;                       // -1 * y (exact)
  const double r = fma(-d, q, -1.0);         // residual of q against the numerator -1
  return fma(r, y, q);
}
// A 64-bit literal cannot be an inline VALU operand.  Left alone, hipcc hoists the polynomial
// coefficients into VGPRs outside the trace loop, runs out of registers and reloads them from
// SCRATCH inside the loop (10 dependent scratch loads per bounce).  sconst() pins a literal to an
// SGPR pair at its point of use (2 s_mov per constant, SALU is otherwise idle); the value is
// unchanged, so the arithmetic is exactly the specified one.
__device__ __forceinline__ double sconst(double x) {
  asm volatile("" : "+s"(x));
  return x;
}
__device__ __forceinline__ double kern_sin(double x) {
  const double S1 = sconst(-1.66666666666666324348e-01), S2 = sconst(8.33333333332248946124e-03),
               S3 = sconst(-1.98412698298579493134e-04), S4 = sconst(2.75573137070700676789e-06),
               S5 = sconst(-2.50507602534068634195e-08), S6 = sconst(1.58969099521155010221e-10);
  const double z = x * x;
  const double r = fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2);
  const double v = z * x;
  return fma(v, fma(z, r, S1), x);
}
__device__ __forceinline__ double kern_cos(double x) {
  const double C1 = sconst(4.16666666666666019037e-02), C2 = sconst(-1.38888888888741095749e-03),
               C3 = sconst(2.48015872894767294178e-05), C4 = sconst(-2.75573143513906633035e-07),
               C5 = sconst(2.08757232129817482790e-09), C6 = sconst(-1.13596475577881948265e-11);
  const double z = x * x;
  const double r = z * fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + z * r);
}
__device__ __forceinline__ void quadrant(int k, double s, double c, double& so, double& co) {
  const bool swap = (k & 1) != 0;
  const double a = swap ? c : s;   // candidate for sin
  const double b = swap ? s : c;   // candidate for cos
  // k&3: 0:(s,c) 1:(c,-s) 2:(-s,-c) 3:(-c,s)
  const bool negs = (k & 2) != 0;            // sin negated for k=2,3
  const bool negc = ((k + 1) & 2) != 0;      // cos negated for k=1,2
  so = negs ? -a : a;
  co = negc ? -b : b;
}
__device__ __forceinline__ void sincos2pi(double u, double& s, double& c) {
  const double PIO2 = sconst(1.57079632679489655800e+00);
  const double t = 4.0 * u;
  const double kd = floor(t + 0.5);
  const double r = t - kd;
  const double x = r * PIO2;
  quadrant((int)kd, kern_sin(x), kern_cos(x), s, c);
}
// Uniform point of the unit circle (same construction as the test oracle): psi in (-pi/4, pi/4), polynomial kernels without
// quadrant logic, two angle doublings.  23 instructions where sincos2pi takes 45.
__device__ __forceinline__ void circle_point_psi(double psi, double& c, double& s) {
  const double C1 = sconst(4.16666666666666019037e-02), C2 = sconst(-1.38888888888741095749e-03),
               C3 = sconst(2.48015872894767294178e-05), C4 = sconst(-2.75573143513906633035e-07),
               C5 = sconst(2.08757232129817482790e-09), C6 = sconst(-1.13596475577881948265e-11);
  const double z = psi * psi;
  const double sn = kern_sin(psi);
  const double cs = fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1), -0.5), 1.0);
  const double c2 = fma(cs, cs, -(sn * sn)), s2 = fma(cs, sn, cs * sn);
  c = fma(c2, c2, -(s2 * s2));
  s = fma(c2, s2, c2 * s2);
}
// the same from u in (0,1): psi = (u - 1/2) pi/2
__device__ __forceinline__ void circle_point(double u, double& c, double& s) {
  const double PIO2 = sconst(1.57079632679489655800e+00);
  circle_point_psi((u - 0.5) * PIO2, c, s);
}
__device__ __forceinline__ void sincos_cw(double x, double& s, double& c) {
  const double INVPIO2 = sconst(6.36619772367581382433e-01);
  const double PIO2_1 = sconst(1.57079632673412561417e+00);
  const double PIO2_1T = sconst(6.07710050650619224932e-11);
  const double kd = floor(fma(x, INVPIO2, 0.5));
  double r = fma(-kd, PIO2_1, x);
  r = fma(-kd, PIO2_1T, r);
  quadrant((int)kd, kern_sin(r), kern_cos(r), s, c);
}
__device__ __forceinline__ double log_pos(double x) {
  // (sconst: see above -- un-pinned, these 9 literals were hoisted into 18 VGPRs across the whole trace loop of every
  //  kernel that can reach a Gaussian draw, and the BRDF / full kernels spilled to scratch because of it)
  const double LN2_HI = sconst(6.93147180369123816490e-01), LN2_LO = sconst(1.90821492927058770002e-10);
  const double Lg1 = sconst(6.666666666666735130e-01), Lg2 = sconst(3.999999999940941908e-01), Lg3 = sconst(2.857142874366239149e-01),
               Lg4 = sconst(2.222219843214978396e-01), Lg5 = sconst(1.818357216161805012e-01), Lg6 = sconst(1.531383769920937332e-01),
               Lg7 = sconst(1.479819860511658591e-01);
  const uint64_t bits = (uint64_t)__double_as_longlong(x);
  int e = (int)(bits >> 52) - 1023;
  const uint64_t mant = bits & 0x000FFFFFFFFFFFFFull;
  const bool big = mant > 0x6A09E667F3BCCull;
  const uint64_t mb = mant | (big ? 0x3FE0000000000000ull : 0x3FF0000000000000ull);
  e += big ? 1 : 0;
  const double m = __longlong_as_double((long long)mb);
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
  const double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)e;
  return dk * LN2_HI - ((hfsq - (s * (hfsq + R) + dk * LN2_LO)) - f);
}

// ---------------------------------------------------------------- next boundary
__device__ __forceinline__ void consider(double t, const V3& q, int kind, double& best, V3& bq, int& bk) {
  if (t > 0.0 && t < best) { best = t; bq = q; bk = kind; }
}

// Generic nearest boundary (all surfaces).  Rare path: port transits, rim, outer sphere, box.
template <class G>
__device__ inline int next_hit_generic(const G& gg, const V3 p, const V3 v, const int on, V3& q_out) {
  // one read of each constant (gg may be a volatile LDS copy)
  struct { double rin2, rout2, zcut_in, zcut_out, k2, H; } g;
  g.rin2 = gg.rin2; g.rout2 = gg.rout2; g.zcut_in = gg.zcut_in; g.zcut_out = gg.zcut_out; g.k2 = gg.k2; g.H = gg.H;
  const double b = dot3(p, v);
  const double pp = dot3(p, p);
  double best = __builtin_inf();
  V3 bq = p;
  int bk = K_BOX;
  {
    const double ci = pp - g.rin2;
    const double di = fma(b, b, -ci);
    if (di >= 0.0) {
      const double s = sqrt(di);
      const double tn = -b - s, tf = s - b;
      if (on == K_INNER && b < 0.0) {          // rule S1' (see next_hit_s1)
        const V3 q = axpy(-2.0 * b, v, p);
        if (q.z >= g.zcut_in) { q_out = q; return K_INNER; }
      } else if (on == K_NONE && ci < 0.0) {
        const V3 q = axpy(tf, v, p);
        if (q.z >= g.zcut_in) { q_out = q; return K_INNER; }
      }
      const bool skip_n = (on == K_INNER && b < 0.0);
      const bool skip_f = (on == K_INNER && !(b < 0.0));
      if (!skip_n) { const V3 q = axpy(tn, v, p); if (q.z >= g.zcut_in) consider(tn, q, K_INNER, best, bq, bk); }
      if (!skip_f) { const V3 q = axpy(tf, v, p); if (q.z >= g.zcut_in) consider(tf, q, K_INNER, best, bq, bk); }
    }
  }
  {
    const double co = pp - g.rout2;
    const double dO = fma(b, b, -co);
    if (dO >= 0.0) {
      const double s = sqrt(dO);
      const double tn = -b - s, tf = s - b;
      const bool skip_n = (on == K_OUTER && b < 0.0);
      const bool skip_f = (on == K_OUTER && !(b < 0.0));
      if (!skip_n) { const V3 q = axpy(tn, v, p); if (q.z >= g.zcut_out) consider(tn, q, K_OUTER, best, bq, bk); }
      if (!skip_f) { const V3 q = axpy(tf, v, p); if (q.z >= g.zcut_out) consider(tf, q, K_OUTER, best, bq, bk); }
    }
  }
  {
    const double A = fma(-g.k2, v.z * v.z, fma(v.x, v.x, v.y * v.y));
    const double B = fma(-g.k2, p.z * v.z, fma(p.x, v.x, p.y * v.y));
    const double C = fma(-g.k2, p.z * p.z, fma(p.x, p.x, p.y * p.y));
    double tc0 = 0.0, tc1 = 0.0;
    int nc = 0;
    if (on == K_CONE) {
      if (A != 0.0) { tc0 = (-2.0 * B) / A; nc = 1; }
    } else if (A == 0.0) {
      if (B != 0.0) { tc0 = (-C) / (2.0 * B); nc = 1; }
    } else {
      const double D = fma(B, B, -(A * C));
      if (D >= 0.0) {
        const double sD = sqrt(D);
        tc0 = (-B - sD) / A;
        tc1 = (-B + sD) / A;
        nc = 2;
      }
    }
    if (nc >= 1) {
      const V3 q = axpy(tc0, v, p);
      const double rr = dot3(q, q);
      if (q.z < 0.0 && rr >= g.rin2 && rr <= g.rout2) consider(tc0, q, K_CONE, best, bq, bk);
    }
    if (nc >= 2) {
      const V3 q = axpy(tc1, v, p);
      const double rr = dot3(q, q);
      if (q.z < 0.0 && rr >= g.rin2 && rr <= g.rout2) consider(tc1, q, K_CONE, best, bq, bk);
    }
  }
  if (bk != K_BOX) { q_out = bq; return bk; }
  const double inf = __builtin_inf();
  const double tx = v.x > 0.0 ? (g.H - p.x) / v.x : (v.x < 0.0 ? (-g.H - p.x) / v.x : inf);
  const double ty = v.y > 0.0 ? (g.H - p.y) / v.y : (v.y < 0.0 ? (-g.H - p.y) / v.y : inf);
  const double tz = v.z > 0.0 ? (g.H - p.z) / v.z : (v.z < 0.0 ? (-g.H - p.z) / v.z : inf);
  double t = tx;
  if (ty < t) t = ty;
  if (tz < t) t = tz;
  q_out = axpy(t, v, p);
  return K_BOX;
}

// Rules S1/S1' alone: returns true (and q) if the far root of the inner sphere is the hit.
// S1' (the bounce-to-bounce case): leaving a point p of the inner sphere inwards (p.v < 0) the ray meets the sphere again at
// t = -2 (p.v)/(v.v) - the non-zero root of |p + t v|^2 = |p|^2 for a direction of ANY length - so the hot path needs
// neither a square root nor a unit vector: the cosine emission hands over n + s un-normalised (interact()).  A ray that
// leaves the rule gets a unit direction (unit_dir) before the generic search; the oracle takes the same two steps.
// FIRST: also the first segment of a ray (on == K_NONE: starts inside the ball; needs rin2 and a square root).  The
// persistent kernels pass FIRST only on step 0 of a loop trip -- rays start there -- and read rin2 from the LDS copy of
// the geometry when they need it; a case this function does not take simply falls to the generic search.
template <bool FIRST = true, class G>
__device__ __forceinline__ bool next_hit_s1(const Hot& h, const G& g, const V3& p, const V3& v, const int on, V3& q_out) {
  const double b = dot3(p, v);
  if (on == K_INNER) {
    if (!(b < 0.0)) return false;
    const double ia = neg_rcp_unit(dot3(v, v));
    const V3 q = axpy((2.0 * b) * ia, v, p);
    if (q.z >= h.zcut_in) { q_out = q; return true; }
    return false;
  }
  if (!FIRST || on != K_NONE) return false;
  const double pp = dot3(p, p);
  const double ci = pp - g.rin2;
  const double di = fma(b, b, -ci);
  if (!(di >= 0.0) || !(ci < 0.0)) return false;
  const double s = sqrt(di);
  const double tf = s - b;
  const V3 q = axpy(tf, v, p);
  if (q.z >= h.zcut_in) { q_out = q; return true; }
  return false;
}

// Rule S1 first (hot path); anything else falls to the generic search (which re-derives the same numbers).
// unit vector of v (three divisions by the length, as the oracle's unit_dir): what a direction becomes when the ray leaves rule S1'
__device__ __forceinline__ void unit_dir(V3& v) {
  const double mag = sqrt(dot3(v, v));
  v.x = v.x / mag; v.y = v.y / mag; v.z = v.z / mag;
}
template <class G>
__device__ __forceinline__ int next_hit(const Hot& h, const G& g, const V3& p, V3& v, const int on, V3& q_out) {
  if (next_hit_s1<true>(h, g, p, v, on, q_out)) return K_INNER;
  if (on == K_INNER) unit_dir(v);
  return next_hit_generic(g, p, v, on, q_out);
}
// ---------------------------------------------------------------- TVector3 arithmetic in ROOT's own op order (no fma)
__device__ __forceinline__ V3 tv_orthogonal(const V3& a) {
  const double xx = a.x < 0.0 ? -a.x : a.x, yy = a.y < 0.0 ? -a.y : a.y, zz = a.z < 0.0 ? -a.z : a.z;
  V3 r;
  if (xx < yy) {
    if (xx < zz) { r.x = 0; r.y = a.z; r.z = -a.y; } else { r.x = a.y; r.y = -a.x; r.z = 0; }
  } else {
    if (yy < zz) { r.x = -a.z; r.y = 0; r.z = a.x; } else { r.x = a.y; r.y = -a.x; r.z = 0; }
  }
  return r;
}
__device__ __forceinline__ V3 tv_cross(const V3& a, const V3& p) {
  V3 r;
  r.x = a.y * p.z - p.y * a.z; r.y = a.z * p.x - p.z * a.x; r.z = a.x * p.y - p.x * a.y;
  return r;
}
__device__ __forceinline__ V3 tv_unit(const V3& a) {
  const double tot2 = a.x * a.x + a.y * a.y + a.z * a.z;
  const double tot = (tot2 > 0) ? 1.0 / sqrt(tot2) : 1.0;
  V3 r;
  r.x = a.x * tot; r.y = a.y * tot; r.z = a.z * tot;
  return r;
}
// TVector3::Unit for the vectors of the lobe sampler, without the range scaling and the special-case fix-ups of the general square
// root and division: for a squared length in [2^-200, 2^200] sqrt_unit() and neg_rcp_unit() ARE the IEEE results (they are the
// compiler's own expansions minus the parts that only act below 2^-767 or on zero / inf / nan operands), so this is tv_unit() bit for
// bit -- checked on the device over the operand families in tests/test_gpu_round5.py -- and anything outside takes tv_unit() itself.
// Three of these per try of the rejection sampler: 38 -> 28 instructions each.
__device__ __forceinline__ V3 tv_unit_n(const V3& a) {
  const double tot2 = a.x * a.x + a.y * a.y + a.z * a.z;
  if (!(tot2 >= 0x1.0p-200 && tot2 <= 0x1.0p200)) return tv_unit(a);
  const double tot = -neg_rcp_unit(sqrt_unit(tot2));   // 1.0 / sqrt(tot2)
  V3 r;
  r.x = a.x * tot; r.y = a.y * tot; r.z = a.z * tot;
  return r;
}
__device__ __forceinline__ V3 tv_setmag1(const V3& a) {
  double f = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
  if (f == 0) return a;
  f = 1.0 / f;
  V3 r;
  r.x = a.x * f; r.y = a.y * f; r.z = a.z * f;
  return r;
}


// "nonLambertianFlux copy.C":31-70 (NonLambertianSurface): cos^2 lobe within 60 deg of the normal by rejection.
// Try k of interaction j draws from Philox block 2j + (stream>>1) of stream 16+k.
// ONE try of the rejection loop (generateScatteredDirection's loop body, "nonLambertianFlux copy.C":45-69): the candidate direction
// and whether it is accepted.  The local frame (w, u, v) is a function of the normal alone, so a caller that spreads the tries of
// one interaction over several steps (the tracer waves of the lobe pipeline: one try per step and lane, isx_kernels.hpp) forms it
// again for every try and gets the very same numbers as the loop below.
__device__ __forceinline__ bool lobe_try(const V3& normal, uint64_t seed, uint64_t ray, uint32_t j, uint32_t stream, uint32_t k, V3& sc) {
  const double maxAngle = 60.0 * 3.14159265358979323846 / 180.0;
  const V3 w = tv_unit_n(normal);
  V3 yxw; yxw.x = w.z; yxw.y = 0.0; yxw.z = -w.x;  // TVector3(0,1,0).Cross(w)
  const V3 u = tv_unit_n(yxw);
  const V3 vv = tv_cross(w, u);
  uint32_t r[4];
  draw_block(seed, ray, 2u * j + (stream >> 1), 16u + k, r);
  const double theta = maxAngle * u01(r[0]);
  double st, ct, sp, cp;
  sincos_cw(theta, st, ct);
  sincos2pi(u01(r[1]), sp, cp);
  const double x = st * cp, y = st * sp, z = ct;
  V3 t;
  t.x = x * u.x + y * vv.x + z * w.x; t.y = x * u.y + y * vv.y + z * w.y; t.z = x * u.z + y * vv.z + z * w.z;
  sc = tv_unit_n(t);
  const double c = sc.x * normal.x + sc.y * normal.y + sc.z * normal.z;
  const double p = c * c;
  return u01(r[2]) <= p;
}
constexpr uint32_t kLobeTries = 64;   // (the reference loops for ever; the acceptance is >= 1/4 per try, so 64 misses never happen: 2^-26 per 1e9 bounces)
// NonLambertianSurface::Reflection's hemisphere fix ("nonLambertianFlux copy.C":205-207)
__device__ __forceinline__ void lobe_hemisphere(const V3& normal, V3& sc) {
  if (sc.x * normal.x + sc.y * normal.y + sc.z * normal.z < 0) { sc.x = -sc.x; sc.y = -sc.y; sc.z = -sc.z; }
}
__device__ inline V3 lobe_sample(const V3 normal, uint64_t seed, uint64_t ray, uint32_t j, uint32_t stream) {
  V3 sc = tv_unit(normal);
  for (uint32_t k = 0; k < kLobeTries; k++)
    if (lobe_try(normal, seed, ray, j, stream, k, sc)) break;
  lobe_hemisphere(normal, sc);
  return sc;
}

// ---------------------------------------------------------------- surface interaction
__device__ __forceinline__ void onb(const V3& n, V3& t1, V3& t2) {
  const double sg = copysign(1.0, n.z);
  const double a = -1.0 / (sg + n.z);
  const double b = (n.x * n.y) * a;
  t1.x = fma(sg * n.x, n.x * a, 1.0);
  t1.y = sg * b;
  t1.z = -(sg * n.x);
  t2.x = b;
  t2.y = fma(n.y, n.y * a, sg);
  t2.z = -n.y;
}

template <class G>
__device__ __forceinline__ V3 surface_normal(const Hot& h, const G& g, int kind, const V3& q) {
  V3 n;
  if (kind == K_INNER) {
    n.x = q.x * h.ninv_rin; n.y = q.y * h.ninv_rin; n.z = q.z * h.ninv_rin;
  } else if (kind == K_OUTER) {
    const double ir = g.inv_rout;
    n.x = q.x * ir; n.y = q.y * ir; n.z = q.z * ir;
  } else {
    const double gz = g.k2 * q.z;
    const double nn = sqrt(fma(q.x, q.x, fma(q.y, q.y, gz * gz)));
    n.x = -q.x / nn; n.y = -q.y / nn; n.z = gz / nn;
  }
  return n;
}

// returns false if absorbed; otherwise v is the re-emitted direction
// ISX_TRACE_CHORD: Lambertian bounce off the inner sphere via the integrating-sphere identity - for cosine-law
// emission from a point of a sphere the far intersection is uniform over the sphere's area, so the next wall
// point T is sampled directly: no direction, no orthonormal basis, no intersection.  Same Philox words as the
// explicit bounce (a -> z, b -> absorb + azimuth).  Returns false if absorbed.
// z of a uniform point of the unit sphere from a Philox word: 1 - 2 (w + 1/2) 2^-32 (every operation exact)
// (as ONE fused operation on the converted word: 1 - 2^-32 - w 2^-31 = (2^32 - 1 - 2w) / 2^32 has at most 33 significant bits, so
//  it is the same real number, exactly represented, as fma(-2, (w + 0.5) 2^-32, 1) -- the oracle's expression -- two instructions less)
__device__ __forceinline__ double sphere_z(uint32_t w) { return fma((double)w, sconst(-0x1.0p-31), sconst(1.0 - 0x1.0p-32)); }
__device__ __forceinline__ bool interact_chord(const Hot& h, V3& T, uint32_t wa, uint32_t wb) {
  if (!((unsigned long long)wb < h.rho_thr)) return false;   // u01(wb) < rho, decided on the integer
  const double zz = sphere_z(wa);
  const double s2 = sqrt_unit(fma(-zz, zz, 1.0));   // 1 - zz^2 in [2^-31, 1]
  double sf, cf;
  circle_point_psi(fma((double)wb, h.psi_k1, h.psi_k0), cf, sf);
  const double rxy = h.r_in * s2;
  T.x = rxy * cf; T.y = rxy * sf; T.z = h.r_in * zz;
  return true;
}

// LEAN = the configuration of the headline path (ROBAST Lambertian border, pencil source): the other surface
// models are compiled out so their registers and code do not burden the hot kernel.
// SURF: the border's model at compile time -- SURF_LAMBERT (= LEAN), SURF_LOBE, SURF_ROUGH (ROBAST's non-Lambertian border:
// specular reflection about a normal tilted by the Gaussian roughness) -- or SURF_ANY: decided at run time from h.surface_model /
// h.lambertian (round 1's full-featured kernels and the end-state interface).
enum : int { SURF_ANY = -1, SURF_LAMBERT = 0, SURF_LOBE = 1, SURF_ROUGH = 2 };
// what interact() does to a lobe / rough-specular direction before it hands it back (also the tail of the lobe pipeline's tries)
__device__ __forceinline__ void finish_direction(const V3& n, V3& w) {
  // into-wall fix
  const double dn = dot3(w, n);
  if (dn <= 0.0) w = axpy(-2.0 * dn, n, w);
  // one Newton step towards unit length (keeps the |v| error at rounding level instead of letting it
  // random-walk multiplicatively through hit point -> normal -> new direction)
  const double k = fma(-0.5, dot3(w, w), sconst(1.5));
  w.x *= k; w.y *= k; w.z *= k;
}
template <bool LEAN, int SURF = (LEAN ? SURF_LAMBERT : SURF_ANY), class G>
__device__ __forceinline__ bool interact(const Hot& h, const G& g, int kind, const V3& q, V3& v, uint64_t seed,
                                         uint64_t ray, uint32_t j, uint32_t stream, uint32_t wa, uint32_t wb) {
  if (!((unsigned long long)wb < h.rho_thr)) return false;   // u01(wb) < rho, decided on the integer
  V3 w;
  if (SURF == SURF_LAMBERT || (SURF == SURF_ANY && h.surface_model != 1 && h.lambertian)) {
    // cosine-law re-emission about the geometric normal; roughness does not act on a Lambertian border.
    // w = n + s, s uniform on the unit sphere in WORLD coordinates (z from word a, azimuth from word b): the direction of
    // n + s follows the cosine law about n exactly; no local frame, and w stays UN-NORMALISED on the inner sphere, where
    // the next step is rule S1' (any length): there w = r_in (n + s) = r_in s - q and the normal is never formed; anywhere
    // else (rim, outer sphere) n + s is normalised here.  (oracle: cosine_emission())
    const double zs = sphere_z(wa);
    const double rs = sqrt_unit(fma(-zs, zs, 1.0));   // 1 - zs^2 in [2^-31, 1]
    double sf, cf;
    circle_point_psi(fma((double)wb, h.psi_k1, h.psi_k0), cf, sf);
    if (kind == K_INNER) {
      const double Rrs = h.r_in * rs;
      w.x = fma(Rrs, cf, -q.x);
      w.y = fma(Rrs, sf, -q.y);
      w.z = fma(h.r_in, zs, -q.z);
    } else {
      const V3 n = surface_normal(h, g, kind, q);
      w.x = fma(rs, cf, n.x);
      w.y = fma(rs, sf, n.y);
      w.z = n.z + zs;
      unit_dir(w);
    }
    v = w;
    return true;
  }
  const V3 n = surface_normal(h, g, kind, q);
  if (SURF == SURF_LOBE || (SURF == SURF_ANY && h.surface_model == 1)) {
    w = lobe_sample(n, seed, ray, j, stream);
  } else {
    V3 M = n;
    const double sigma = g.sigma;
    if (sigma != 0.0) {
      V3 A, Bv;
      onb(n, A, Bv);
      uint32_t wr[4];
      draw_block(seed, ray, j, stream + 64u, wr);   // the roughness draws have their own stream
      const double u1 = u01(wr[0]), u2 = u01(wr[1]), u3 = u01(wr[2]);
      const double R = sqrt(-2.0 * log_pos(u1));
      double s2, c2;
      sincos2pi(u2, s2, c2);
      const double delta = sigma * (R * c2);
      double sd, cd, sp, cp;
      sincos_cw(delta, sd, cd);
      sincos2pi(u3, sp, cp);
      V3 e;
      e.x = fma(cp, A.x, sp * Bv.x); e.y = fma(cp, A.y, sp * Bv.y); e.z = fma(cp, A.z, sp * Bv.z);
      M.x = fma(cd, n.x, sd * e.x); M.y = fma(cd, n.y, sd * e.y); M.z = fma(cd, n.z, sd * e.z);
    }
    const double d2 = -2.0 * dot3(v, M);
    w = axpy(d2, M, v);
  }
  // (only the lobe and rough-specular surfaces get here: the cosine emission returned above)
  finish_direction(n, w);
  v = w;
  return true;
}

// ---------------------------------------------------------------- BRDF re-scatter (nonLambertianFlux.C:147-208)
template <class G>
__device__ inline V3 brdf_sample(const G& g, const V3 normal, const V3 incident, uint64_t seed, uint64_t ray) {
  uint32_t w[4];
  draw_block(seed, ray, 0u, 1u, w);
  if (u01(w[0]) < g.brdf_spec) {
    const double a = 2 * (incident.x * normal.x + incident.y * normal.y + incident.z * normal.z);
    V3 refl;
    refl.x = incident.x - a * normal.x; refl.y = incident.y - a * normal.y; refl.z = incident.z - a * normal.z;
    refl = tv_setmag1(refl);
    const double Rg = sqrt(-2.0 * log_pos(u01(w[1])));
    double s2, c2;
    sincos2pi(u01(w[2]), s2, c2);
    const double theta = g.brdf_theta_scale * (Rg * c2);
    double st, ct, sp, cp;
    sincos_cw(theta, st, ct);
    sincos2pi(u01(w[3]), sp, cp);
    const V3 p1 = tv_orthogonal(refl);
    const V3 p2 = tv_cross(refl, p1);
    V3 res;
    res.x = refl.x + st * (cp * p1.x + sp * p2.x);
    res.y = refl.y + st * (cp * p1.y + sp * p2.y);
    res.z = refl.z + st * (cp * p1.z + sp * p2.z);
    return tv_setmag1(res);
  } else {
    const double u = u01(w[1]);
    const double ct = sqrt(u), st = sqrt(1.0 - u);
    double sp, cp;
    sincos2pi(u01(w[3]), sp, cp);
    const V3 uu = tv_orthogonal(normal);
    const V3 vv = tv_cross(normal, uu);
    const double x = st * cp, y = st * sp, z = ct;
    V3 res;
    res.x = x * uu.x + y * vv.x + z * normal.x;
    res.y = x * uu.y + y * vv.y + z * normal.y;
    res.z = x * uu.z + y * vv.z + z * normal.z;
    return tv_unit(res);
  }
}

// ---------------------------------------------------------------- detector test (exact; fluxAtObserver.C:70-107)
// det = x,y,z,nx,ny,nz exactly as Detector::setPosition stores them; plain +,-,*,/ in source order.
__device__ __forceinline__ bool check_intersection(const double* __restrict__ det, double half_w2, const V3& lp,
                                                   const V3& dir) {
  double e[6];
  load6(det, e);   // (the detector table lives in global memory whatever pointer type reached this point)
  const double x = e[0], y = e[1], z = e[2], nx = e[3], ny = e[4], nz = e[5];
  const double dot = dir.x * nx + dir.y * ny + dir.z * nz;
  if (fabs(dot) < 1e-10) return false;
  const double dx = lp.x - x;
  const double dy = lp.y - y;
  const double dz = lp.z - z;
  const double t = -(dx * nx + dy * ny + dz * nz) / dot;
  const double ix = lp.x + dir.x * t;
  const double iy = lp.y + dir.y * t;
  const double iz = lp.z + dir.z * t;
  const double rx = ix - x;
  const double ry = iy - y;
  const double rz = iz - z;
  const double ux = ny * rz - nz * ry;
  const double uy = nz * rx - nx * rz;
  const double uz = nx * ry - ny * rx;
  const double r2 = ux * ux + uy * uy + uz * uz;
  return r2 <= half_w2;
}

}  // namespace isx
