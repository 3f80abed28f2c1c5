/*
 * isx_oracle.c — CPU ORACLE (test infrastructure, NOT product code; see isx_oracle.h).
 *
 * Plain C11 + OpenMP.  Build: oracle/Makefile  (-O2 -ffp-contract=off -mfma).
 * Every floating-point operation below is an IEEE-754 binary64 +,-,*,/,sqrt or an
 * EXPLICIT fma(); with contraction off the expression trees in this file ARE the
 * numeric specification the HIP kernels must reproduce bit for bit.
 *
 * What follows which part of the reference:
 *   geometry parameters ............ fluxAtObserverOptimize.C:33-41,192-230
 *   source ray ..................... fluxAtObserverOptimize.C:289-292 (ARay ctor, direction normalised)
 *   trace loop (ROBAST, inferred) .. call sites fluxAtObserverOptimize.C:254,295; SURVEY.md §8a a2
 *   port test ...................... fluxAtObserver.C:162-166
 *   detector position + hit test ... fluxAtObserver.C:49-107
 *   grid, fraction ................. fluxAtObserver.C:352-358, fluxAtObserverOptimize.C:571
 *   BRDF re-scatter ................ nonLambertianFlux.C:147-208,235-304
 *   physical disc .................. integratingSphereDetectorSweep.C:134-172
 *   exit-direction histogram ....... distributionSphereDetectorSweep.C:61-103
 */
#include "isx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* vectors                                                                   */
/* ------------------------------------------------------------------------- */
typedef struct { double x, y, z; } v3;

static inline double dot3(v3 a, v3 b) { return fma(a.x, b.x, fma(a.y, b.y, a.z * b.z)); }
/* q = p + t*v, one fma per component */
static inline v3 axpy(double t, v3 v, v3 p) {
  v3 q = { fma(t, v.x, p.x), fma(t, v.y, p.y), fma(t, v.z, p.z) };
  return q;
}

/* ------------------------------------------------------------------------- */
/* Philox4x32-10 (Salmon et al., SC'11; Random123 constants)                  */
/* ------------------------------------------------------------------------- */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

void isxo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
    uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += PHILOX_W0; k1 += PHILOX_W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* RNG addressing: key = (seed_lo, seed_hi); counter = (ray_lo, ray_hi, block, stream).
 * stream 0: primary trace; 1: BRDF re-scatter draw; 2: scattered-ray trace.
 * Mirror interaction j (0-based) of a trace in stream s takes two words (a, b) from block j/2 of stream s -- words
 * (0,1) if j is even, (2,3) if odd: a -> polar angle, b -> absorption (survive iff b < rho_thr) and, rescaled by
 * 1/rho_thr, the azimuth.  The rough-specular branch draws block j of stream s+64 (Box-Muller u1,u2, azimuth);
 * the cos^2-lobe rejection tries use streams 16+k. */
static inline void draw_block(uint64_t seed, uint64_t ray, uint32_t block, uint32_t stream, uint32_t w[4]) {
  uint32_t ctr[4] = { (uint32_t)ray, (uint32_t)(ray >> 32), block, stream };
  uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
  isxo_philox4x32_10(ctr, key, w);
}

/* u in (0,1): exact in binary64 */
double isxo_u01(uint32_t w) { return ((double)w + 0.5) * 0x1.0p-32; }

/* ------------------------------------------------------------------------- */
/* deterministic elementary functions (fdlibm-style kernels, fma Horner)     */
/* ------------------------------------------------------------------------- */
static inline double kern_sin(double x) { /* |x| <= pi/4 (+slack) */
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x;
  double r = fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2);
  double v = z * x;
  return fma(v, fma(z, r, S1), x);
}
static inline double kern_cos(double x) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double r = z * fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
  double hz = 0.5 * z;
  double w = 1.0 - hz;
  /* cos = w + (((1-w)-hz) + z*r)   (fdlibm's compensated form, y = 0) */
  return w + (((1.0 - w) - hz) + z * r);
}
static inline void quadrant(int k, double s, double c, double* so, double* co) {
  switch (k & 3) {
    case 0: *so = s;  *co = c;  break;
    case 1: *so = c;  *co = -s; break;
    case 2: *so = -s; *co = -c; break;
    default: *so = -c; *co = s; break;
  }
}
/* sin/cos(2*pi*u), u in [0,1): t=4u, k=floor(t+0.5), r=t-k in [-.5,.5], x=r*pi/2 */
void isxo_sincos2pi(double u, double* s, double* c) {
  const double PIO2 = 1.57079632679489655800e+00;
  double t = 4.0 * u;
  double kd = floor(t + 0.5);
  double r = t - kd;
  double x = r * PIO2;
  quadrant((int)kd, kern_sin(x), kern_cos(x), s, c);
}
/* A point uniformly distributed on the unit circle, for the azimuth of the cosine emission (only uniformity matters
 * there, not which angle a given word maps to): psi lies in (-pi/4, pi/4), where the polynomial kernels need no quadrant
 * logic, and two angle doublings carry it to 4 psi, uniform on (-pi, pi).  The cosine is the plain Horner form (no
 * fdlibm tail correction): |c^2 + s^2 - 1| < 3e-15 -- the emission adds this point, scaled, to the surface normal and
 * the intersection that follows is exact for a direction of any length, so nothing relies on it being exactly unit. */
static inline void circle_point_psi(double psi, double* c, double* s) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = psi * psi;
  double sn = kern_sin(psi);
  double cs = fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1), -0.5), 1.0);
  double c2 = fma(cs, cs, -(sn * sn)), s2 = fma(cs, sn, cs * sn);
  *c = fma(c2, c2, -(s2 * s2));
  *s = fma(c2, s2, c2 * s2);
}
/* the same from u in (0,1): psi = (u - 1/2) pi/2 */
void isxo_circle_point(double u, double* c, double* s) {
  const double PIO2 = 1.57079632679489655800e+00;
  circle_point_psi((u - 0.5) * PIO2, c, s);
}
/* general argument, Cody-Waite two-constant reduction by pi/2 */
void isxo_sincos(double x, double* s, double* c) {
  const double INVPIO2 = 6.36619772367581382433e-01;
  const double PIO2_1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
  const double PIO2_1T = 6.07710050650619224932e-11; /* pi/2 - PIO2_1 */
  double kd = floor(fma(x, INVPIO2, 0.5));
  double r = fma(-kd, PIO2_1, x);
  r = fma(-kd, PIO2_1T, r);
  quadrant((int)(long long)kd, kern_sin(r), kern_cos(r), s, c);
}
/* natural log of a positive normal double (fdlibm e_log structure) */
double isxo_log(double x) {
  const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  uint64_t bits;
  memcpy(&bits, &x, 8);
  int e = (int)(bits >> 52) - 1023;
  uint64_t mant = bits & 0x000FFFFFFFFFFFFFull;
  uint64_t mb;
  if (mant > 0x6A09E667F3BCCull) { /* m > sqrt(2): use m/2, e+1 */
    mb = mant | 0x3FE0000000000000ull;
    e += 1;
  } else {
    mb = mant | 0x3FF0000000000000ull;
  }
  double m;
  memcpy(&m, &mb, 8);
  double f = m - 1.0;
  double s = f / (2.0 + f);
  double z = s * s;
  double w = z * z;
  double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
  double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
  double R = t2 + t1;
  double hfsq = 0.5 * f * f;
  double dk = (double)e;
  return dk * LN2_HI - ((hfsq - (s * (hfsq + R) + dk * LN2_LO)) - f);
}

/* ------------------------------------------------------------------------- */
/* prepared geometry                                                          */
/* ------------------------------------------------------------------------- */
enum { K_NONE = 0, K_INNER = 1, K_OUTER = 2, K_CONE = 3, K_BOX = 4 };

typedef struct {
  double rin2, rout2, zcut_in, zcut_out, k2, ninv_rin, inv_rout, H, rho, sigma;
  int lambertian, limit, surface_model, chord;
  double r_in;
  uint64_t rho_thr;   /* survive iff word < rho_thr  <=>  (word + 0.5) 2^-32 < rho */
  double inv_thr;     /* 1 / rho_thr: the surviving word b, rescaled, is the azimuth's uniform u2 = (b + 1/2) / rho_thr */
  double psi_k1, psi_k0;   /* circle_point_psi's angle straight from the word: psi = (u2 - 1/2) pi/2 = fma(b, psi_k1, psi_k0) */
  v3 src, dir0;
} geom;

static int prepare(const isxo_config* c, geom* g) {
  if (!(c->r_in > 0) || !(c->r_out > c->r_in)) return -2;
  if (!(c->theta_max_deg > 90.0) || !(c->theta_max_deg < 180.0)) return -2;
  if (!(c->box_half > c->r_out)) return -2;
  if (c->max_points < 1) return -2;
  g->rin2 = c->r_in * c->r_in;
  g->rout2 = c->r_out * c->r_out;
  double th = c->theta_max_deg * M_PI / 180.0;
  double ct = cos(th);
  double tt = tan(th);
  g->zcut_in = c->r_in * ct;
  g->zcut_out = c->r_out * ct;
  g->k2 = tt * tt;
  g->ninv_rin = -1.0 / c->r_in;
  g->inv_rout = 1.0 / c->r_out;
  g->H = c->box_half;
  g->rho = c->reflectance;
  {
    /* (w + 0.5) * 2^-32 < rho  <=>  w < rho * 2^32 - 0.5 =: x (both scalings exact)  <=>  w < ceil(x) for integer w */
    double x = ldexp(c->reflectance, 32) - 0.5;
    g->rho_thr = !(x > 0.0) ? 0ull : (x >= 4294967296.0 ? 4294967296ull : (uint64_t)ceil(x));
    g->inv_thr = g->rho_thr ? 1.0 / (double)g->rho_thr : 0.0;
    g->psi_k1 = g->inv_thr * 1.57079632679489655800e+00;
    g->psi_k0 = (0.5 * g->inv_thr - 0.5) * 1.57079632679489655800e+00;
  }
  g->sigma = c->roughness_rad;
  g->lambertian = c->lambertian;
  g->limit = c->max_points;
  g->surface_model = c->surface_model;
  if (c->trace_mode != 0 && c->trace_mode != 1) return -2;
  g->chord = c->trace_mode;
  g->r_in = c->r_in;
  if (c->surface_model != 0 && c->surface_model != 1) return -2;
  if (c->hit_line_mode != 0 && c->hit_line_mode != 1) return -2;
  g->src.x = c->src[0]; g->src.y = c->src[1]; g->src.z = c->src[2];
  /* ARay constructor normalises the direction (fluxAtObserverOptimize.C:290 passes (5,0,0)) */
  double dx = c->dir[0], dy = c->dir[1], dz = c->dir[2];
  double mag = sqrt(dx * dx + dy * dy + dz * dz);
  if (!(mag > 0)) return -2;
  g->dir0.x = dx / mag; g->dir0.y = dy / mag; g->dir0.z = dz / mag;
  return 0;
}

/* ------------------------------------------------------------------------- */
/* next boundary along p + t v  (replaces TGeoNavigator::FindNextBoundaryAndStep
 * over TGeoSphere(r_in,r_out,0,theta_max) inside TGeoBBox(H))                 */
/* ------------------------------------------------------------------------- */
static inline void consider(double t, v3 q, int kind, double* best, v3* bq, int* bk) {
  if (t > 0.0 && t < *best) { *best = t; *bq = q; *bk = kind; }
}

/* Rule S1' in its general form: a ray that leaves a point p of the inner sphere inwards (p.v < 0) meets the sphere again at
 * t = -2 (p.v)/(v.v) -- the non-zero root of |p + t v|^2 = |p|^2, for a direction v of ANY length -- and if that point lies
 * on the mirror patch nothing can be nearer.  The cosine emission of interact() hands over an un-normalised direction, so
 * the bounce-to-bounce step needs neither a square root nor a unit vector; |q| - r_in random-walks at the 1e-16 level
 * (|q|^2 = |p|^2 + t (2 p.v + t v.v), and the bracket vanishes to rounding) instead of being re-solved.
 * Returns 1 and q if the rule applies. */
static inline int s1_inner(const geom* g, v3 p, v3 v, v3* q_out) {
  double b = dot3(p, v);
  if (!(b < 0.0)) return 0;
  double vv = dot3(v, v);
  double ia = -1.0 / vv;
  double t = (2.0 * b) * ia;
  v3 q = axpy(t, v, p);
  if (q.z >= g->zcut_in) { *q_out = q; return 1; }
  return 0;
}
/* unit vector of v (three divisions by the length, as the ARay constructor does it): what a direction becomes when the
 * ray leaves rule S1' for the general search */
static inline v3 unit_dir(v3 v) {
  double mag = sqrt(dot3(v, v));
  v3 r = { v.x / mag, v.y / mag, v.z / mag };
  return r;
}

/* general search; v is a UNIT vector */
static int next_hit_general(const geom* g, v3 p, v3 v, int on, v3* q_out) {
  double b = dot3(p, v);
  double pp = dot3(p, p);
  double best = INFINITY;
  v3 bq = p;
  int bk = K_BOX;

  /* inner sphere r = r_in, mirror patch z >= zcut_in */
  double ci = pp - g->rin2;
  double di = fma(b, b, -ci);
  if (di >= 0.0) {
    double s = sqrt(di);
    double tn = -b - s, tf = s - b;
    /* Rule S1: a ray inside (or on, heading into) the inner ball whose far root lies on
     * the mirror patch hits there; nothing else can be nearer.
     * Rule S1' for a unit direction: leaving the inner sphere inwards the far root is taken as -2b (the root of
     * t^2 + 2bt = 0).  trace_one() tries s1_inner() first; this branch is reached only by a ray whose s1_inner() point
     * missed the mirror patch (it re-derives the point from the unit direction). */
    if (on == K_INNER && b < 0.0) {
      v3 q = axpy(-2.0 * b, v, p);
      if (q.z >= g->zcut_in) { *q_out = q; return K_INNER; }
    } else if (on == K_NONE && ci < 0.0) {
      v3 q = axpy(tf, v, p);
      if (q.z >= g->zcut_in) { *q_out = q; return K_INNER; }
    }
    int skip_n = (on == K_INNER && b < 0.0);  /* self root ~0 */
    int skip_f = (on == K_INNER && !(b < 0.0));
    if (!skip_n) { v3 q = axpy(tn, v, p); if (q.z >= g->zcut_in) consider(tn, q, K_INNER, &best, &bq, &bk); }
    if (!skip_f) { v3 q = axpy(tf, v, p); if (q.z >= g->zcut_in) consider(tf, q, K_INNER, &best, &bq, &bk); }
  }
  /* outer sphere r = r_out, mirror patch z >= zcut_out */
  double co = pp - g->rout2;
  double dO = fma(b, b, -co);
  if (dO >= 0.0) {
    double s = sqrt(dO);
    double tn = -b - s, tf = s - b;
    int skip_n = (on == K_OUTER && b < 0.0);
    int skip_f = (on == K_OUTER && !(b < 0.0));
    if (!skip_n) { v3 q = axpy(tn, v, p); if (q.z >= g->zcut_out) consider(tn, q, K_OUTER, &best, &bq, &bk); }
    if (!skip_f) { v3 q = axpy(tf, v, p); if (q.z >= g->zcut_out) consider(tf, q, K_OUTER, &best, &bq, &bk); }
  }
  /* conical rim x^2+y^2 = k2 z^2, z<0, r_in <= r <= r_out */
  {
    double A = fma(-g->k2, v.z * v.z, fma(v.x, v.x, v.y * v.y));
    double B = fma(-g->k2, p.z * v.z, fma(p.x, v.x, p.y * v.y));
    double C = fma(-g->k2, p.z * p.z, fma(p.x, p.x, p.y * p.y));
    double tc[2];
    int nc = 0;
    if (on == K_CONE) {
      if (A != 0.0) tc[nc++] = (-2.0 * B) / A;
    } else if (A == 0.0) {
      if (B != 0.0) tc[nc++] = (-C) / (2.0 * B);
    } else {
      double D = fma(B, B, -(A * C));
      if (D >= 0.0) {
        double sD = sqrt(D);
        tc[nc++] = (-B - sD) / A;
        tc[nc++] = (-B + sD) / A;
      }
    }
    for (int i = 0; i < nc; ++i) {
      v3 q = axpy(tc[i], v, p);
      double rr = dot3(q, q);
      if (q.z < 0.0 && rr >= g->rin2 && rr <= g->rout2) consider(tc[i], q, K_CONE, &best, &bq, &bk);
    }
  }
  if (bk != K_BOX) { *q_out = bq; return bk; }
  /* world box */
  double tx = v.x > 0.0 ? (g->H - p.x) / v.x : (v.x < 0.0 ? (-g->H - p.x) / v.x : INFINITY);
  double ty = v.y > 0.0 ? (g->H - p.y) / v.y : (v.y < 0.0 ? (-g->H - p.y) / v.y : INFINITY);
  double tz = v.z > 0.0 ? (g->H - p.z) / v.z : (v.z < 0.0 ? (-g->H - p.z) / v.z : INFINITY);
  double t = tx;
  if (ty < t) t = ty;
  if (tz < t) t = tz;
  *q_out = axpy(t, v, p);
  return K_BOX;
}

/* One boundary step: rule S1' while the ray bounces inside the inner sphere, else the general search along the unit
 * direction (*v is replaced by it: the direction of an exiting ray is a unit vector, as ROBAST's is). */
static int next_hit(const geom* g, v3 p, v3* v, int on, v3* q_out) {
  if (on == K_INNER) {
    if (s1_inner(g, p, *v, q_out)) return K_INNER;
    *v = unit_dir(*v);
  }
  return next_hit_general(g, p, *v, on, q_out);
}

/* ------------------------------------------------------------------------- */
/* surface interaction (replaces ROBAST mirror handling: reflectance test,    */
/* Gaussian roughness, Lambertian re-emission)              */
/* ------------------------------------------------------------------------- */
/* Branch-free orthonormal basis (Duff et al. 2017) */
static inline void onb(v3 n, v3* t1, v3* t2) {
  double sg = copysign(1.0, n.z);
  double a = -1.0 / (sg + n.z);
  double b = (n.x * n.y) * a;
  t1->x = fma(sg * n.x, n.x * a, 1.0);
  t1->y = sg * b;
  t1->z = -(sg * n.x);
  t2->x = b;
  t2->y = fma(n.y, n.y * a, sg);
  t2->z = -n.y;
}

static inline v3 surface_normal(const geom* g, int kind, v3 q) {
  v3 n;
  if (kind == K_INNER) {
    n.x = q.x * g->ninv_rin; n.y = q.y * g->ninv_rin; n.z = q.z * g->ninv_rin;
  } else if (kind == K_OUTER) {
    n.x = q.x * g->inv_rout; n.y = q.y * g->inv_rout; n.z = q.z * g->inv_rout;
  } else { /* cone: towards the axis */
    double gz = g->k2 * q.z;
    double nn = sqrt(fma(q.x, q.x, fma(q.y, q.y, gz * gz)));
    n.x = -q.x / nn; n.y = -q.y / nn; n.z = gz / nn;
  }
  return n;
}

/* "nonLambertianFlux copy.C":31-70: cos^2 lobe within 60 deg of the normal, rejection sampled
 * (TVector3 arithmetic in its own order).  Try k of interaction j draws from Philox block
 * 2j + (stream>>1) of stream 16+k. */
static inline v3 tv_unit(v3 a);
static inline v3 tv_cross(v3 a, v3 p);
static inline double tv_dot(v3 a, v3 b);
static v3 lobe_sample(v3 normal, uint64_t seed, uint64_t ray, uint32_t j, uint32_t stream) {
  const double maxAngle = 60.0 * M_PI / 180.0;
  v3 w = tv_unit(normal);
  v3 yxw = { w.z, 0.0, -w.x };  /* TVector3(0,1,0).Cross(w) */
  v3 u = tv_unit(yxw);
  v3 vv = tv_cross(w, u);
  v3 sc = w;
  for (uint32_t k = 0; k < 64; k++) {
    uint32_t r[4];
    draw_block(seed, ray, 2u * j + (stream >> 1), 16u + k, r);
    double theta = maxAngle * isxo_u01(r[0]);
    double st, ct, sp, cp;
    isxo_sincos(theta, &st, &ct);
    isxo_sincos2pi(isxo_u01(r[1]), &sp, &cp);
    double x = st * cp, y = st * sp, z = ct;
    v3 t = { x * u.x + y * vv.x + z * w.x, x * u.y + y * vv.y + z * w.y, x * u.z + y * vv.z + z * w.z };
    sc = tv_unit(t);
    double c = tv_dot(sc, normal);
    double p = c * c;   /* pow(|cosTheta|, 2.0) */
    if (isxo_u01(r[2]) <= p) break;
  }
  if (tv_dot(sc, normal) < 0) { sc.x = -sc.x; sc.y = -sc.y; sc.z = -sc.z; }
  return sc;
}

/* cosine-law emission from the surface point q of boundary `kind` for the two Philox words of the interaction
 * (a: z of the sphere point, b: its azimuth; b is a surviving word, b < rho_thr) -- see interact() */
static inline v3 cosine_emission(const geom* g, int kind, v3 q, uint32_t wa, uint32_t wb) {
  double zs = fma(-0x1.0p-31, (double)wa, 1.0 - 0x1.0p-32);   /* = 1 - 2 (wa + 1/2) 2^-32, exact */
  double rs = sqrt(fma(-zs, zs, 1.0));                         /* 1 - zs^2 in [2^-31, 1] */
  double sf, cf;
  circle_point_psi(fma((double)wb, g->psi_k1, g->psi_k0), &cf, &sf);
  v3 w;
  if (kind == K_INNER) {
    double Rrs = g->r_in * rs;
    w.x = fma(Rrs, cf, -q.x);
    w.y = fma(Rrs, sf, -q.y);
    w.z = fma(g->r_in, zs, -q.z);
  } else {
    v3 n = surface_normal(g, kind, q);
    w.x = fma(rs, cf, n.x);
    w.y = fma(rs, sf, n.y);
    w.z = n.z + zs;
    w = unit_dir(w);
  }
  return w;
}

/* returns 0 if absorbed, 1 otherwise (v updated) */
static int interact(const geom* g, int kind, v3 q, v3* v, uint64_t seed, uint64_t ray, uint32_t j, uint32_t stream) {
  /* Random words of interaction j: block j/2 of the trace's stream serves two interactions, words
   * (0,1) the even one, (2,3) the odd one.  Word a -> polar angle.  Word b -> absorption AND azimuth: the ray survives
   * iff b < rho_thr, and a surviving b is uniform on [0, rho_thr), so (b + 1/2)/rho_thr is a fresh uniform. */
  uint32_t wl[4];
  draw_block(seed, ray, j >> 1, stream, wl);
  const uint32_t wa = wl[2u * (j & 1u)], wb = wl[2u * (j & 1u) + 1u];
  if (!((uint64_t)wb < g->rho_thr)) return 0;
  v3 n = surface_normal(g, kind, q);
  v3 w;
  if (g->surface_model == 1) {
    w = lobe_sample(n, seed, ray, j, stream);
  } else if (g->lambertian) {
    /* EnableLambertian(true): cosine-law re-emission about the GEOMETRIC normal.  The Gaussian
     * roughness does not act on a Lambertian border: the reference's own sigma=0.5 map
     * (flux_at_observer/fluxmap_data.csv) is reproduced with the roughness ignored and is
     * missed by 9.5 % on axis with a roughness-tilted normal. */
    /* w = n + s with s uniformly distributed on the unit sphere: the direction of n + s follows the cosine law about n
     * exactly (the sphere of radius 1 about the tip of n touches the surface at the emission point: the integrating-sphere
     * identity in the small).  s is sampled in WORLD coordinates -- z uniform in (-1, 1), azimuth uniform -- so no
     * local frame is needed, and w is left UN-NORMALISED on the inner sphere, where the next step is s1_inner(), which
     * takes a direction of any length: there w = r_in (n + s) = r_in s - q for the inward normal n = -q / r_in, and the
     * normal itself is never formed.  Anywhere else (rim, outer sphere) w = n + s is normalised here. */
    *v = cosine_emission(g, kind, q, wa, wb);
    return 1;
  } else {
    /* specular reflection about the normal, tilted by a Gaussian polar angle (SetGaussianRoughness) */
    v3 M = n;
    if (g->sigma != 0.0) {
      v3 A, Bv;
      onb(n, &A, &Bv);
      uint32_t wr[4];
      draw_block(seed, ray, j, stream + 64u, wr);   /* the roughness draws have their own stream */
      double u1 = isxo_u01(wr[0]), u2 = isxo_u01(wr[1]), u3 = isxo_u01(wr[2]);
      double R = sqrt(-2.0 * isxo_log(u1));
      double s2, c2;
      isxo_sincos2pi(u2, &s2, &c2);
      double delta = g->sigma * (R * c2);
      double sd, cd, sp, cp;
      isxo_sincos(delta, &sd, &cd);
      isxo_sincos2pi(u3, &sp, &cp);
      v3 e = { fma(cp, A.x, sp * Bv.x), fma(cp, A.y, sp * Bv.y), fma(cp, A.z, sp * Bv.z) };
      M.x = fma(cd, n.x, sd * e.x); M.y = fma(cd, n.y, sd * e.y); M.z = fma(cd, n.z, sd * e.z);
    }
    double d2 = -2.0 * dot3(*v, M);
    w = axpy(d2, M, *v);
  }
  /* never leave into the wall: mirror the direction back to the free side */
  double dn = dot3(w, n);
  if (dn <= 0.0) w = axpy(-2.0 * dn, n, w);
  /* One Newton step towards unit length: w *= (3 - |w|^2)/2.  Without it the loop
   * |v|-error -> hit point off the sphere -> normal not unit -> |v|-error is a multiplicative random
   * walk: 3e-4 of the rays at rho=.99 ended with | |v|^2-1 | up to 7e-5, and with rho=1 (10 000-point
   * limit) it is unbounded.  The step squares the error, so it stays at rounding level. */
  double k = fma(-0.5, dot3(w, w), 1.5);
  w.x *= k; w.y *= k; w.z *= k;
  *v = w;
  return 1;
}

typedef struct { int status, npts, on; v3 p, v, prev; uint64_t wall_hits; } endstate;  /* prev: start of the last segment */

/* ISX_TRACE_CHORD: Lambertian bounce off the inner sphere.  For cosine-law emission from a point of a
 * sphere the far intersection is uniformly distributed over the sphere (the integrating-sphere identity:
 * the form factor between two surface elements of a sphere does not depend on where they are), so the next
 * wall point T is sampled directly.  Same Philox words as the explicit bounce: a -> z, b -> absorb + azimuth.
 * Returns 0 if absorbed. */
static int interact_chord(const geom* g, v3* T, uint64_t seed, uint64_t ray, uint32_t j, uint32_t stream) {
  uint32_t wl[4];
  draw_block(seed, ray, j >> 1, stream, wl);
  const uint32_t wa = wl[2u * (j & 1u)], wb = wl[2u * (j & 1u) + 1u];
  if (!((uint64_t)wb < g->rho_thr)) return 0;
  double zz = fma(-0x1.0p-31, (double)wa, 1.0 - 0x1.0p-32);   /* = 1 - 2 (wa + 1/2) 2^-32, exact */
  double s2 = sqrt(fma(-zz, zz, 1.0));
  double sf, cf;
  circle_point_psi(fma((double)wb, g->psi_k1, g->psi_k0), &cf, &sf);
  double rxy = g->r_in * s2;
  T->x = rxy * cf; T->y = rxy * sf; T->z = g->r_in * zz;
  return 1;
}

static void trace_one(const geom* g, uint64_t seed, uint64_t ray, uint32_t stream, v3 p, v3 v, int on, endstate* out) {
  int npts = 1;
  uint32_t j = 0;
  int status;
  int tgt = 0;   /* chord mode: v holds the next wall point T instead of a direction */
  v3 prev = p;
  for (;;) {
    v3 q;
    int kind;
    prev = p;
    if (tgt) {
      v3 d = { v.x - p.x, v.y - p.y, v.z - p.z };  /* chord P -> T */
      tgt = 0;
      if (v.z >= g->zcut_in) {
        kind = K_INNER; q = v; v = d;              /* arrived; v keeps the (unnormalised) last chord */
      } else {                                     /* T lies in the port opening: leave along the chord */
        v = unit_dir(d);
        kind = next_hit_general(g, p, v, on, &q);
      }
    } else {
      kind = next_hit(g, p, &v, on, &q);
    }
    p = q;
    npts++;
    if (kind == K_BOX) { status = ISXO_EXITED; on = K_BOX; break; }
    on = kind;
    int alive;
    if (g->chord && kind == K_INNER && g->lambertian && g->surface_model == 0) {
      alive = interact_chord(g, &v, seed, ray, j, stream);
      tgt = alive;
    } else {
      alive = interact(g, kind, q, &v, seed, ray, j, stream);
    }
    j++;
    if (!alive) { status = ISXO_ABSORBED; break; }
    if (npts > g->limit) { status = ISXO_SUSPENDED; break; }
  }
  out->status = status; out->npts = npts; out->on = on; out->p = p; out->v = v; out->prev = prev; out->wall_hits = j;
}

/* ------------------------------------------------------------------------- */
/* nonLambertianFlux.C BRDF (TVector3 arithmetic restated in its own op order) */
/* ------------------------------------------------------------------------- */
static inline v3 tv_orthogonal(v3 a) { /* TVector3::Orthogonal */
  double xx = a.x < 0.0 ? -a.x : a.x, yy = a.y < 0.0 ? -a.y : a.y, zz = a.z < 0.0 ? -a.z : a.z;
  v3 r;
  if (xx < yy) {
    if (xx < zz) { r.x = 0; r.y = a.z; r.z = -a.y; } else { r.x = a.y; r.y = -a.x; r.z = 0; }
  } else {
    if (yy < zz) { r.x = -a.z; r.y = 0; r.z = a.x; } else { r.x = a.y; r.y = -a.x; r.z = 0; }
  }
  return r;
}
static inline v3 tv_cross(v3 a, v3 p) { /* TVector3::Cross */
  v3 r = { a.y * p.z - p.y * a.z, a.z * p.x - p.z * a.x, a.x * p.y - p.x * a.y };
  return r;
}
static inline double tv_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 tv_unit(v3 a) { /* TVector3::Unit */
  double tot2 = a.x * a.x + a.y * a.y + a.z * a.z;
  double tot = (tot2 > 0) ? 1.0 / sqrt(tot2) : 1.0;
  v3 r = { a.x * tot, a.y * tot, a.z * tot };
  return r;
}
static inline v3 tv_setmag1(v3 a) { /* TVector3::SetMag(1.0) */
  double f = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
  if (f == 0) return a;
  f = 1.0 / f;
  v3 r = { a.x * f, a.y * f, a.z * f };
  return r;
}

static v3 brdf_sample(const isxo_config* c, v3 normal, v3 incident, uint64_t seed, uint64_t ray) {
  uint32_t w[4];
  draw_block(seed, ray, 0u, 1u, w);
  double rough = c->brdf[0], spec = c->brdf[1], diff = c->brdf[2];
  double sum = spec + diff;   /* BRDF ctor nonLambertianFlux.C:157-159 */
  spec /= sum;
  if (isxo_u01(w[0]) < spec) { /* SampleSpecular :172-189 */
    double a = 2 * tv_dot(incident, normal);
    v3 refl = { incident.x - a * normal.x, incident.y - a * normal.y, incident.z - a * normal.z };
    refl = tv_setmag1(refl);
    double Rg = sqrt(-2.0 * isxo_log(isxo_u01(w[1])));
    double s2, c2;
    isxo_sincos2pi(isxo_u01(w[2]), &s2, &c2);
    double theta = (rough * M_PI / 6) * (Rg * c2); /* gRandom->Gaus(0, rough*pi/6) */
    double st, ct, sp, cp;
    isxo_sincos(theta, &st, &ct);
    isxo_sincos2pi(isxo_u01(w[3]), &sp, &cp);    /* gRandom->Uniform(0,2pi) */
    v3 p1 = tv_orthogonal(refl);
    v3 p2 = tv_cross(refl, p1);
    v3 res = { refl.x + st * (cp * p1.x + sp * p2.x), refl.y + st * (cp * p1.y + sp * p2.y),
               refl.z + st * (cp * p1.z + sp * p2.z) };
    return tv_setmag1(res);
  } else { /* SampleDiffuse :191-207; theta = acos(sqrt(u)) => cos = sqrt(u), sin = sqrt(1-u) */
    double u = isxo_u01(w[1]);
    double ct = sqrt(u), st = sqrt(1.0 - u);
    double sp, cp;
    isxo_sincos2pi(isxo_u01(w[3]), &sp, &cp);
    v3 uu = tv_orthogonal(normal);
    v3 vv = tv_cross(normal, uu);
    double x = st * cp, y = st * sp, z = ct;
    v3 res = { x * uu.x + y * vv.x + z * normal.x, x * uu.y + y * vv.y + z * normal.y,
               x * uu.z + y * vv.z + z * normal.z };
    return tv_unit(res);
  }
}

/* full per-ray path for either source model; wall_hits accumulates both traces */
static void trace_ray(const isxo_config* c, const geom* g, uint64_t seed, uint64_t ray, endstate* es) {
  trace_one(g, seed, ray, 0u, g->src, g->dir0, K_NONE, es);
  if (c->source_model == 1) {
    /* nonLambertianFlux.C:253-268: normal = lastPoint.Unit(), incident = INITIAL direction */
    uint64_t wh = es->wall_hits;
    v3 normal = tv_unit(es->p);
    v3 nd = brdf_sample(c, normal, g->dir0, seed, ray);
    /* ARay ctor normalises again */
    double mag = sqrt(nd.x * nd.x + nd.y * nd.y + nd.z * nd.z);
    v3 d = { nd.x / mag, nd.y / mag, nd.z / mag };
    int on = (es->on == K_BOX) ? K_NONE : es->on;
    endstate e2;
    trace_one(g, seed, ray, 2u, es->p, d, on, &e2);
    e2.wall_hits += wh;
    *es = e2;
  }
}

/* ------------------------------------------------------------------------- */
/* detector (fluxAtObserver.C:49-107), restated operation for operation       */
/* ------------------------------------------------------------------------- */
static void det_set_position(double theta, double phi, double radius, double portz, double d[6]) {
  double theta_rad = theta * M_PI / 180.0;
  double phi_rad = phi * M_PI / 180.0;
  /* g++ -O2 (ACLiC) merges sin(a),cos(a) into one sincos(a) call; glibc's sincos differs from
   * separate sin/cos by 1 ulp for a few arguments, so the call is made explicit here. */
  double st, ct, sp, cp;
  sincos(theta_rad, &st, &ct);
  sincos(phi_rad, &sp, &cp);
  double x = radius * st * cp;
  double y = radius * st * sp;
  double z = portz - radius * ct;
  double dx = x - 0;
  double dy = y - 0;
  double dz = z - (portz);
  double mag = sqrt(dx * dx + dy * dy + dz * dz);
  d[0] = x; d[1] = y; d[2] = z;
  d[3] = -dy / mag;
  d[4] = dx / mag;
  d[5] = dz / mag;
}

int isxo_detector_table(const isxo_config* c, double* out) {
  if (!c || !out || c->n_theta < 1 || c->n_phi < 1) return -3;
  for (int i = 0; i < c->n_theta; i++) {
    double theta = (i + 0.5) * 90.0 / c->n_theta;
    for (int j = 0; j < c->n_phi; j++) {
      double phi = (j + 0.5) * 360.0 / c->n_phi;
      det_set_position(theta, phi, c->det_distance, c->exit_port_z, out + 6 * ((size_t)i * c->n_phi + j));
    }
  }
  return 0;
}

int isxo_check_intersection(const double det[6], double width, const double lastPoint[3], const double direction[3]) {
  double x = det[0], y = det[1], z = det[2], nx = det[3], ny = det[4], nz = det[5];
  double dot = direction[0] * nx + direction[1] * ny + direction[2] * nz;
  if (fabs(dot) < 1e-10) return 0;
  double dx = lastPoint[0] - x;
  double dy = lastPoint[1] - y;
  double dz = lastPoint[2] - z;
  double t = -(dx * nx + dy * ny + dz * nz) / dot;
  double ix = lastPoint[0] + direction[0] * t;
  double iy = lastPoint[1] + direction[1] * t;
  double iz = lastPoint[2] + direction[2] * t;
  double rx = ix - x;
  double ry = iy - y;
  double rz = iz - z;
  double ux = ny * rz - nz * ry;
  double uy = nz * rx - nx * rz;
  double uz = nx * ry - ny * rx;
  double r2 = ux * ux + uy * uy + uz * uz;
  return r2 <= (width / 2) * (width / 2);
}

/* ------------------------------------------------------------------------- */
/* drivers                                                                    */
/* ------------------------------------------------------------------------- */
/* ---- hooks for the reference-side bounce dump (tools/ref_dump/dumpBounces.C -> tests/test_robast_dump.py): the pieces of
 * the trace loop one bounce at a time, so that ROBAST's own points and random draws can be replayed through them. */
int isxo_next_boundary(const isxo_config* c, const double p[3], const double v[3], int on, double q_out[3], double v_out[3]) {
  geom g;
  if (prepare(c, &g)) return -1;
  v3 q, P = { p[0], p[1], p[2] }, V = { v[0], v[1], v[2] };
  const int kind = next_hit(&g, P, &V, on, &q);
  q_out[0] = q.x; q_out[1] = q.y; q_out[2] = q.z;
  if (v_out) { v_out[0] = V.x; v_out[1] = V.y; v_out[2] = V.z; }   /* (the unit direction, if the step left rule S1') */
  return kind;
}
/* unit surface normal (towards the free side) of boundary `kind` at q */
int isxo_surface_normal(const isxo_config* c, int kind, const double q[3], double n_out[3]) {
  geom g;
  if (prepare(c, &g)) return -1;
  v3 Q = { q[0], q[1], q[2] };
  const v3 n = surface_normal(&g, kind, Q);
  n_out[0] = n.x; n_out[1] = n.y; n_out[2] = n.z;
  return 0;
}
/* the cosine-law emission of interact() from the point q of surface `kind` for the two Philox words of an interaction (a: z of
 * the sphere point, b: azimuth; b must be a surviving word, b < rho_thr): r_in s - q, un-normalised, on the inner sphere
 * (kind 1), the unit vector of n + s elsewhere */
int isxo_cosine_emission(const isxo_config* c, int kind, const double q[3], uint32_t wa, uint32_t wb, double w_out[3]) {
  geom g;
  if (prepare(c, &g)) return -1;
  v3 Q = { q[0], q[1], q[2] };
  const v3 w = cosine_emission(&g, kind, Q, wa, wb);
  w_out[0] = w.x; w_out[1] = w.y; w_out[2] = w.z;
  return 0;
}

void isxo_default_config(isxo_config* c) {
  memset(c, 0, sizeof(*c));
  c->struct_size = (uint32_t)sizeof(*c);
  c->r_in = 100.1; c->r_out = 101.0; c->theta_max_deg = 170.0;
  c->reflectance = 0.99; c->roughness_rad = 0.01; c->box_half = 300.0;
  c->lambertian = 1; c->max_points = 50000;
  c->src[0] = -60; c->src[1] = 0; c->src[2] = -75;
  c->dir[0] = 5; c->dir[1] = 0; c->dir[2] = 0;
  c->n_theta = 180; c->n_phi = 90;
  c->det_diameter = 40.0; c->det_distance = 100.0; c->exit_port_z = -100.0;
  c->source_model = 0;
  c->brdf[0] = 0.3; c->brdf[1] = 0.4; c->brdf[2] = 0.6;
}

int isxo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* the line handed to Detector::checkIntersection */
static inline void hit_line(const isxo_config* c, const endstate* es, double lp[3], double d[3]) {
  if (c->hit_line_mode == 1) {
    /* fluxAtObserverFast.C:1181-1201: secondLastPoint stays (0,0,0); dir = (last-0)/|last-0|; then the
     * ARay constructor (:1285) normalises the direction once more */
    double dx = es->p.x - 0.0, dy = es->p.y - 0.0, dz = es->p.z - 0.0;
    double mag = sqrt(dx * dx + dy * dy + dz * dz);
    double ex = dx / mag, ey = dy / mag, ez = dz / mag;
    double m2 = sqrt(ex * ex + ey * ey + ez * ez);
    lp[0] = 0.0; lp[1] = 0.0; lp[2] = 0.0;
    d[0] = ex / m2; d[1] = ey / m2; d[2] = ez / m2;
  } else {
    lp[0] = es->p.x; lp[1] = es->p.y; lp[2] = es->p.z;
    d[0] = es->v.x; d[1] = es->v.y; d[2] = es->v.z;
  }
}

static inline void census(const endstate* es, double portz, isxo_stats* st, int* counted) {
  st->launched++;
  st->wall_hits += es->wall_hits;
  *counted = 0;
  if (es->status == ISXO_EXITED) {
    st->exited++;
    /* isRayPassingThroughExitPort (fluxAtObserver.C:162-166), applied to exited rays only
     * (fluxAtObserverOptimize.C:298-327) */
    if (es->p.z < portz) { st->counted_below_z++; *counted = 1; }
  } else if (es->status == ISXO_ABSORBED) st->absorbed++;
  else st->suspended++;
}

static void stats_add(isxo_stats* a, const isxo_stats* b) {
  a->launched += b->launched; a->exited += b->exited; a->counted_below_z += b->counted_below_z;
  a->absorbed += b->absorbed; a->suspended += b->suspended; a->bin_increments += b->bin_increments;
  a->wall_hits += b->wall_hits;
}

int isxo_trace_endstates(const isxo_config* c, uint64_t n, uint64_t seed, uint64_t first, int32_t* status,
                         int32_t* npts, double* lp, double* dir) {
  geom g;
  if (!c) return -3;
  int rc = prepare(c, &g);
  if (rc) return rc;
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t i = 0; i < (int64_t)n; i++) {
    endstate es;
    trace_ray(c, &g, seed, first + (uint64_t)i, &es);
    if (status) status[i] = es.status;
    if (npts) npts[i] = es.npts;
    if (lp) { lp[3 * i] = es.p.x; lp[3 * i + 1] = es.p.y; lp[3 * i + 2] = es.p.z; }
    if (dir) { dir[3 * i] = es.v.x; dir[3 * i + 1] = es.v.y; dir[3 * i + 2] = es.v.z; }
  }
  return 0;
}

int isxo_fluxmap(const isxo_config* c, uint64_t n, uint64_t seed, uint64_t first, uint64_t* hits, isxo_stats* stats,
                 int nthreads) {
  geom g;
  if (!c || !hits) return -3;
  int rc = prepare(c, &g);
  if (rc) return rc;
  if (c->n_theta < 1 || c->n_phi < 1) return -2;
  size_t nb = (size_t)c->n_theta * c->n_phi;
  double* tab = (double*)malloc(nb * 6 * sizeof(double));
  if (!tab) return -3;
  isxo_detector_table(c, tab);
  memset(hits, 0, nb * sizeof(uint64_t));
  isxo_stats tot;
  memset(&tot, 0, sizeof(tot));
  double t0 = now_ms();
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
  else omp_set_num_threads(omp_get_num_procs());
#endif
#pragma omp parallel
  {
    uint64_t* h = (uint64_t*)calloc(nb, sizeof(uint64_t));
    isxo_stats st;
    memset(&st, 0, sizeof(st));
#pragma omp for schedule(dynamic, 64)
    for (int64_t i = 0; i < (int64_t)n; i++) {
      endstate es;
      trace_ray(c, &g, seed, first + (uint64_t)i, &es);
      int counted;
      census(&es, c->exit_port_z, &st, &counted);
      if (counted) {
        double lp[3], d[3];
        hit_line(c, &es, lp, d);
        /* trace-once loop fluxAtObserverFast.C:1269-1294 with the per-position semantics
         * of fluxAtObserverOptimize.C:309 (last point + final direction) */
        for (size_t k = 0; k < nb; k++)
          if (isxo_check_intersection(tab + 6 * k, c->det_diameter, lp, d)) { h[k]++; st.bin_increments++; }
      }
    }
#pragma omp critical
    {
      for (size_t k = 0; k < nb; k++) hits[k] += h[k];
      stats_add(&tot, &st);
    }
    free(h);
  }
  tot.t_kernel_ms = now_ms() - t0;
  if (stats) *stats = tot;
  free(tab);
  return 0;
}

int isxo_fluxmap_per_position(const isxo_config* c, uint64_t rpp, int32_t fold, uint64_t first_group, uint64_t n_groups,
                              uint64_t seed, uint64_t first, uint64_t* hits, isxo_stats* stats, int nthreads) {
  geom g;
  if (!c || !hits || rpp < 1 || (fold != 1 && fold != 2)) return -3;
  int rc = prepare(c, &g);
  if (rc) return rc;
  if (c->n_theta < 1 || c->n_phi < 1 || (fold == 2 && (c->n_phi % 2))) return -2;
  size_t nb = (size_t)c->n_theta * c->n_phi;
  if (first_group + n_groups > nb / (size_t)fold) return -3;
  double* tab = (double*)malloc(nb * 6 * sizeof(double));
  isxo_detector_table(c, tab);
  memset(hits, 0, nb * sizeof(uint64_t));
  isxo_stats tot;
  memset(&tot, 0, sizeof(tot));
  double t0 = now_ms();
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
  else omp_set_num_threads(omp_get_num_procs());
#endif
#pragma omp parallel
  {
    isxo_stats st;
    memset(&st, 0, sizeof(st));
#pragma omp for schedule(dynamic, 4)
    for (int64_t gi = 0; gi < (int64_t)n_groups; gi++) {
      uint64_t grp = first_group + (uint64_t)gi;
      size_t b0, b1 = 0;
      if (fold == 2) {
        int half = c->n_phi / 2;
        b0 = (size_t)(grp / (uint64_t)half) * c->n_phi + (size_t)(grp % (uint64_t)half);
        b1 = b0 + half;
      } else b0 = (size_t)grp;
      uint64_t h0 = 0, h1 = 0;
      for (uint64_t k = 0; k < rpp; k++) { /* traceRaysParallel, fluxAtObserverOptimize.C:281-333 */
        endstate es;
        trace_ray(c, &g, seed, first + grp * rpp + k, &es);
        int counted;
        census(&es, c->exit_port_z, &st, &counted);
        if (counted) {
          double lp[3], d[3];
          hit_line(c, &es, lp, d);
          if (isxo_check_intersection(tab + 6 * b0, c->det_diameter, lp, d)) h0++;
          if (fold == 2 && isxo_check_intersection(tab + 6 * b1, c->det_diameter, lp, d)) h1++;
        }
      }
      hits[b0] = h0;
      if (fold == 2) hits[b1] = h1;
      st.bin_increments += h0 + h1;
    }
#pragma omp critical
    stats_add(&tot, &st);
  }
  tot.t_kernel_ms = now_ms() - t0;
  if (stats) *stats = tot;
  free(tab);
  return 0;
}

int isxo_trace_rays_detector(const isxo_config* c, const double* det, double width, uint64_t n, uint64_t seed,
                             uint64_t first, uint64_t* hit_count, isxo_stats* stats) {
  geom g;
  if (!c || !det || !hit_count) return -3;
  int rc = prepare(c, &g);
  if (rc) return rc;
  isxo_stats st;
  memset(&st, 0, sizeof(st));
  uint64_t h = 0;
  for (uint64_t i = 0; i < n; i++) {
    endstate es;
    trace_ray(c, &g, seed, first + i, &es);
    int counted;
    census(&es, c->exit_port_z, &st, &counted);
    if (counted) {
      double lp[3], d[3];
      hit_line(c, &es, lp, d);
      if (isxo_check_intersection(det, width, lp, d)) h++;
    }
  }
  st.bin_increments = h;
  *hit_count = h;
  if (stats) *stats = st;
  return 0;
}

/* forward segment [0,tmax] of p+t*v enters the tube {|s|<=h, rho<=r} about centre c, unit axis a */
static int segment_hits_tube(v3 p, v3 v, double tmax, const double ca[6], double r, double h) {
  v3 c = { ca[0], ca[1], ca[2] }, a = { ca[3], ca[4], ca[5] };
  v3 w = { p.x - c.x, p.y - c.y, p.z - c.z };
  double ws = dot3(w, a), vs = dot3(v, a);
  /* axial slab */
  double t0 = 0.0, t1 = tmax;
  if (vs != 0.0) {
    double ta = (-h - ws) / vs, tb = (h - ws) / vs;
    if (ta > tb) { double tmp = ta; ta = tb; tb = tmp; }
    if (ta > t0) t0 = ta;
    if (tb < t1) t1 = tb;
  } else if (fabs(ws) > h) return 0;
  if (t0 > t1) return 0;
  /* radial: |w + t v|^2 - (ws + t vs)^2 <= r^2 */
  double A = dot3(v, v) - vs * vs;
  double B = dot3(w, v) - ws * vs;
  double C = dot3(w, w) - ws * ws - r * r;
  if (A <= 0.0) return C <= 0.0;
  double D = fma(B, B, -(A * C));
  if (D < 0.0) return 0;
  double sD = sqrt(D);
  double ra = (-B - sD) / A, rb = (-B + sD) / A;
  if (ra > t0) t0 = ra;
  if (rb < t1) t1 = rb;
  return t0 <= t1;
}

int isxo_disc_sweep(const isxo_config* c, const double* ca, int32_t nd, double radius, double half_thick, uint64_t n,
                    uint64_t seed, uint64_t first, uint64_t* hits, isxo_stats* stats, int nthreads) {
  geom g;
  if (!c || !hits || !ca || nd < 1) return -3;
  int rc = prepare(c, &g);
  if (rc) return rc;
  memset(hits, 0, (size_t)nd * sizeof(uint64_t));
  isxo_stats tot;
  memset(&tot, 0, sizeof(tot));
  double t0 = now_ms();
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
  else omp_set_num_threads(omp_get_num_procs());
#endif
#pragma omp parallel
  {
    uint64_t* h = (uint64_t*)calloc((size_t)nd, sizeof(uint64_t));
    isxo_stats st;
    memset(&st, 0, sizeof(st));
#pragma omp for schedule(dynamic, 64)
    for (int64_t i = 0; i < (int64_t)n; i++) {
      /* the last mirror point and the exit direction define the forward segment */
      endstate es;
      trace_one(&g, seed, first + (uint64_t)i, 0u, g.src, g.dir0, K_NONE, &es);
      const int status = es.status;
      const v3 seg_start = es.prev, p = es.p, v = es.v;
      int counted;
      census(&es, c->exit_port_z, &st, &counted);
      if (status == ISXO_EXITED) {
        v3 dlt = { p.x - seg_start.x, p.y - seg_start.y, p.z - seg_start.z };
        double tmax = dot3(dlt, v);
        for (int k = 0; k < nd; k++)
          if (segment_hits_tube(seg_start, v, tmax, ca + 6 * k, radius, half_thick)) { h[k]++; st.bin_increments++; }
      }
    }
#pragma omp critical
    {
      for (int k = 0; k < nd; k++) hits[k] += h[k];
      stats_add(&tot, &st);
    }
    free(h);
  }
  tot.t_kernel_ms = now_ms() - t0;
  if (stats) *stats = tot;
  return 0;
}

/* the sweep as the reference writes it (integratingSphereDetectorSweep.C:54-77): disc k sees only its own rays
 * [first + k*rpp, +rpp) */
int isxo_disc_sweep_per_position(const isxo_config* c, const double* ca, int32_t nd, double radius, double half_thick,
                                 uint64_t rpp, uint64_t seed, uint64_t first, uint64_t* hits, isxo_stats* stats, int nthreads) {
  geom g;
  if (!c || !hits || !ca || nd < 1 || rpp < 1) return -3;
  int rc = prepare(c, &g);
  if (rc) return rc;
  memset(hits, 0, (size_t)nd * sizeof(uint64_t));
  isxo_stats tot;
  memset(&tot, 0, sizeof(tot));
  double t0 = now_ms();
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
  else omp_set_num_threads(omp_get_num_procs());
#endif
#pragma omp parallel
  {
    isxo_stats st;
    memset(&st, 0, sizeof(st));
#pragma omp for schedule(dynamic, 1)
    for (int k = 0; k < nd; k++) {
      uint64_t h = 0;
      for (uint64_t i = 0; i < rpp; i++) {
        endstate es;
        trace_one(&g, seed, first + (uint64_t)k * rpp + i, 0u, g.src, g.dir0, K_NONE, &es);
        const int status = es.status;
        const v3 seg_start = es.prev, p = es.p, v = es.v;
        int counted;
        census(&es, c->exit_port_z, &st, &counted);
        if (status == ISXO_EXITED) {
          v3 dlt = { p.x - seg_start.x, p.y - seg_start.y, p.z - seg_start.z };
          double tmax = dot3(dlt, v);
          if (segment_hits_tube(seg_start, v, tmax, ca + 6 * k, radius, half_thick)) h++;
        }
      }
      hits[k] = h;
      st.bin_increments += h;
    }
#pragma omp critical
    stats_add(&tot, &st);
  }
  tot.t_kernel_ms = now_ms() - t0;
  if (stats) *stats = tot;
  return 0;
}

int isxo_exit_dz_hist(const isxo_config* c, uint64_t n, uint64_t seed, uint64_t first, int32_t nbins, uint64_t* hist,
                      isxo_stats* stats, int nthreads) {
  geom g;
  if (!c || !hist || nbins < 1) return -3;
  int rc = prepare(c, &g);
  if (rc) return rc;
  memset(hist, 0, (size_t)nbins * sizeof(uint64_t));
  isxo_stats tot;
  memset(&tot, 0, sizeof(tot));
  double t0 = now_ms();
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
  else omp_set_num_threads(omp_get_num_procs());
#endif
#pragma omp parallel
  {
    uint64_t* h = (uint64_t*)calloc((size_t)nbins, sizeof(uint64_t));
    isxo_stats st;
    memset(&st, 0, sizeof(st));
#pragma omp for schedule(dynamic, 256)
    for (int64_t i = 0; i < (int64_t)n; i++) {
      endstate es;
      trace_ray(c, &g, seed, first + (uint64_t)i, &es);
      int counted;
      census(&es, c->exit_port_z, &st, &counted);
      if (counted) {
        /* TH1D("hDirectionZ",100,-1,1)->Fill(dz) (distributionSphereDetectorSweep.C:54,91) */
        double f = (es.v.z + 1.0) * 0.5 * nbins;
        int b = (int)floor(f);
        if (b >= 0 && b < nbins) { h[b]++; st.bin_increments++; }
      }
    }
#pragma omp critical
    {
      for (int k = 0; k < nbins; k++) hist[k] += h[k];
      stats_add(&tot, &st);
    }
    free(h);
  }
  tot.t_kernel_ms = now_ms() - t0;
  if (stats) *stats = tot;
  return 0;
}

int isxo_exit_directions(const isxo_config* c, uint64_t n, uint64_t seed, uint64_t first, uint64_t cap, uint64_t* ids,
                         double* dirs, uint64_t* count) {
  geom g;
  if (!c || !ids || !dirs || !count) return -3;
  int rc = prepare(c, &g);
  if (rc) return rc;
  uint64_t k = 0;
  for (uint64_t i = 0; i < n; i++) {
    endstate es;
    trace_ray(c, &g, seed, first + i, &es);
    if (es.status == ISXO_EXITED && es.p.z < c->exit_port_z) { /* distributionSphereDetectorSweep.C:76-88 */
      if (k < cap) { ids[k] = first + i; dirs[3 * k] = es.v.x; dirs[3 * k + 1] = es.v.y; dirs[3 * k + 2] = es.v.z; }
      k++;
    }
  }
  *count = k;
  return 0;
}
