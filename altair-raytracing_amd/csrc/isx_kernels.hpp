// isx_kernels.hpp — gfx950 kernels: persistent-wave trace loop + wave-cooperative detector binning.
//
// Execution model:
//   * one ray per lane, ray state in VGPRs; every wave owns a contiguous range of ray
//     indices and refills dead lanes from it (no atomics on the refill path), so lanes stay
//     busy although the bounce count per ray is geometric (mean ~57, tail >> mean);
//   * RNG is Philox4x32-10 keyed by (seed) and counted by (ray index, draw block, stream):
//     a ray's history does not depend on which lane/wave/GPU traces it; one block serves two
//     bounces and is computed on the even steps of a loop trip only (bounce_words);
//   * ISX_STEPS bounce steps per loop trip; refill, generic-search flush, census, BRDF re-scatter
//     and the binning of the lines that left are paid once per trip;
//   * the lanes whose rays left through the port prepare their lines themselves (prep_record:
//     line vs the sphere of detector centres, cap, row range), then each line is broadcast and
//     ALL 64 lanes of the wave bin it - lane = detector row, walking the row's phi-window
//     (walk_rows); the cull never decides a result, the exact reference-order test does;
//   * bins live in a per-block LDS histogram (u32[n_theta*n_phi] = 64.8 KB for 180x90),
//     flushed once per block with global 64-bit atomics.
#pragma once
#include <type_traits>
#include "isx_device.hpp"

namespace isx {

struct DetGrid {
  int n_theta, n_phi, nbins, bin_mode;  // bin_mode 0: brute exact, 1: culled + classified
  int hit_line_mode, pad0;              // ISX_HITLINE_*: which line the detector test sees
  double half_w2;                       // (width/2)*(width/2)   fluxAtObserver.C:106
  double rho_d;                         // width/2
  double R;                             // detector distance from (0,0,portz)
  double portz;                         // exitPortZ
  const double* table;                  // [nbins][6]  x,y,z,nx,ny,nz (Detector::setPosition)
  const double* rowtab;                 // [n_theta][4] sin(theta), cos(theta), z_i, A_i=R*sin
  const double* coltab;                 // [n_phi][2]   cos(phi), sin(phi)
  // SINK_DISC: physical discs (integratingSphereDetectorSweep.C:145-172); nbins == n_disc
  const double* discs;                  // [n_disc][6] centre, unit axis
  double disc_r, disc_h;
  // isx_bin_discs_kernel: the discs in spatial order, eight to a cluster -- clusters[k] = centre and radius of a ball that
  // holds the bounding balls of discs 8k..8k+7 of that order (binary32, radius rounded up), disc_perm[j] = the caller's index of
  // the j-th disc of that order (host: cluster_discs, isx_api.hip)
  const float* clusters;                // [n_clusters][4]
  const int* disc_perm;                 // [n_disc]
  int n_clusters, pad1;
  // SINK_DISCPOS: the reference's physical-disc loop as it is written (integratingSphereDetectorSweep.C:54-77): disc g
  // sees only its own rays [map_first + g*rays_per_group, +rays_per_group) -- one launch for all positions.
  // SINK_PERPOS: the reference's per-position maps (fluxAtObserverOptimize.C:542-579): rays
  // [map_first + g*rays_per_group, +rays_per_group) belong to detector group g and are tested
  // against that group's detector(s) only.  fold 1: group g = bin g.  fold 2 ("twofold",
  // fluxAtObserverFast.C:336-408): group g -> bins (i,j) and (i,j+n_phi/2), i=g/(n_phi/2).
  uint64_t map_first, rays_per_group;
  int fold;
  int rec_stage;                        // 1: the launch reserved LDS for the per-lane exit-line records (prep_record)
  // SINK_LOG: un-binned exit log (3dRayLog.txt): records {ray id, dx, dy, dz} = 32 B per counted ray,
  // the one sink with real HBM output.  log_count is the device-side cursor; records beyond log_cap are dropped
  // (the cursor still counts them, so the caller can tell).
  // SINK_REC (two-kernel pipeline of the flux map): the exit line (last point, direction: 48 B) of every counted ray goes to
  // HBM, into the slice of rec_lines that belongs to the wave's own ray range (slot = ray offset of the range + running
  // count: no atomics, no overflow -- a wave cannot have more exits than rays); rec_counts[wave] = lines written.
  double* rec_lines;             // [n rays of the launch][6]
  uint32_t* rec_counts;          // [waves of the launch]
  double* log_rec;               // [log_cap][4]  (id bit-cast into the first double)
  unsigned long long* log_count;
  uint64_t log_cap;
};

// -DISX_DIAG (tuning builds only, never the shipped library): where the binning work goes.
// [0] lines via the per-lane fast path, [1] via bin_culled caps, [2] via the whole-row fallback, [3] skipped (miss),
// [4..6] column-loop iterations (wave level) of those three paths, [7..9] candidates (lane level), [10] split passes,
// [11] row passes
// [16..]: where the trace loop's time and lanes go (tools/diag_trace.py): wave cycles (s_memtime) per region of a loop trip --
// [16] refill, [17] step-0 boundary search, [18] generic-search flush, [19] step-0 interaction, [20] steps 1..N-1, [21] census +
// re-scatter, [22] sink -- then [23] trips, [24] lanes running / [25] parked after the refill, [26] trips after the wave's range
// ran out ("drain"), [27] lanes running in those, [28] rays ended, [29] flushes, [30] lanes flushed, [31] lanes refilled
#ifdef ISX_DIAG
__device__ unsigned long long g_diag[48];   // [32..47]: the assist wave of assist_body (tools/diag_trace.py)
#define ISX_DIAG_ADD(k, v) do { if (lane == 0) atomicAdd(&g_diag[k], (unsigned long long)(v)); } while (0)
#define ISX_DIAG_ADD_LANES(k, v) atomicAdd(&g_diag[k], (unsigned long long)(v))
#define ISX_TD_DECL unsigned long long td_[16] = {0}; unsigned long long tdc_ = clock64()
#define ISX_TD_MARK(k) do { const unsigned long long c_ = clock64(); td_[k] += c_ - tdc_; tdc_ = c_; } while (0)
#define ISX_TD_ADD(k, v) do { td_[k] += (unsigned long long)(v); } while (0)
#define ISX_TD_FLUSH() do { if (lane == 0) for (int k_ = 0; k_ < 16; ++k_) atomicAdd(&g_diag[16 + k_], td_[k_]); } while (0)
#define ISX_TD_FLUSH_AT(b_) do { if (lane == 0) for (int k_ = 0; k_ < 16; ++k_) atomicAdd(&g_diag[(b_) + k_], td_[k_]); } while (0)
// binning kernel with slot queues: wave cycles since the previous mark go to region k (the timestamp lives in the two unused
// counter words of the wave's SlotQueues): [16] batch preparation, [17] producers (owner search, windows), [18] push,
// [19] pop + line fetch + coefficients, [20] column walk, [21] unit bookkeeping
#define ISX_BD_MARK(sq, k) do { if (lane == 0) { const unsigned long long c_ = clock64(); \
    const unsigned long long p_ = ((unsigned long long)(unsigned)(sq).head[7] << 32) | (unsigned)(sq).tail[7]; \
    atomicAdd(&g_diag[16 + (k)], c_ - p_); (sq).tail[7] = (int)(unsigned)c_; (sq).head[7] = (int)(unsigned)(c_ >> 32); } } while (0)
#define ISX_BD_INIT(sq) do { if (lane == 0) { const unsigned long long c_ = clock64(); (sq).tail[7] = (int)(unsigned)c_; (sq).head[7] = (int)(unsigned)(c_ >> 32); } } while (0)
#else
#define ISX_BD_INIT(sq) do { } while (0)
#define ISX_BD_MARK(sq, k) do { } while (0)
#define ISX_DIAG_ADD(k, v) do { } while (0)
#define ISX_DIAG_ADD_LANES(k, v) do { } while (0)
#define ISX_TD_DECL do { } while (0)
#define ISX_TD_MARK(k) do { } while (0)
#define ISX_TD_ADD(k, v) do { } while (0)
#define ISX_TD_FLUSH() do { } while (0)
#define ISX_TD_FLUSH_AT(b_) do { } while (0)
#endif

enum : int { SINK_FLUX = 0, SINK_DZ = 1, SINK_DISC = 2, SINK_PERPOS = 3, SINK_LOG = 4, SINK_DISCPOS = 5, SINK_REC = 6 };

struct Work {
  uint64_t seed, first, n;    // one launch traces rays [first, first + n), n < 2^31 (a lane keeps a 31-bit offset from `first`)
  unsigned long long* hist;   // [nbins] global accumulators (+=)
  unsigned long long* stats;  // [8]: launched, exited, counted, absorbed, suspended, increments, wall_hits
  // Work queue of the launch (zeroed by the host before it): the persistent waves take rays off ctr[Q_RAYS] (= offset of
  // the next ray nobody has taken) whenever their lanes run dry, so every wave keeps refilling until the LAUNCH has no rays
  // left -- with one fixed slice per wave, a wave spent its last ~45 loop trips (18 % of them at 1200 rays per wave) waiting
  // for its longest rays with a dozen live lanes.  A wave asks for `sub` rays, and for fewer as the queue runs out (1/(2 W)
  // of what was left at its last visit, W = waves of the launch, at least 64): the waves then finish close together.
  // ctr[Q_REGIONS]: exit-line regions handed out (SINK_REC), ctr[Q_BIN]: quarter regions taken by the binning kernel.
  uint32_t* ctr;
  uint32_t sub, pad;          // pad (binning kernels): work units are (quarter regions) >> pad -- 0, or 2 for small launches (64-line units)
};
enum : int { Q_RAYS = 0 /* 64-bit: words 0-1 (it keeps counting after the last ray) */, Q_REGIONS = 2, Q_BIN = 3, Q_WORDS = 4 };
// SINK_REC: a wave appends its exit lines to a private REGION of kRegion slots of the workspace and reserves the next one
// (one atomic on ctr[Q_REGIONS]) when a trip's lines no longer fit; rec_counts[region] = lines in it.  A region is closed with
// at least kRegion - 63 lines unless it is a wave's last, so a launch of n rays on W waves needs at most
// n / (kRegion - 63) + W + 1 regions (isx_api.hip: ensure_pipeline).
constexpr uint32_t kRegion = 1024;

#ifndef ISX_BLOCK
#define ISX_BLOCK 1024
#endif
#ifndef ISX_WAVES_PER_EU
#define ISX_WAVES_PER_EU 0   // 0: let the compiler choose
#endif
constexpr int kBlock = ISX_BLOCK;
#ifndef ISX_ABL
#define ISX_ABL 0
#endif
#ifndef ISX_WALK4
#define ISX_WALK4 1
#endif
#ifndef ISX_ATOM_BRANCH
#define ISX_ATOM_BRANCH 0
#endif
#ifndef ISX_ASSIST_MIN
#define ISX_ASSIST_MIN 64     // the assist wave waits for this many queued rays ... (round 3, trace kernel of 5e7 rays: 16: 11.69 ms, 32: 11.50, 48: 11.34,
                              // 56: 11.44, 64: 11.48; round 4, after the bounce got 10 % shorter: 40: 10.40, 48: 10.36, 64: 10.32)
#endif
#ifndef ISX_ASSIST_LAZY
#define ISX_ASSIST_LAZY 128   // ... for at most this many polls (1024: 12.5 ms)
#endif
#ifndef ISX_STEPS
#define ISX_STEPS 8           // (round 1: 4 / 6 / 8: -3 % / best / -1 %; round 4, with the per-trip bookkeeping of the queues and a shorter
                              //  bounce: 4: 10.74 ms, 6: 10.36, 8: 10.18, 10: 10.20 -- and every other configuration 1-4 % faster at 8)
#endif
constexpr int kStepsPerTrip = ISX_STEPS;   // bounces attempted per trip of the persistent loop
#if ISX_WAVES_PER_EU > 0
#define ISX_KERNEL_ATTR __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(ISX_WAVES_PER_EU, ISX_WAVES_PER_EU)))
#else
#define ISX_KERNEL_ATTR __launch_bounds__(kBlock)
#endif
constexpr int kWavesPerBlock = kBlock / 64;

__device__ __forceinline__ double readlane_f64(double x, int lane) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// ------------------------------------------------------------------ cull helpers (never decide a result)
// raw hardware f32 reciprocal / sqrt (about 1 ulp; the IEEE-exact expansions cost ~10 instructions
// each and buy nothing for a conservative pre-selection)
__device__ __forceinline__ float rcp_cull(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sqrt_cull(float x) { return __builtin_amdgcn_sqrtf(x); }

// atan2 in f32, |error| < 2e-5 rad (checked in tests/test_cull_math.py against numpy)
__device__ __forceinline__ float atan2_cull(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const float a = mx > 0.f ? mn * rcp_cull(mx) : 0.f;
  const float s = a * a;
  float r = fmaf(s, fmaf(s, fmaf(s, fmaf(s, fmaf(s, -0.01172120f, 0.05265332f), -0.11643287f), 0.19354346f), -0.33262347f),
                 0.99997726f) * a;
  if (ay > ax) r = 1.57079637f - r;
  if (x < 0.f) r = 3.14159274f - r;
  if (y < 0.f) r = -r;
  return r;
}

// One column of the detector grid as the binning loop wants it: (cos phi_j, sin phi_j) and the byte offset of
// column (j mod n_phi) inside a histogram row.  The LDS table holds 2*n_phi entries, so a lane walks its
// phi-window with a running pointer and never wraps an index.
struct __align__(16) ColX {
  double c, s;
  uint32_t off4;
  float c32, s32;   // the same cosine / sine rounded to binary32 (fast classifier of walk_columns)
  uint32_t pad;
};

// (LDS-typed pointers: a volatile access through a GENERIC pointer compiles to flat_load / flat_store with system scope, which
//  a wave waits on for hundreds of cycles; through these it is a ds_read / ds_write)
typedef __attribute__((address_space(3))) uint32_t LdsWord;
typedef __attribute__((address_space(3))) int LdsInt;

// acos in f32 for |x| <= 1, |error| < 1e-4 rad (Abramowitz & Stegun 4.4.45 on |x|, reflected for x < 0;
// restated in numpy and checked in tests/test_cull_math.py)
__device__ __forceinline__ float acos_cull(float x) {
  const float ax = fabsf(x);
  const float p = fmaf(ax, fmaf(ax, fmaf(ax, -0.0187293f, 0.0742610f), -0.2121144f), 1.5707288f);
  const float r = sqrt_cull(fmaxf(0.f, 1.f - ax)) * p;
  return x < 0.f ? 3.14159274f - r : r;
}

// ------------------------------------------------------------------ binning of one exit line by a whole wave
// P,V are wave-uniform.  (The number of increments is read off the LDS histogram when the block flushes it.)
template <class DG>
__device__ __forceinline__ void bin_brute(const DG& dd, uint32_t* __restrict__ hist, const V3& P, const V3& V,
                                              int lane) {
  const int nbins = dd.nbins;
  const double* table = dd.table;
  const double half_w2 = dd.half_w2;
  for (int b0 = 0; b0 < nbins; b0 += 64) {
    const int b = b0 + lane;
    bool hit = false;
    if (b < nbins) hit = check_intersection(table + 6 * (size_t)b, half_w2, P, V);
    if (hit) atomicAdd(&hist[b], 1u);
  }
}

// Culled binning, "lane = detector row": every lane owns one theta-row of the cap around a piercing
// point, derives the row's phi-window (cull, f32) and the row's affine coefficients of the hit
// polynomial (f64), then walks its window column by column.  For fixed ray (P,V) and row i
//   dot = V.n, num = (P-c).n, dv = (P-c).V, dd = |P-c|^2
// are all of the form k0 + k1*cos(phi_j) + k2*sin(phi_j)  (c = (A c, A s, z), n = (-S s, S c, -C)),
// so one candidate costs 8 fma + the sign test of  dd*dot^2 - 2*num*dot*dv + num^2 - (w/2)^2*dot^2.
struct CapWin {            // what the cap windows need: the cap around one piercing point
  float Fz, AF2, AF, jf, ch2, inv_dphi;
};

// ---- "box" windows: the general construction (any line; the cap windows above are the cheaper special case of a line that
// passes near O).  A detector centre c that the line X(s) = H + s V (H = foot of O = (0,0,portz) on the line) can hit lies
// within rho_d of it: c = X(s) + e, e perpendicular to V, |e| <= rho.  With m = V_xy/|V_xy| and n = the horizontal unit vector
// perpendicular to it (n.V = 0), and |V| = 1:
//   |e.n| <= rho,   |e.m| <= rho |V_z|,   |e_z| <= rho |V_xy|            (components of unit vectors perpendicular to V)
//   c_z = z_i  =>  s V_z in [z_i - H_z - rho|V_xy|, z_i - H_z + rho|V_xy|]      (the stretch of the line that can reach row i)
//   |c - O| = R =>  s^2 = R^2 - |H-O|^2 - 2 e.(H-O) - |e|^2  in [R^2 - (h+rho)^2, (R+rho)^2 - h^2]     (smin^2, smax^2)
// so in the plane of row i the centre lies in the rectangle  c.n in [dn - rho, dn + rho],  c.m in [Hm + sa|V_xy| - rho|V_z|,
// Hm + sb|V_xy| + rho|V_z|]  (the bounding box of the ellipse in which the rho-tube around the line cuts that plane), and on
// the circle of radius A_i about the z-axis: at most one arc on either side of the direction n, each reduced to the hull of
// its part inside the rectangle -> at most two column windows per row, disjoint by construction (trimmed in whole columns).
// All in binary32 with explicit slack (rho: 0.1 % + 2e-3 cm, angles 1.5e-3 rad, 1e-2 column); restated in numpy and checked
// against the exact test on oracle exit lines in tests/test_cull_math.py.  Never decides a result.
struct BoxLine {
  float smax, smin, vxy, avz, ivz, dn, Hm, Hz, phin, rs, sig;
};

template <class D>
__device__ __forceinline__ bool box_line(const D& d, const V3& P, const V3& V, float inv_dth, BoxLine& b, int& ilo, int& ihi) {
  const double wz = P.z - d.portz;
  const double wv = fma(P.x, V.x, fma(P.y, V.y, wz * V.z));
  const double hx = fma(-wv, V.x, P.x), hy = fma(-wv, V.y, P.y), hz = fma(-wv, V.z, wz);
  const float dO2 = (float)fma(hx, hx, fma(hy, hy, hz * hz));
  const float Rf = (float)d.R, rho = (float)d.rho_d;
  const float dO = sqrt_cull(dO2);
  if (dO - rho > 1.001f * Rf) return false;          // dist(O, line) > R + rho_d: nothing can be hit
  b.rs = fmaf(rho, 1.001f, 2e-3f);
  const float Rr = Rf + b.rs;
  b.smax = sqrt_cull(fmaxf(0.f, fmaf(Rr, Rr, -dO2))) * 1.001f + 1e-2f;
  const float a1 = dO + b.rs;
  b.smin = a1 >= Rf ? 0.f : fmaxf(0.f, sqrt_cull(fmaf(Rf, Rf, -(a1 * a1))) * 0.999f - 1e-2f);
  const float Vx = (float)V.x, Vy = (float)V.y, Vz = (float)V.z;
  const float vxy2 = fmaf(Vx, Vx, Vy * Vy);
  float mx = 1.f, my = 0.f;
  b.vxy = 0.f;
  if (vxy2 > 1e-10f) {
    b.vxy = sqrt_cull(vxy2);
    const float iv = rcp_cull(b.vxy);
    mx = Vx * iv; my = Vy * iv;
  }
  b.avz = fabsf(Vz);
  b.ivz = b.avz > 1e-3f ? rcp_cull(Vz) : 0.f;
  float nx = -my, ny = mx;
  const float Hx = (float)hx, Hy = (float)hy;
  b.Hz = (float)hz;
  float dn = fmaf(Hx, nx, Hy * ny);
  b.sig = -1.f;                                       // phi = phin + sig * psi, psi measured from n towards m
  if (dn < 0.f) { nx = -nx; ny = -ny; dn = -dn; b.sig = 1.f; }
  b.dn = dn;
  b.Hm = fmaf(Hx, mx, Hy * my);
  b.phin = atan2_cull(ny, nx);
  // rows: z_i - O_z = -R cos(theta_i) within dz of H_z over the stretch |s| <= smax
  const float dz = fmaf(b.smax, b.avz, b.rs * b.vxy) + 1e-3f;
  const float iR = rcp_cull(Rf);
  const float clo = fminf(1.f, fmaxf(-1.f, -(b.Hz - dz) * iR)), chi = fminf(1.f, fmaxf(-1.f, -(b.Hz + dz) * iR));
  ilo = max((int)floorf((acos_cull(clo) - 2e-3f) * inv_dth - 0.5f - 1e-2f), 0);
  ihi = min((int)ceilf((acos_cull(chi) + 2e-3f) * inv_dth - 0.5f + 1e-2f), d.n_theta - 1);
  return ihi >= ilo;
}

// the (at most two) column windows of row (zrel0 = z_i - O_z, A = A_i): [j0, j0 + c0) and [j1, j1 + c1), starts in [0, n_phi)
__device__ __forceinline__ void box_window(const BoxLine& b, float zrel0, float A, int n_phi, float inv_dphi, int& j0, int& c0,
                                           int& j1, int& c1) {
  j0 = 0; c0 = 0; j1 = 0; c1 = 0;
  const float kPi = 3.14159274f, kHalfPi = 1.57079637f, dl = 1.5e-3f;
  const float zrel = zrel0 - b.Hz;
  const float rz = b.rs * b.vxy;
  float sa, sb;
  bool ok = true;
  if (b.avz > 1e-3f) {
    const float s1 = (zrel - rz) * b.ivz, s2 = (zrel + rz) * b.ivz;
    sa = fminf(s1, s2); sb = fmaxf(s1, s2);
    const float pad = fmaf(1e-4f, fabsf(sa) + fabsf(sb), 1e-3f);
    sa -= pad; sb += pad;
  } else {                                            // nearly horizontal: the whole stretch inside the shell, or nothing
    ok = fabsf(zrel) <= fmaf(b.smax, b.avz, rz) + 1e-2f;
    sa = -b.smax; sb = b.smax;
  }
  sa = fmaxf(sa, -b.smax); sb = fminf(sb, b.smax);
  ok = ok && sa <= sb && !(sa > -b.smin && sb < b.smin);
  const float em = b.rs * b.avz;
  const float iA = rcp_cull(A);
  const float sl = (fmaf(sa, b.vxy, b.Hm) - em) * iA, sh = (fmaf(sb, b.vxy, b.Hm) + em) * iA;
  const float chi = (b.dn + b.rs) * iA, clo = (b.dn - b.rs) * iA;
  if (!ok || clo > 1.f) return;
  const float psi1 = chi >= 1.f ? 0.f : fmaxf(0.f, acos_cull(chi) - dl);
  const float psi2 = clo <= -1.f ? kPi : fminf(kPi, acos_cull(clo) + dl);
  // hull of { psi in [psi1, psi2] : l <= sin(psi) <= h }
  auto arc = [&](float l, float h, float& lo, float& hi) -> bool {
    const float al = l <= 0.f ? 0.f : (kHalfPi - acos_cull(fminf(l, 1.f))) - dl;
    const float be = h >= 1.f ? kHalfPi : (kHalfPi - acos_cull(fmaxf(h, 0.f))) + dl;
    const float a_lo = fmaxf(al, psi1), a_hi = fminf(be, psi2);
    const float b_lo = fmaxf(kPi - be, psi1), b_hi = fminf(kPi - al, psi2);
    const bool ha = a_lo <= a_hi, hb = b_lo <= b_hi;
    lo = ha ? a_lo : b_lo;
    hi = hb ? b_hi : a_hi;
    return h >= 0.f && l <= 1.f && (ha || hb);
  };
  float p_lo, p_hi, q_lo, q_hi;
  const bool p_ne = arc(sl, sh, p_lo, p_hi), q_ne = arc(-sh, -sl, q_lo, q_hi);
  const bool up = b.sig > 0.f;                        // upper window: phi = phin + psi, lower: phi = phin - psi
  const float u_lo = up ? p_lo : q_lo, u_hi = up ? p_hi : q_hi, l_lo = up ? q_lo : p_lo, l_hi = up ? q_hi : p_hi;
  const bool u_ne = up ? p_ne : q_ne, l_ne = up ? q_ne : p_ne;
  const int jU0 = (int)ceilf(fmaf(b.phin + u_lo, inv_dphi, -0.51f)), jU1 = (int)floorf(fmaf(b.phin + u_hi, inv_dphi, -0.49f));
  const int jL0 = (int)ceilf(fmaf(b.phin - l_hi, inv_dphi, -0.51f)), jL1 = (int)floorf(fmaf(b.phin - l_lo, inv_dphi, -0.49f));
  int cU = u_ne ? max(0, jU1 - jU0 + 1) : 0, cL = l_ne ? max(0, jL1 - jL0 + 1) : 0;
  // no column twice: L ends below U (they meet at psi = 0), U ends below L's next period (they meet at psi = pi)
  if (cU > 0 && cL > 0 && jL0 + cL > jU0) cL = max(0, jU0 - jL0);
  if (cU > 0 && cL > 0 && jU0 + cU > jL0 + n_phi) cU = max(0, jL0 + n_phi - jU0);
  cU = min(cU, n_phi); cL = min(cL, n_phi);
  if (cL > 0 && cU > 0) {
    if (jL0 + cL == jU0) { j0 = jL0; c0 = cL + cU; }                   // adjacent: one window
    else if (jU0 + cU == jL0 + n_phi) { j0 = jU0; c0 = cU + cL; }
    else { j0 = jL0; c0 = cL; j1 = jU0; c1 = cU; }
  } else if (cL > 0) { j0 = jL0; c0 = cL; }
  else { j0 = jU0; c0 = cU; }
  c0 = min(c0, n_phi);
  if (j0 < 0) j0 += n_phi;
  if (j0 < 0) j0 += n_phi;
  if (j0 >= n_phi) j0 -= n_phi;
  if (j0 >= n_phi) j0 -= n_phi;
  if (j1 < 0) j1 += n_phi;
  if (j1 < 0) j1 += n_phi;
  if (j1 >= n_phi) j1 -= n_phi;
  if (j1 >= n_phi) j1 -= n_phi;
}

// Coefficients of row i for the line (P,V) and the walk over columns [jlo + start, jlo + start + len) of its window, with the
// exact decision (shared by every row-to-lane mapping below).  jlo may be negative on entry (wrapped here).
//
// Three tiers per candidate, each handing on only what it cannot decide (the cheap ones never decide wrongly):
//   1. binary32, packed: the four affine forms dot, num, -2dv, dd-(w/2)^2 in two v_pk_fma_f32 pairs and the sign of
//      g = dot (dot (dd-(w/2)^2) - 2 num dv) + num^2  -- evaluated about the point Pq = P + t0 V of the SAME line that is nearest
//      to the centre of the detector sphere (|Pq - c| <= ~125 cm instead of up to 350 from the world box, so the forms are 8x
//      better conditioned).  Error bound: every coefficient, cosine and sine carries a relative rounding u = 2^-24 and a form is
//      two fused operations, so |delta form| <= 5u M_form with M_form = |k0| + |k1| + |k2| >= max |form| over the row; through
//      dg = (2 dot ddw + num dv') d(dot) + dot^2 d(ddw) + (dot dv' + 2 num) d(num) + dot num d(dv')  plus the three roundings of
//      the evaluation of g itself:  |delta g| <= (15 + 3) u S,  S = Md (Md Mf + Mn Mv) + Mn^2  (one number per row).
//      18u = 1.07e-6; the band is 1.3e-6 S.  |g32| > band32 decides; about 1e-3 of the candidates do not get that far.
//   2. binary64: the same sign with the 2.1e-9 relative band of round 1 (coefficients derived again from the row constants --
//      they are no longer resident, which frees 18 VGPRs);
//   3. the reference's own operation sequence (check_intersection) from the detector table.
// The reference rejects |dot| < 1e-10 before anything else.  Tier 1 needs no test of its own for that: outside the band the sign
// of g32 is the sign of the exact g (the bound above does not depend on dot); with |dot| < 1e-10 a NEGATIVE exact g needs
// |num| < 2e-10 Mv, hence |g| < 1e-15, and |g32| <= |g| + 18u S exceeds the band 1.3e-6 S only if S < 4.3e-9 -- rows with
// S < 1e-6 are not decided by tier 1 at all (band = inf) -- so tier 1 never says "hit" there, and "miss" is the reference's
// answer.  For 1e-10 <= |dot| the reference's own float evaluation has the sign of the exact g whenever |g| > 1e-15 (dd dot^2 +
// num^2), nine orders of magnitude inside the band.  (Tier 2 keeps its |dot| < 1e-4 guard.)  One compare and one register
// less per column step; a -DISX_DIAG build re-checks every tier-1 decision against the reference-order test.
typedef float isx_f2 __attribute__((ext_vector_type(2)));
template <class D>
__device__ __forceinline__ void walk_columns(const D& d, uint32_t* __restrict__ hist, const ColX* __restrict__ colx, const V3& P0,
                                             const V3& V0, double t0, int lane, int i, const double* __restrict__ rowt, int jlo,
                                             int start, int len, int path, const double* refetch = nullptr) {
  // refetch (packed walk of the binning kernel: per-lane lines): the line is read again from the workspace by the rare
  // binary64 / reference-order tiers instead of living in 12 VGPRs through the column loop
  const V3 &P = P0, &V = V0;
  isx_f2 k0a = {0.f, 0.f}, k1a = {0.f, 0.f}, k2a = {0.f, 0.f};   // (dot, num)
  isx_f2 k0b = {0.f, 0.f}, k1b = {0.f, 0.f}, k2b = {0.f, 0.f};   // (-2 dv, dd - (w/2)^2)
  float band32 = 0.f;
  if (len > 0) {
    const double Sd = rowt[4 * i + 0], Cd = rowt[4 * i + 1], zd = rowt[4 * i + 2], Ad = rowt[4 * i + 3];
    const double qx = fma(t0, V.x, P.x), qy = fma(t0, V.y, P.y), qz = fma(t0, V.z, P.z);   // Pq: same line, nearest to O
    const double pz = qz - zd;
    const double a0 = -(Cd * V.z), a1 = Sd * V.y, a2 = -(Sd * V.x);
    const double b0 = -(Cd * pz), b1 = Sd * qy, b2 = -(Sd * qx);
    const double e0 = -2.0 * fma(qx, V.x, fma(qy, V.y, pz * V.z)), e1 = 2.0 * (Ad * V.x), e2 = 2.0 * (Ad * V.y);
    const double f0 = fma(qx, qx, fma(qy, qy, fma(Ad, Ad, pz * pz))) - d.half_w2, f1 = -2.0 * (Ad * qx), f2 = -2.0 * (Ad * qy);
    k0a.x = (float)a0; k1a.x = (float)a1; k2a.x = (float)a2;
    k0a.y = (float)b0; k1a.y = (float)b1; k2a.y = (float)b2;
    k0b.x = (float)e0; k1b.x = (float)e1; k2b.x = (float)e2;
    k0b.y = (float)f0; k1b.y = (float)f1; k2b.y = (float)f2;
    const double Md = fabs(a0) + (fabs(a1) + fabs(a2)), Mn = fabs(b0) + (fabs(b1) + fabs(b2));
    const double Mv = fabs(e0) + (fabs(e1) + fabs(e2)), Mf = fabs(f0) + (fabs(f1) + fabs(f2));
    const double S = fma(Md, fma(Md, Mf, Mn * Mv), Mn * Mn);
    // (a row whose magnitudes all vanish -- the line through the row's own centre point on the axis -- is left to the binary64 tier)
    band32 = S >= 1e-6 ? (float)(1.3e-6 * S) * 1.000001f + 1e-30f : __builtin_inff();
    if (jlo < 0) jlo += d.n_phi;   // start column in [0, n_phi); the window then runs to < 2 n_phi
  }
  const ColX* cp = colx + (jlo + start);
  const uint32_t rowoff = (uint32_t)(i * d.n_phi) * 4u;
  // the row's bins through an LDS-typed pointer: one add per column step (row base + column offset), no generic address
  typedef __attribute__((address_space(3))) uint32_t LdsU32;
  typedef __attribute__((address_space(3))) unsigned char LdsByte;
  LdsByte* const rowbins = reinterpret_cast<LdsByte*>((__attribute__((address_space(3))) void*)hist) + rowoff;
  if (len > 0) ISX_DIAG_ADD_LANES(7 + path, len);
  for (int k = 0;; ++k) {                    // until the widest (part of a) window of the wave is done
    const bool act = k < len;
    if (__ballot(act) == 0ull) break;
    ISX_DIAG_ADD(4 + path, 1);
    if (act) {
      bool hit = false;
      const float c32 = cp->c32, s32 = cp->s32;
      const uint32_t o4 = cp->off4;
      LdsByte* const bin = rowbins + o4;
      const isx_f2 cc = {c32, c32}, ss = {s32, s32};
      const isx_f2 ta = __builtin_elementwise_fma(k1a, cc, __builtin_elementwise_fma(k2a, ss, k0a));   // (dot, num)
      const isx_f2 tb = __builtin_elementwise_fma(k1b, cc, __builtin_elementwise_fma(k2b, ss, k0b));   // (-2dv, ddw)
      const float g = fmaf(ta.x, fmaf(ta.x, tb.y, ta.y * tb.x), ta.y * ta.y);
      hit = g < 0.f;
      if (!(fabsf(g) > band32)) {
        // tier 2: binary64 about the original point, exactly the test of round 1
        ISX_DIAG_ADD_LANES(12, 1);
        const double cph = cp->c, sph = cp->s;
        // (the row constants are read again here, from the LDS row table and behind a compiler barrier: otherwise the twelve
        //  binary64 coefficients below are hoisted out of the column loop as loop invariants and the kernel spills 59 VGPRs
        //  to keep them next to the binary32 ones)
        int ir = i;
        asm volatile("" : "+v"(ir));
        const double sd = rowt[4 * ir + 0], cd = rowt[4 * ir + 1], zz = rowt[4 * ir + 2], ad = rowt[4 * ir + 3];
        V3 P, V;
        if (refetch != nullptr) {
          load_line(refetch + (ir - i), P, V);   // (ir - i = 0, behind the barrier: no CSE with the first read)
        } else { P = P0; V = V0; }
        const double pz = P.z - zz;
        const double dot = fma(sd * V.y, cph, fma(-(sd * V.x), sph, -(cd * V.z)));
        const double num = fma(sd * P.y, cph, fma(-(sd * P.x), sph, -(cd * pz)));
        const double m2dv = fma(2.0 * (ad * V.x), cph, fma(2.0 * (ad * V.y), sph, -2.0 * fma(P.x, V.x, fma(P.y, V.y, pz * V.z))));
        const double f0 = fma(P.x, P.x, fma(P.y, P.y, fma(ad, ad, pz * pz)));
        const double f1c = -2.0 * (ad * P.x), f2c = -2.0 * (ad * P.y);
        const double ddw = fma(f1c, cph, fma(f2c, sph, f0 - d.half_w2));
        // sign of  dot^2 (dd - (w/2)^2) - 2 num dot dv + num^2  (|V| = 1 to rounding: Newton-renormalised);
        // evaluation error ~1e-15 of the terms' scale, the band is 2.1e-9 (2 dd_max + (w/2)^2) >= 2e-9 of that scale:
        // |dot| <= 1, num^2 <= dd (Cauchy-Schwarz), dd <= f0 + |f1| + |f2|
        const double diff = fma(dot, fma(dot, ddw, num * m2dv), num * num);
        const double bandc = 2.1e-9 * fma(2.0, f0 + (fabs(f1c) + fabs(f2c)), d.half_w2);
        hit = diff < 0.0;
        if (fabs(dot) < 1e-4 || fabs(diff) <= bandc) {   // tier 3: too close to call, exact reference-order test
          ISX_DIAG_ADD_LANES(13, 1);
          // (the table pointer passes a compiler barrier here: as a loop invariant its VGPR copy for the flat load
          //  would be made at kernel start and kept -- or spilled -- through everything)
          const double* tab = d.table;
          asm volatile("" : "+v"(tab));
          hit = check_intersection(tab + 6 * (size_t)((uint32_t)(bin - reinterpret_cast<LdsByte*>((__attribute__((address_space(3))) void*)hist)) >> 2), d.half_w2, P, V);
        }
      }
#ifdef ISX_DIAG
      {   // tuning builds: a decision taken by tier 1 must be the reference's
        const bool ref = check_intersection(d.table + 6 * (size_t)((uint32_t)(bin - reinterpret_cast<LdsByte*>((__attribute__((address_space(3))) void*)hist)) >> 2), d.half_w2, P, V);
        if (ref != hit) ISX_DIAG_ADD_LANES(14, 1);
      }
#endif
      cp++;
      // (inside the active branch, through an LDS-typed pointer: no zero-initialised offset to carry out of it, no generic
      //  address to rebuild -- two VALU instructions less per column step)
      if (hit) __hip_atomic_fetch_add(reinterpret_cast<LdsU32*>(bin), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

// Two columns per walk step (isx_bin_slots_kernel): entry b of the table holds columns b and b+1 (mod n_phi) side by side, so the
// four forms and g of BOTH candidates are packed binary32 operations -- the same expressions in the same order as walk_columns,
// hence the same error bound and band -- 12 v_pk instructions for two candidates where one at a time took 4 + 5 each.  The second
// candidate of the last step of an odd window is masked.  Tiers 2 and 3 as in walk_columns, for whichever of the two needs them.
constexpr int kDeferCapW = 252;   // (the deferred list of a wave: 256 words -- count, three spare, 252 entries)
// Tiers 2 and 3 of the decision for candidate (row ir, column jr) of the line at src6 (walk_columns: binary64 about the original point
// with its 2.1e-9 band, then the reference's own test from the detector table).
template <class D>
__device__ __forceinline__ bool decide_exact(const D& d, const double* __restrict__ rowt, const ColX* __restrict__ colx,
                                             const double* __restrict__ src6, int ir, int jr) {
  const double sd = rowt[4 * ir + 0], cd = rowt[4 * ir + 1], zz = rowt[4 * ir + 2], ad = rowt[4 * ir + 3];
  const double cph = colx[jr].c, sph = colx[jr].s;
  V3 P, V;
  {
    load_line(src6, P, V);
  }
  const double pz = P.z - zz;
  const double dotd = fma(sd * V.y, cph, fma(-(sd * V.x), sph, -(cd * V.z)));
  const double numd = fma(sd * P.y, cph, fma(-(sd * P.x), sph, -(cd * pz)));
  const double m2dv = fma(2.0 * (ad * V.x), cph, fma(2.0 * (ad * V.y), sph, -2.0 * fma(P.x, V.x, fma(P.y, V.y, pz * V.z))));
  const double f0 = fma(P.x, P.x, fma(P.y, P.y, fma(ad, ad, pz * pz)));
  const double f1c = -2.0 * (ad * P.x), f2c = -2.0 * (ad * P.y);
  const double ddwd = fma(f1c, cph, fma(f2c, sph, f0 - d.half_w2));
  const double diff = fma(dotd, fma(dotd, ddwd, numd * m2dv), numd * numd);
  const double bandc = 2.1e-9 * fma(2.0, f0 + (fabs(f1c) + fabs(f2c)), d.half_w2);
  bool hit = diff < 0.0;
  if (fabs(dotd) < 1e-4 || fabs(diff) <= bandc) {
    ISX_DIAG_ADD_LANES(13, 1);
    const double* tab = d.table;
    asm volatile("" : "+v"(tab));
    hit = check_intersection(tab + 6 * (size_t)(ir * d.n_phi + jr), d.half_w2, P, V);
  }
  return hit;
}

struct __align__(16) ColP { float c0, c1, s0, s1; uint32_t off0, off1, pad0, pad1; };

template <class D>
__device__ __forceinline__ void walk_columns_pairs(const D& d, uint32_t* __restrict__ hist, const ColX* __restrict__ colx,
                                                   const ColP* __restrict__ colp, double t0, int i, const double* __restrict__ rowt,
                                                   int jlo, int len, const double* __restrict__ line6, int lane,
                                                   LdsWord* defer = nullptr, int line = 0) {
  float a0f = 0.f, a1f = 0.f, a2f = 0.f, b0f = 0.f, b1f = 0.f, b2f = 0.f, e0f = 0.f, e1f = 0.f, e2f = 0.f, f0f = 0.f, f1f = 0.f, f2f = 0.f;
  float band32 = 0.f;
  if (len > 0) {
    V3 P, V;
    {
      load_line(line6, P, V);
    }
    const double Sd = rowt[4 * i + 0], Cd = rowt[4 * i + 1], zd = rowt[4 * i + 2], Ad = rowt[4 * i + 3];
    const double qx = fma(t0, V.x, P.x), qy = fma(t0, V.y, P.y), qz = fma(t0, V.z, P.z);   // Pq: same line, nearest to O
    const double pz = qz - zd;
    const double a0 = -(Cd * V.z), a1 = Sd * V.y, a2 = -(Sd * V.x);
    const double b0 = -(Cd * pz), b1 = Sd * qy, b2 = -(Sd * qx);
    const double e0 = -2.0 * fma(qx, V.x, fma(qy, V.y, pz * V.z)), e1 = 2.0 * (Ad * V.x), e2 = 2.0 * (Ad * V.y);
    const double f0 = fma(qx, qx, fma(qy, qy, fma(Ad, Ad, pz * pz))) - d.half_w2, f1 = -2.0 * (Ad * qx), f2 = -2.0 * (Ad * qy);
    a0f = (float)a0; a1f = (float)a1; a2f = (float)a2; b0f = (float)b0; b1f = (float)b1; b2f = (float)b2;
    e0f = (float)e0; e1f = (float)e1; e2f = (float)e2; f0f = (float)f0; f1f = (float)f1; f2f = (float)f2;
    const double Md = fabs(a0) + (fabs(a1) + fabs(a2)), Mn = fabs(b0) + (fabs(b1) + fabs(b2));
    const double Mv = fabs(e0) + (fabs(e1) + fabs(e2)), Mf = fabs(f0) + (fabs(f1) + fabs(f2));
    const double S = fma(Md, fma(Md, Mf, Mn * Mv), Mn * Mn);
    band32 = S >= 1e-6 ? (float)(1.3e-6 * S) * 1.000001f + 1e-30f : __builtin_inff();
    if (jlo < 0) jlo += d.n_phi;
  }
  const isx_f2 vA0 = {a0f, a0f}, vA1 = {a1f, a1f}, vA2 = {a2f, a2f}, vB0 = {b0f, b0f}, vB1 = {b1f, b1f}, vB2 = {b2f, b2f},
               vE0 = {e0f, e0f}, vE1 = {e1f, e1f}, vE2 = {e2f, e2f}, vF0 = {f0f, f0f}, vF1 = {f1f, f1f}, vF2 = {f2f, f2f};
  const ColP* cp = colp + jlo;
  const uint32_t rowoff = (uint32_t)(i * d.n_phi) * 4u;
  typedef __attribute__((address_space(3))) uint32_t LdsU32;
  typedef __attribute__((address_space(3))) unsigned char LdsByte;
  LdsByte* const rowbins = reinterpret_cast<LdsByte*>((__attribute__((address_space(3))) void*)hist) + rowoff;
  if (len > 0) ISX_DIAG_ADD_LANES(7, len);
  for (int k = 0;; k += 2) {
    const bool act = k < len;
    if (__ballot(act) == 0ull) break;
    ISX_DIAG_ADD(4, 2);
    if (act) {
      const float4 cs = *reinterpret_cast<const float4*>(cp);
      const uint2 off = *reinterpret_cast<const uint2*>(&cp->off0);
      const bool two = k + 1 < len;
      const isx_f2 CC = {cs.x, cs.y}, SS = {cs.z, cs.w};
      const isx_f2 dot = __builtin_elementwise_fma(vA1, CC, __builtin_elementwise_fma(vA2, SS, vA0));
      const isx_f2 num = __builtin_elementwise_fma(vB1, CC, __builtin_elementwise_fma(vB2, SS, vB0));
      const isx_f2 mdv = __builtin_elementwise_fma(vE1, CC, __builtin_elementwise_fma(vE2, SS, vE0));
      const isx_f2 ddw = __builtin_elementwise_fma(vF1, CC, __builtin_elementwise_fma(vF2, SS, vF0));
      const isx_f2 g = __builtin_elementwise_fma(dot, __builtin_elementwise_fma(dot, ddw, num * mdv), num * num);
      bool hit0 = g.x < 0.f, hit1 = g.y < 0.f;
      if (!(fminf(fabsf(g.x), fabsf(g.y)) > band32)) {
#pragma unroll 1
        for (int t = 0; t < 2; ++t) {
          const float gt = t == 0 ? g.x : g.y;
          if (fabsf(gt) > band32 || (t == 1 && !two)) continue;
          ISX_DIAG_ADD_LANES(12, 1);
          int ir = i;
          asm volatile("" : "+v"(ir));
          const uint32_t o4 = t == 0 ? off.x : off.y;
          const int jc = (int)(o4 >> 2);
          // put aside for flush_deferred (a miss here), or -- no list (the kernels that walk one line at a time), list full,
          // tuning build -- decided on the spot
          bool hit = false;
          typedef __attribute__((address_space(3))) uint32_t LdsCnt;
#ifdef ISX_DIAG
          const uint32_t pos = (uint32_t)kDeferCapW;
#else
          const uint32_t pos = defer ? __hip_atomic_fetch_add(reinterpret_cast<LdsCnt*>(defer), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                                     : (uint32_t)kDeferCapW;
#endif
          if (pos < (uint32_t)kDeferCapW) ((volatile LdsWord*)defer)[4 + pos] = (uint32_t)line | ((uint32_t)jc << 8) | ((uint32_t)ir << 16);
          else hit = decide_exact(d, rowt, colx, line6 + (ir - i), ir, jc);
          if (t == 0) hit0 = hit; else hit1 = hit;
        }
      }
      hit1 = hit1 && two;
#ifdef ISX_DIAG
      {   // tuning builds: a decision taken by tier 1 must be the reference's
        V3 P, V;
        load_line(line6, P, V);
        for (int t = 0; t < (two ? 2 : 1); ++t) {
          const int jc = (int)((t == 0 ? off.x : off.y) >> 2);
          const bool ref = check_intersection(d.table + 6 * (size_t)(i * d.n_phi + jc), d.half_w2, P, V);
          if (ref != (t == 0 ? hit0 : hit1)) ISX_DIAG_ADD_LANES(14, 1);
        }
      }
#endif
      cp += 2;
      // (the column-slot walk adds 0 or 1 without a branch; here, on the BRDF source, that was measured slower -- 50.8 -> 51.4 ms:
      //  this kernel is VALU-bound and two thirds of its candidates are misses)
      if (hit0) __hip_atomic_fetch_add(reinterpret_cast<LdsU32*>(rowbins + off.x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (hit1) __hip_atomic_fetch_add(reinterpret_cast<LdsU32*>(rowbins + off.y), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

// angular radius of a cap of chord `ch` on the sphere of radius R: 2 asin(ch / 2R), padded (acos_cull: 1e-4 rad).  (Until
// round 2 this was ch/R * 1.01 + 2e-3, which is below 2 asin(ch/2R) once ch > 0.55 R: detectors with rho_d > R/2 lost rows --
// found by tools/soak_cull.py on 300 random geometries, 7 of them with rho_d/R >= 0.53; no BASELINE configuration has
// rho_d/R above 0.2.)
__device__ __forceinline__ float cap_angle(float ch, float iR) {
  const float x = fminf(1.f, 0.5f * ch * iR);
  return 2.0f * (1.57079637f - acos_cull(x)) + 3e-3f;
}

// the column window of row (z_i, A_i) inside the cap `w` (the start may be negative: walk_columns wraps it)
__device__ __forceinline__ void cap_window(const CapWin& w, float zi, float Ai, int n_phi, int& jlo, int& cnt) {
  const float dzi = zi - w.Fz;
  const float num = fmaf(Ai, Ai, fmaf(dzi, dzi, w.AF2)) - w.ch2;
  const float den = 2.0f * Ai * w.AF;
  const float slack = 2e-5f * (fmaf(Ai, Ai, w.AF2) + w.ch2);  // f32 rounding of num
  if (num - slack <= -den) { jlo = 0; cnt = n_phi; }
  else if (num - slack > den) { jlo = 0; cnt = 0; }
  else {
    float K = (num - slack) * rcp_cull(den) - 2e-5f;
    K = fminf(1.f, fmaxf(-1.f, K));
    const float dl = acos_cull(K) + 1e-3f;
    const float hw = dl * w.inv_dphi;
    const int lo = (int)ceilf(w.jf - hw), hi = (int)floorf(w.jf + hw);
    jlo = lo; cnt = hi - lo + 1;
    if (cnt < 0) cnt = 0;
    if (cnt >= n_phi) { jlo = 0; cnt = n_phi; }
  }
}

// Rows ilo..ihi, 64 at a time (lane = row).  MODE_CAP: each row's phi-window is its intersection with the cap `w`;
// MODE_BOX: the (at most two) box windows of the row for the line `bx` -- the second windows, where any row has one, in a
// second sweep over the same rows.  Then the column walk with the exact decision.
//
// Long windows (`split` = 64 ints of wave-private LDS): a column pass lasts as long as the widest window of the wave, and
// the widths are heavy-tailed -- the rows next to the pole of the detector hemisphere span the whole ring (n_phi columns)
// while a typical row has ~8: 4 % of the headline's exit lines owned 27 % of the column iterations.  So when a wave meets
// windows longer than kSplitAt + 16 columns and at most 32 rows
// have more than kSplitAt, pass 0 stops every window at kSplitAt and a pass 1 deals the remainders to Q = 2, 4 or 8 lanes per
// long row (lane -> (row, part) through the LDS list; the row's window and coefficients are simply derived again by its new
// lanes, from wave-uniform inputs, so they are the same numbers).  Same candidates, same decisions, fewer idle lanes.
#ifndef ISX_SPLIT_AT
#define ISX_SPLIT_AT 24
#endif
constexpr int kSplitAt = ISX_SPLIT_AT;
constexpr int MODE_CAP = 1, MODE_BOX = 2;
template <int MODE, class D>
__device__ __forceinline__ void walk_rows(const D& d, uint32_t* __restrict__ hist, const double* __restrict__ rowt,
                                          const ColX* __restrict__ colx, const V3& P, const V3& V, int lane, int ilo,
                                          int ihi, const CapWin& w, const BoxLine& bx, LdsInt* split, int path) {
  // parameter of the point of the line nearest to O = (0,0,portz), the centre of the detector sphere (walk_columns)
  const double t0 = -fma(P.x, V.x, fma(P.y, V.y, (P.z - d.portz) * V.z));
#pragma unroll 1
  for (int i0 = ilo; i0 <= ihi; i0 += 64) {
    int nsweep = 1;
#pragma unroll 1
    for (int sweep = 0; sweep < nsweep; ++sweep) {
      int npass = 1, logq = 0, nlong = 0;
#pragma unroll 1
      for (int pass = 0; pass < npass; ++pass) {
        int i = i0 + lane, part = 0;
        bool have = i <= ihi;
        if (pass == 1) {
          const int slot = lane >> logq;
          part = lane & ((1 << logq) - 1);
          have = slot < nlong;
          i = have ? ((volatile LdsInt*)split)[slot] : 0;
        }
        int jlo = 0, cnt = 0;
        bool second = false;
        if (have) {
          const float zi = (float)rowt[4 * i + 2], Ai = (float)rowt[4 * i + 3];
          if (MODE == MODE_BOX) {
            int j0, c0, j1, c1;
            box_window(bx, zi - (float)d.portz, Ai, d.n_phi, w.inv_dphi, j0, c0, j1, c1);
            second = c1 > 0;
            jlo = sweep == 0 ? j0 : j1;
            cnt = sweep == 0 ? c0 : c1;
          } else {
            cap_window(w, zi, Ai, d.n_phi, jlo, cnt);
          }
        }
        if (MODE == MODE_BOX && sweep == 0 && pass == 0 && __ballot(second) != 0ull) nsweep = 2;
        // the part of the window this lane walks in this pass: [start, start + len)
        int start = 0, len = cnt;
        if (split == nullptr) {
          // (kernels without the wave-private list: whole windows, one pass)
        } else if (pass == 0) {
          const unsigned long long lm = __ballot(cnt > kSplitAt);
          if (lm != 0ull && __ballot(cnt >= kSplitAt + 16) != 0ull) {
            nlong = (int)__popcll(lm);
            logq = nlong <= 8 ? 3 : (nlong <= 16 ? 2 : (nlong <= 32 ? 1 : 0));
            if (logq > 0) {
              npass = 2;
              if (cnt > kSplitAt) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));
                ((volatile LdsInt*)split)[rank] = i;
                len = kSplitAt;
              }
              __builtin_amdgcn_wave_barrier();
            }
          }
        } else {
          const int rem = cnt - kSplitAt;                       // > 0: the row was listed because cnt > kSplitAt
          const int chunk = (rem + (1 << logq) - 1) >> logq;
          start = kSplitAt + part * chunk;
          len = rem - part * chunk;
          len = len < 0 ? 0 : (len > chunk ? chunk : len);
          if (!have) len = 0;
        }
        ISX_DIAG_ADD(11, 1); if (pass == 1) ISX_DIAG_ADD(10, 1);
        walk_columns(d, hist, colx, P, V, t0, lane, i, rowt, jlo, start, len, path);
      }
    }
  }
}

// The wave-uniform part of bin_culled (line vs S(O,R), the lower piercing point, its cap, the row range) evaluated
// PER LANE by the lanes whose rays just left: once per loop trip for all of them instead of once per exit line.  Covers
// the normal case only -- cap construction valid, the upper piercing point's cap above every detector row; anything
// else (mode 1) goes through bin_culled as before.  Same formulas, same margins as bin_culled.
struct RecPre {
  float Fz, AF, jf, ch2;
  int rows;   // ilo | ihi << 16 (fast path: cap around the lower piercing point only), -1: general path, -2: cannot hit anything
};
struct GridConst { float Rf, rho, portz, inv_dphi, inv_dth; int n_theta; };   // wave-uniform, read once per trip
__device__ __forceinline__ RecPre prep_record(const GridConst& k, const V3& P, const V3& V) {
  RecPre o;
  o.Fz = 0.f; o.AF = 0.f; o.jf = 0.f; o.ch2 = 0.f; o.rows = -1;
  const double wz = P.z - (double)k.portz;
  const double wv = fma(P.x, V.x, fma(P.y, V.y, wz * V.z));
  const double hx = fma(-wv, V.x, P.x), hy = fma(-wv, V.y, P.y), hz = fma(-wv, V.z, wz);
  const float dO2 = (float)fma(hx, hx, fma(hy, hy, hz * hz));
  const float R2 = k.Rf * k.Rf;
  const float dO = sqrt_cull(dO2);
  const float a1 = dO + k.rho;
  if (dO - k.rho > 1.001f * k.Rf) { o.rows = -2; return o; }   // farther than R + rho_d from O (box_line's own test)
  if (!(a1 < 0.999f * k.Rf)) return o;
  const float sF = sqrt_cull(R2 - dO2);
  const float smin = sqrt_cull(R2 - a1 * a1);
  const float a0 = fmaxf(0.f, dO - k.rho);
  const float smax = sqrt_cull(R2 - a0 * a0);
  const float ext = fmaxf(sF - smin, smax - sF);
  const float ch2 = fmaf(ext, ext, k.rho * k.rho) * 1.0001f + 1e-3f;
  if (!(4.0f * (R2 - dO2) > 4.04f * ch2)) return o;
  const float ch = sqrt_cull(ch2);
  const float omega = cap_angle(ch, rcp_cull(k.Rf));
  const double s0 = (double)sF - wv, s1 = -(double)sF - wv;
  const float Fz0 = (float)fma(s0, V.z, P.z), Fz1 = (float)fma(s1, V.z, P.z);
  // fast path: the cap of side 0 reaches detector rows, the cap of side 1 lies above all of them
  if (Fz0 - ch > k.portz || !(Fz1 - ch > k.portz)) return o;
  const float Fx = (float)fma(s0, V.x, P.x), Fy = (float)fma(s0, V.y, P.y);
  const float AF2 = fmaf(Fx, Fx, Fy * Fy);
  const float AF = sqrt_cull(AF2);
  float phiF = atan2_cull(Fy, Fx);
  if (phiF < 0.f) phiF += 6.28318530718f;
  const float thF = atan2_cull(AF, k.portz - Fz0);
  const int ilo = max((int)floorf((thF - omega) * k.inv_dth - 0.5f - 1e-3f), 0);
  const int ihi = min((int)ceilf((thF + omega) * k.inv_dth - 0.5f + 1e-3f), k.n_theta - 1);
  o.Fz = Fz0; o.AF = AF; o.jf = phiF * k.inv_dphi - 0.5f; o.ch2 = ch2;
  if (ihi >= ilo && k.n_theta <= 32767) o.rows = ilo | (ihi << 16);
  return o;
}

// CAPS_TOO: lines that pass well inside S(O,R) take cap windows around their two piercing points (cheaper to set up than box
// windows; the binning kernel of the pipeline).  The fused kernels, at their 128-VGPR limit, send every line that is not on
// their fast path through the box windows: one copy less of the column walk, no scratch.  Same candidates' decisions either way.
template <bool CAPS_TOO, class DG>
__device__ inline void bin_culled(const DG& dd, uint32_t* __restrict__ hist,
                                      const double* __restrict__ rowt, const ColX* __restrict__ colx,
                                      const V3 P, const V3 V, int lane, LdsInt* split) {
  // one read of each constant (dd is a volatile LDS copy: nothing of it lives in SGPRs across the trace loop)
  struct { int n_theta, n_phi; double half_w2, rho_d, R, portz; const double* table; } d;
  d.n_theta = dd.n_theta; d.n_phi = dd.n_phi; d.half_w2 = dd.half_w2; d.rho_d = dd.rho_d; d.R = dd.R;
  d.portz = dd.portz; d.table = dd.table;
  // ---- line vs the sphere of detector centres S(O,R), O=(0,0,portz): wave-uniform, f32 is enough (cull only)
  const double wz = P.z - d.portz;
  const double wv = fma(P.x, V.x, fma(P.y, V.y, wz * V.z));
  const double hx = fma(-wv, V.x, P.x), hy = fma(-wv, V.y, P.y), hz = fma(-wv, V.z, wz);
  const float dO2 = (float)fma(hx, hx, fma(hy, hy, hz * hz));
  const float Rf = (float)d.R, rho = (float)d.rho_d;
  const float R2 = Rf * Rf;
  const float dO = sqrt_cull(dO2);
  const float a1 = dO + rho;
  CapWin w;
  w.inv_dphi = (float)d.n_phi * 0.15915494309f;            // 1/dphi
  w.Fz = 0.f; w.AF2 = 0.f; w.AF = 0.f; w.jf = 0.f;
  const float inv_dth = (float)d.n_theta * 0.63661977237f;   // 1 / row spacing in theta
  // Normal case (a line that left through the port passes near O): every detector centre within rho_d of the line
  // lies within chord ch of one of the two points where the line pierces S(O,R) -> cap windows.
  bool caps = CAPS_TOO && a1 < 0.999f * Rf;
  float sF = 0.f, ch = 0.f, omega = 0.f;
  w.ch2 = 0.f;
  if (caps) {
    sF = sqrt_cull(R2 - dO2);
    const float smin = sqrt_cull(R2 - a1 * a1);
    const float a0 = fmaxf(0.f, dO - rho);
    const float smax = sqrt_cull(R2 - a0 * a0);
    const float ext = fmaxf(sF - smin, smax - sF);
    w.ch2 = fmaf(ext, ext, rho * rho) * 1.0001f + 1e-3f;
    caps = 4.0f * (R2 - dO2) > 4.04f * w.ch2;
    ch = sqrt_cull(w.ch2);
    omega = cap_angle(ch, rcp_cull(Rf));
  }
  if (!caps) {
    // Grazing or nearly tangent line (the caps would merge), or one that misses S(O,R): typical for the re-scattered rays of the
    // BRDF source model, which start on the world box.  Per-row box windows around the stretch of the line that can reach
    // the row (BoxLine above); same exact decision as on the cap path.
    BoxLine bx;
    int ilo, ihi;
    if (!box_line(d, P, V, inv_dth, bx, ilo, ihi)) { ISX_DIAG_ADD(3, 1); return; }
    ISX_DIAG_ADD(2, 1);
    walk_rows<MODE_BOX>(d, hist, rowt, colx, P, V, lane, ilo, ihi, w, bx, split, 2);
    return;
  }
  if (!CAPS_TOO) return;
  ISX_DIAG_ADD(1, 1);
#pragma unroll 1
  for (int side = 0; side < 2; ++side) {
    const double s = side == 0 ? ((double)sF - wv) : (-(double)sF - wv);
    const float Fx = (float)fma(s, V.x, P.x), Fy = (float)fma(s, V.y, P.y);
    w.Fz = (float)fma(s, V.z, P.z);
    if (w.Fz - ch > (float)d.portz) continue;  // cap entirely above every detector row
    w.AF2 = fmaf(Fx, Fx, Fy * Fy);
    w.AF = sqrt_cull(w.AF2);
    float phiF = atan2_cull(Fy, Fx);
    if (phiF < 0.f) phiF += 6.28318530718f;
    w.jf = phiF * w.inv_dphi - 0.5f;
    // rows that can intersect the cap: |theta_i - theta_F| <= omega, theta measured from -z about O
    const float thF = atan2_cull(w.AF, (float)d.portz - w.Fz);
    const int ilo = max((int)floorf((thF - omega) * inv_dth - 0.5f - 1e-3f), 0);
    const int ihi = min((int)ceilf((thF + omega) * inv_dth - 0.5f + 1e-3f), d.n_theta - 1);
    walk_rows<MODE_CAP>(d, hist, rowt, colx, P, V, lane, ilo, ihi, w, BoxLine(), split, 1);
  }
}

// ---- the fast-path lines of a batch of 64, their rows PACKED over the lanes (binning kernel).  A line of the headline
// configuration owns ~48 detector rows, so "lane = row, one line at a time" leaves a quarter of the wave idle through the window
// set-up, the coefficient set-up and the column walk.  Here the rows of all fast-path lines of the batch form one list (line l
// owns the slots [excl_l, incl_l) of it: a wave scan of the row counts) and the wave takes 64 slots at a time, whatever lines
// they belong to: lane -> slot -> (owner line, row).  The owner is found without a search: every line that owns a slot of
// the pass marks its first slot there in wave-private LDS, and a lane's owner is the nearest mark at or below it (ballot +
// count-leading-zeros).  The owner's cap comes through ds_bpermute, its line from the workspace again; the rest (window, split pass for long windows,
// coefficients, column walk, exact decision) is the code of walk_rows / walk_columns with per-lane instead of wave-uniform
// line data -- the same candidates, each once, so the same histogram.
__device__ __forceinline__ double shfl_f64(double x, int src) {
  const long long b = __double_as_longlong(x);
  const int lo = __shfl((int)(b & 0xffffffffll), src, 64), hi = __shfl((int)(b >> 32), src, 64);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <class D>
__device__ __forceinline__ void walk_lines_packed(const D& d, uint32_t* __restrict__ hist, const double* __restrict__ rowt,
                                                  const ColX* __restrict__ colx, const double* __restrict__ lines,
                                                  const RecPre& pre, int nrow, int excl, int incl, int total, float inv_dphi,
                                                  int lane, LdsInt* mark, LdsInt* split) {
#pragma unroll 1
  for (int base = 0; base < total; base += 64) {
    int npass = 1, logq = 0, nlong = 0;
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
      int owner = 0, i = 0, part = 0;
      bool have = false;
      if (pass == 0) {
        const int g = base + lane;
        have = g < total;
        volatile LdsInt* mk = mark;
        mk[lane] = 0;
        __builtin_amdgcn_wave_barrier();
        if (nrow > 0 && excl < base + 64 && incl > base) mk[(excl > base ? excl : base) - base] = lane + 1;
        __builtin_amdgcn_wave_barrier();
        const int m = mk[lane];
        const unsigned long long low = __ballot(m != 0) & (~0ull >> (63 - lane));   // marks at or below this lane
        const int pos = 63 - __builtin_clzll(low | 1ull);                           // (slot `base` always carries one)
        owner = mk[pos] - 1;
        if (!have || owner < 0) { owner = 0; have = false; }
        const int o_excl = __shfl(excl, owner, 64), o_rows = __shfl(pre.rows, owner, 64);
        i = (o_rows & 0xffff) + (g - o_excl);
      } else {
        const int slot = lane >> logq;
        part = lane & ((1 << logq) - 1);
        have = slot < nlong;
        const int e = have ? ((volatile LdsInt*)split)[slot] : 0;
        owner = e & 63;
        i = e >> 8;
      }
      if (!have) i = 0;
      // the owner's line and cap (every lane takes part in the exchange)
      CapWin w;
      w.inv_dphi = inv_dphi;
      w.Fz = __shfl(pre.Fz, owner, 64); w.AF = __shfl(pre.AF, owner, 64); w.AF2 = w.AF * w.AF;
      w.jf = __shfl(pre.jf, owner, 64); w.ch2 = __shfl(pre.ch2, owner, 64);
      V3 P, V;   // (read again from the batch's 3 KB of exit lines: lanes of one owner share the cache lines)
      {
        load_line(lines + 6 * owner, P, V);
      }
      int jlo = 0, cnt = 0;
      if (have) cap_window(w, (float)rowt[4 * i + 2], (float)rowt[4 * i + 3], d.n_phi, jlo, cnt);
      int start = 0, len = cnt;
      if (pass == 0) {
        const unsigned long long lm = __ballot(cnt > kSplitAt);
        if (lm != 0ull && __ballot(cnt >= kSplitAt + 16) != 0ull) {
          nlong = (int)__popcll(lm);
          logq = nlong <= 8 ? 3 : (nlong <= 16 ? 2 : (nlong <= 32 ? 1 : 0));
          if (logq > 0) {
            npass = 2;
            if (cnt > kSplitAt) {
              const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));
              ((volatile LdsInt*)split)[rank] = owner | (i << 8);
              len = kSplitAt;
            }
            __builtin_amdgcn_wave_barrier();
          }
        }
      } else {
        const int rem = cnt - kSplitAt;
        const int chunk = (rem + (1 << logq) - 1) >> logq;
        start = kSplitAt + part * chunk;
        len = rem - part * chunk;
        len = len < 0 ? 0 : (len > chunk ? chunk : len);
        if (!have) len = 0;
      }
      ISX_DIAG_ADD(11, 1); if (pass == 1) ISX_DIAG_ADD(10, 1);
      const double t0 = -fma(P.x, V.x, fma(P.y, V.y, (P.z - d.portz) * V.z));
      walk_columns(d, hist, colx, P, V, t0, lane, i, rowt, jlo, start, len, 0, lines + 6 * owner);
    }
  }
}

// ---- slot queues (isx_bin_slots_kernel).  A "slot" is one (exit line, detector row) pair with its column window; a column
// pass of 64 slots lasts as long as its widest window, and with slots taken in arrival order a pass of the headline ran at 58 %
// lane fill (windows of 1..24 columns side by side, mean 8.9).  Here every slot is first pushed into one of kClasses
// wave-private LDS queues by window length (1-4, 5-6, 7-8, 9-10, 11-12, 13-16, 17-24 columns; a longer window goes in as
// 16-column pieces), and a pass is made of 64 slots of ONE class, as soon as a class holds that many -- whatever lines and
// producers (packed rows of fast-path lines, cap rows, box rows of grazing lines) they came from.  The queues of a wave
// persist over the 256 lines of a work unit and are emptied at its end.  Same candidates, each once, same three-tier decision
// (walk_columns), so the same histogram.  Slot record, 32 bits: line within the unit | row << 8 | first column << 16 |
// columns << 24 -- the kernel serves grids of at most 256 rows and 255 columns (anything else: isx_bin_lines_kernel).
constexpr int kClasses = 7, kQueueCap = 128;   // a class never holds more than 63 + 64 slots
constexpr int kSlotWaveWords = kClasses * kQueueCap + 16 + 64;   // LDS words per wave: queues, counters, owner marks
constexpr int kDeferCap = kDeferCapW;   // candidates a wave can put aside for tiers 2 and 3
constexpr int kColWaveWords = kClasses * kQueueCap + 16 + 64 + 4 + kDeferCap;   // the same + the deferred list (count, 3 spare, entries)
constexpr int kPiece = 16, kLongest = 24;
struct SlotQueues {
  LdsWord* q;         // [kClasses][kQueueCap]
  LdsInt* tail;       // [8] slots pushed per class (running)
  LdsInt* head;       // [8] slots popped per class (running)
  const ColP* colp;   // column-pair table of the workgroup (isx_bin_slots_kernel)
  LdsWord* defer;     // [4 + kDeferCap] isx_bin_cols_kernel: [0] = entries held, [4..] = candidates that wait for tiers 2 and 3
};
__device__ __forceinline__ int slot_class(int cnt) {   // cnt in 1..kLongest
  return cnt <= 4 ? 0 : (cnt <= 12 ? (cnt - 3) >> 1 : (cnt <= 16 ? 5 : 6));
}

// The candidates a wave has put aside for tiers 2 and 3 (consume_cols).  A walk that meets a candidate tier 1 cannot decide appends
// it to the wave's deferred list -- one LDS atomic for the position, one write; the candidate counts as a miss in the walk -- and
// here the wave decides what has collected, lane = candidate, 64 at a time: one line fetch and one exact decision per lane with
// the whole wave at it, instead of one lane of a diverged wave waiting for its line to come back from L2 while the other 63 stand
// still (the ablation without tiers 2 and 3 runs 1.16 of 6.9 ms shorter on the headline, for 8e-4 of its candidates; the grazing
// lines of the BRDF source send 6e-3 of theirs this way -- their forms are seven times worse conditioned -- which is why the list is
// a DENSE append list, flushed whenever 64 have collected, since round 4: with two entries per lane, flushed twice per unit, four in
// five of that source's undecided candidates found their lane's entries taken and were decided on the spot).
// Entry: line within the unit | column << 8 | row << 16.  A hit goes to the bin; the walk added 0 for the candidate.
constexpr uint32_t kDeferFlushAt = 64;
template <class D>
__device__ __forceinline__ void flush_deferred(const D& d, uint32_t* __restrict__ hist, const double* __restrict__ rowt,
                                               const ColX* __restrict__ colx, const double* __restrict__ lines, const SlotQueues& sq,
                                               int lane) {
  volatile LdsWord* df = sq.defer;
  typedef __attribute__((address_space(3))) uint32_t LdsU32;
  const uint32_t held = df[0];                                        // (same address in every lane: wave-uniform; it keeps counting past the capacity)
  const uint32_t n = held < (uint32_t)kDeferCap ? held : (uint32_t)kDeferCap;
#pragma unroll 1
  for (uint32_t base = 0; base < n; base += 64u) {
    const bool have = base + (uint32_t)lane < n;
    const uint32_t e = have ? df[4u + base + (uint32_t)lane] : 0u;
    const int line = (int)(e & 255u), jr = (int)((e >> 8) & 255u), ir = (int)((e >> 16) & 255u);
    bool hit = false;
    if (have) hit = decide_exact(d, rowt, colx, lines + 6 * line, ir, jr);
    if (hit) __hip_atomic_fetch_add(reinterpret_cast<LdsU32*>((__attribute__((address_space(3))) void*)hist) + (ir * d.n_phi + jr), 1u,
                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) df[0] = 0u;
  __builtin_amdgcn_wave_barrier();
}

// one pass: 64 slots (fewer when a unit's leftovers are flushed), lane = slot
template <class D>
__device__ __forceinline__ void consume_slots(const D& d, uint32_t* __restrict__ hist, const double* __restrict__ rowt,
                                              const ColX* __restrict__ colx, const double* __restrict__ lines, uint32_t rec,
                                              bool active, int lane, const SlotQueues& sq) {
  const int line = (int)(rec & 255u), i = (int)((rec >> 8) & 255u), jlo = (int)((rec >> 16) & 255u);
  const int len = active ? (int)(rec >> 24) : 0;
  const double* src6 = lines + 6 * line;   // (an idle lane reads line 0 of the unit, which exists)
  double t0 = 0.0;
  {
    V3 P, V;
    load_line(src6, P, V);
    t0 = -fma(P.x, V.x, fma(P.y, V.y, (P.z - d.portz) * V.z));
  }
  ISX_DIAG_ADD(11, 1);
  ISX_BD_MARK(sq, 3);
  walk_columns_pairs(d, hist, colx, sq.colp, t0, i, rowt, jlo, len, src6, lane, sq.defer, line);
  ISX_BD_MARK(sq, 4);
}

// passes for every class that holds at least `least` slots (64 while a unit is being produced, 1 at its end)
template <class D>
__device__ __forceinline__ void drain_slots(const D& d, uint32_t* __restrict__ hist, const double* __restrict__ rowt,
                                            const ColX* __restrict__ colx, const double* __restrict__ lines,
                                            const SlotQueues& sq, int least, int lane) {
  volatile LdsInt* tl = sq.tail;
  volatile LdsInt* hd = sq.head;
  const int held = lane < kClasses ? tl[lane] - hd[lane] : 0;
  unsigned long long m = __ballot(held >= least);
  while (m) {
    const int c = __builtin_ctzll(m);
    m &= m - 1ull;
    const int h = hd[c], n = tl[c] - h;          // (same address in every lane: wave-uniform)
    const int take = n < 64 ? n : 64;
    const uint32_t rec = lane < take ? ((volatile LdsWord*)sq.q)[c * kQueueCap + ((h + lane) & (kQueueCap - 1))] : 0u;
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) hd[c] = h + take;
    __builtin_amdgcn_wave_barrier();
    consume_slots(d, hist, rowt, colx, lines, rec, lane < take, lane, sq);
  }
}

// every lane hands in (at most) one window: columns [jlo, jlo + cnt) of row i for line `line` of the unit (cnt = 0: nothing).
// jlo may be negative by less than n_phi (cap_window).
template <class D>
__device__ __forceinline__ void push_slots(const D& d, uint32_t* __restrict__ hist, const double* __restrict__ rowt,
                                           const ColX* __restrict__ colx, const double* __restrict__ lines,
                                           const SlotQueues& sq, int line, int i, int jlo, int cnt, int lane) {
  ISX_BD_MARK(sq, 1);
  int j = jlo < 0 ? jlo + d.n_phi : jlo, rem = cnt;
  for (;;) {
    const int piece = rem > kLongest ? kPiece : rem;
    if (piece > 0) {
      const int c = slot_class(piece);
      const int pos = __hip_atomic_fetch_add(sq.tail + c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      ((volatile LdsWord*)sq.q)[c * kQueueCap + (pos & (kQueueCap - 1))] =
          (uint32_t)line | ((uint32_t)i << 8) | ((uint32_t)j << 16) | ((uint32_t)piece << 24);
      ISX_DIAG_ADD_LANES(7, piece);
    }
    rem -= piece;
    j += piece;
    if (j >= d.n_phi) j -= d.n_phi;
    __builtin_amdgcn_wave_barrier();
    ISX_BD_MARK(sq, 2);
    drain_slots(d, hist, rowt, colx, lines, sq, 64, lane);   // every class is below 64 again before the next push
    if (__ballot(rem > 0) == 0ull) break;
  }
}

// What the two caps of a line share.  Every detector centre c the line can hit lies within rho of it (the disc of radius rho about c
// holds a point of the line), i.e. on the stretch of S(O,R) inside the tube of radius rho about the line.  With H the foot of O on the
// line, h = |H|, eta = h/R, kappa = rho/R and c^ = a h^ + b V + e u (u = V x h^), the tube is (a - eta)^2 + e^2 <= kappa^2; on the side
// b > 0 it meets the plane of the line and O between A1 = (h + rho) h^ + smin V and A2 = (h - rho) h^ + smx V, smin = sqrt(R^2 - (h +
// rho)^2), smx = sqrt(R^2 - (h - rho)^2).  The cap about their bisector m^ that passes through both holds the whole stretch: for a
// fixed `a` the point of the stretch farthest from m^ = (cm, sm, 0), sm > 0, has the largest e allowed (the smallest b), there
// b^2 = 1 - kappa^2 + eta^2 - 2 eta a, and cos(distance) = a cm + sm sqrt(1 - kappa^2 + eta^2 - 2 eta a) is concave in a, so its minimum
// over a in [eta - kappa, eta + kappa] is at an end: A1 or A2, both at the cap's rim.  m^ is the direction of A1 + A2 = 2 H + (smin +
// smx) V, i.e. of the line's point at parameter sM = (smin + smx)/2 from the foot, and |A2 - A1|^2 = 4 rho^2 + (smx - smin)^2 =
// 4 R^2 sin^2 w.  (Until round 4 the cap was drawn about the piercing point F with the chord sqrt(ext^2 + rho^2), ext the longer of
// the two stretches of the LINE inside the tube's ends: the same for lines through O, up to twice the radius for lines that pass O at
// 0.5-0.8 R -- a fifth of the BRDF source's candidates.)  Binary32 with slack: |A2 - A1|^2 takes 1e-4 relative + 4e-3 cm^2, cos w
// another 2e-6; the directions carry the roundings that cap_rows / prep_cols' slack terms cover, as before.
struct CapShared {
  double wv;              // line parameter of the foot of O, negated: the caps' centres are the directions of the points at +-sM - wv
  float sM, iL, cosw;     // (smin + smx)/2, 1/|H + sM V|, cos w
  int kind;               // 0 the line has caps, -1 no caps: grazing line, -2 the line cannot hit anything
};
__device__ __forceinline__ CapShared prep_shared(const GridConst& k, const V3& P, const V3& V) {
  CapShared o;
  o.wv = 0.0; o.sM = o.iL = o.cosw = 0.f; o.kind = -1;
  const double wz = P.z - (double)k.portz;
  const double wv = fma(P.x, V.x, fma(P.y, V.y, wz * V.z));
  const double hx = fma(-wv, V.x, P.x), hy = fma(-wv, V.y, P.y), hz = fma(-wv, V.z, wz);
  const float dO2 = (float)fma(hx, hx, fma(hy, hy, hz * hz));
  const float R2 = k.Rf * k.Rf;
  const float dO = sqrt_cull(dO2);
  const float a1 = dO + k.rho;
  if (dO - k.rho > 1.001f * k.Rf) { o.kind = -2; return o; }
  if (!(a1 < 0.999f * k.Rf)) return o;
  const float am = dO - k.rho;
  const float smin = sqrt_cull(fmaf(-a1, a1, R2));
  const float smx = sqrt_cull(fmaf(-am, am, R2));
  const float sM = 0.5f * (smin + smx), ds = smx - smin;
  const float d2 = fmaf(ds, ds, 4.0f * k.rho * k.rho) * 1.0001f + 4e-3f;
  const float iR = rcp_cull(k.Rf);
  const float cosw = sqrt_cull(fmaxf(0.f, fmaf(-0.25f * d2, iR * iR, 1.0f))) - 2e-6f;
  const float iL = __builtin_amdgcn_rsqf(fmaf(sM, sM, dO2));
  const float sinw = sqrt_cull(fmaxf(0.f, fmaf(-cosw, cosw, 1.0f)));
  // caps well apart (their centres are 2 asin(sM / L) apart as seen from O) and narrower than 60 degrees, or the line is taken as a
  // grazing line (a cap about h^ and a band: right for any line)
  if (!(cosw > 0.5f && sM * iL > fmaf(sinw, 1.01f, 1e-3f))) return o;
  o.wv = wv; o.sM = sM; o.iL = iL; o.cosw = cosw; o.kind = 0;
  return o;
}
// chord^2 from a cap's centre ON the sphere to its rim: 2 R^2 (1 - cos w)
__device__ __forceinline__ float cap_chord2(const GridConst& k, const CapShared& sh) { return 2.0f * k.Rf * k.Rf * (1.0f - sh.cosw); }
// the centre of the cap of side `side` (0: the larger line parameter) as a point of the line: (x, y, z - portz), binary32
__device__ __forceinline__ void cap_centre(const GridConst& k, const CapShared& sh, const V3& P, const V3& V, int side, float& mx, float& my,
                                           float& mz) {
  const double s0 = side == 0 ? ((double)sh.sM - sh.wv) : (-(double)sh.sM - sh.wv);
  mx = (float)fma(s0, V.x, P.x); my = (float)fma(s0, V.y, P.y); mz = (float)(fma(s0, V.z, P.z) - (double)k.portz);
}
// whether the cap of side `side` reaches detector rows at all (its lowest point lies below the port plane)
__device__ __forceinline__ bool cap_is_low(const GridConst& k, const CapShared& sh, const V3& P, const V3& V, int side) {
  float mx, my, mz;
  cap_centre(k, sh, P, V, side, mx, my, mz);
  return !(mz * sh.iL * k.Rf - sqrt_cull(cap_chord2(k, sh)) > 0.f);
}
// Can a bin lie in the row ranges (cap_rows) of BOTH caps of the line?  prep_shared keeps the caps' centres more than 2.02 sin w
// apart (as sines of half their distance seen from O), so the caps themselves are disjoint -- but a row range carries slack, and on a
// coarse grid the two ranges of one column can meet (tools/soak_cull.py found exactly that on 2- and 3-row grids).  Along the
// meridian of column j the angular distance D to the cap's centre obeys cos D = rho cos(theta - tc), rho = sqrt(a^2 + b^2) <= 1;
// cap_rows hands out the rows with |theta_i - tc| <= acos(cos w / rho) + s, s <= 2.5e-3 + 1e-4 (acos_cull) + 2e-5 (atan2_cull) + 1e-3
// rows (<= 1.6e-3 rad on a one-row grid) <= 4.3e-3, with cos w and rho good to 3e-6 (the 2e-6 taken off cos w, binary32 rounding):
// for those cos D >= cos w - s sqrt(rho^2 - cos^2 w) - s^2/2 - 3e-6 >= cos w - s sin w - 1.3e-5 >= cos(w + 5e-3), i.e. every row
// handed out has its centre within w + 5e-3 rad of the cap's centre.  A common bin therefore needs the centres within 2 w + 1e-2 of
// each other as seen from O; they are 2 asin(sM / L) apart, L = |H + sM V|.  Hence: no common bin if sM / L > sin(w + 5e-3), for
// which sM / L > sin w + 5e-3 suffices while w + 5e-3 < pi/2.  Evaluated in binary32 with its own margin (8e-3, 1e-4 relative;
// cos w > 0.05).  A line that fails the test AND has its second cap among the detector rows is taken as a grazing line (one cap about
// h^ and the band: every bin at most once) instead of its two caps: rare (none in any BASELINE configuration: there sM / L >= 0.6
// against sin w ~ 0.2), correct for any line.
__device__ __forceinline__ bool caps_may_touch(const GridConst& k, const CapShared& sh) {
  const float sinw = sqrt_cull(fmaxf(0.f, fmaf(-sh.cosw, sh.cosw, 1.0f)));
  return !(sh.cosw > 0.05f && sh.sM * sh.iL > fmaf(sinw, 1.0001f, 8e-3f));
}
// The cap of side `side` of a line with caps as the ROW producer wants it (prep_record's and bin_culled's formulas and margins):
// own.{smax, smin, vxy, avz} carry {Fz, AF, jf, ch2} of the CapWin (the producer's per-owner data is one BoxLine either way),
// [ilo, ihi] the rows the cap can reach (empty if it lies above every detector row).
__device__ __forceinline__ void cap_rows_pre(const GridConst& k, const V3& P, const V3& V, const CapShared& sh, int side, BoxLine& own,
                                             int& ilo, int& ihi) {
  ilo = 0; ihi = -1;
  float mx, my, mz;
  cap_centre(k, sh, P, V, side, mx, my, mz);
  const float ch2 = cap_chord2(k, sh), ch = sqrt_cull(ch2);
  const float sc = sh.iL * k.Rf;                                     // the centre ON the sphere: G = O + R m^
  const float Gz = fmaf(mz, sc, k.portz);
  if (Gz - ch > k.portz) return;                                     // cap entirely above every detector row
  const float Gx = mx * sc, Gy = my * sc;
  const float AF = sqrt_cull(fmaf(Gx, Gx, Gy * Gy));
  float phiF = atan2_cull(Gy, Gx);
  if (phiF < 0.f) phiF += 6.28318530718f;
  const float omega = cap_angle(ch, rcp_cull(k.Rf));
  const float thF = atan2_cull(AF, k.portz - Gz);
  ilo = max((int)floorf((thF - omega) * k.inv_dth - 0.5f - 1e-3f), 0);
  ihi = min((int)ceilf((thF + omega) * k.inv_dth - 0.5f + 1e-3f), k.n_theta - 1);
  own.smax = Gz; own.smin = AF; own.vxy = phiF * k.inv_dphi - 0.5f; own.avz = ch2;
}

// producer of isx_bin_slots_kernel: the rows of the lines of a batch that take part in one pass, packed over the lanes -- lane =
// (line, row); the owner of a slot is found without a search (every line marks its first slot of the 64 in wave-private LDS; nearest
// mark at or below the lane: ballot + clz) and its per-line data (`own`: a cap, or the box of a grazing line) comes from the line's
// own lane through ds_bpermute.  `boxes`: the rows get the (at most two) box windows of a grazing line, else the cap window.
// One producer and ONE push site for every kind of line since round 4 (before: fast-path caps packed, grazing lines' boxes packed
// in a second copy, the lines with two caps one at a time, lane = row, ~900 instructions per line: five inlined copies of the
// consumer, 144 bytes of scratch per lane).  Same windows as before, each (line, row) once: the same slots in another order.
template <class D>
__device__ __forceinline__ void produce_rows_packed(const D& d, uint32_t* __restrict__ hist, const double* __restrict__ rowt,
                                                    const ColX* __restrict__ colx, const double* __restrict__ lines, const SlotQueues& sq,
                                                    const BoxLine& own, int own_ilo, int nrow, int excl, int incl, int total, bool boxes,
                                                    float inv_dphi, float portz, int first_line, int lane, LdsInt* mark) {
#pragma unroll 1
  for (int base = 0; base < total; base += 64) {
    const int g = base + lane;
    bool have = g < total;
    volatile LdsInt* mk = mark;
    mk[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    if (nrow > 0 && excl < base + 64 && incl > base) mk[(excl > base ? excl : base) - base] = lane + 1;
    __builtin_amdgcn_wave_barrier();
    const int m = mk[lane];
    const unsigned long long low = __ballot(m != 0) & (~0ull >> (63 - lane));   // marks at or below this lane
    const int pos = 63 - __builtin_clzll(low | 1ull);                           // (slot `base` always carries one)
    int owner = mk[pos] - 1;
    if (!have || owner < 0) { owner = 0; have = false; }
    const int o_excl = __shfl(excl, owner, 64), o_ilo = __shfl(own_ilo, owner, 64);
    BoxLine b;
    b.smax = __shfl(own.smax, owner, 64); b.smin = __shfl(own.smin, owner, 64); b.vxy = __shfl(own.vxy, owner, 64);
    b.avz = __shfl(own.avz, owner, 64);
    b.ivz = b.dn = b.Hm = b.Hz = b.phin = b.rs = b.sig = 0.f;
    if (boxes) {
      b.ivz = __shfl(own.ivz, owner, 64); b.dn = __shfl(own.dn, owner, 64); b.Hm = __shfl(own.Hm, owner, 64);
      b.Hz = __shfl(own.Hz, owner, 64); b.phin = __shfl(own.phin, owner, 64); b.rs = __shfl(own.rs, owner, 64);
      b.sig = __shfl(own.sig, owner, 64);
    }
    const int i = have ? o_ilo + (g - o_excl) : 0;
    int j0 = 0, c0 = 0, j1 = 0, c1 = 0;
    if (have) {
      const float zi = (float)rowt[4 * i + 2], Ai = (float)rowt[4 * i + 3];
      if (boxes) box_window(b, zi - portz, Ai, d.n_phi, inv_dphi, j0, c0, j1, c1);
      else {
        CapWin w;
        w.inv_dphi = inv_dphi; w.Fz = b.smax; w.AF = b.smin; w.AF2 = b.smin * b.smin; w.jf = b.vxy; w.ch2 = b.avz;
        cap_window(w, zi, Ai, d.n_phi, j0, c0);
      }
    }
    const int nwin = (boxes && __ballot(c1 > 0) != 0ull) ? 2 : 1;   // (the second windows of the rows that have one: a second push)
#pragma unroll 1
    for (int w = 0; w < nwin; ++w) push_slots(d, hist, rowt, colx, lines, sq, first_line + owner, i, w == 0 ? j0 : j1, w == 0 ? c0 : c1, lane);
  }
}

// ---- COLUMN slots (isx_bin_cols_kernel).  The grid is anisotropic -- rows 0.5 deg apart, columns 4 deg -- so the cap of an exit
// line (angular radius ~11.5 deg for the 40 cm detector) spans ~46 rows but only ~9 columns: a (line, row) slot holds ~9 candidates,
// a (line, COLUMN) slot ~30, and everything that is paid per slot -- owner search, window, queue push and pop, line fetch,
// coefficients -- is paid six times less often per line.  For a fixed column (cos phi, sin phi) the four forms of the hit
// polynomial are affine in the row quantities S = sin theta_i, C = cos theta_i, T = -R cos^2 theta_i:
//   with q = the line's point nearest to O (relative to O), c - O = (R S cph, R S sph, -R C), n = (-S sph, S cph, -C),
//   al = Vy cph - Vx sph, be = qy cph - qx sph, ga = Vx cph + Vy sph, de = qx cph + qy sph:
//     dot  = V.n              = al S - Vz C
//     num  = (q - c).n        = be S - qz C + T
//     m2dv = -2 (q - c).V     = -2 q.V + 2 R ga S - 2 R Vz C
//     ddw  = |q-c|^2-(w/2)^2  = |q|^2 + R^2 - (w/2)^2 - 2 R de S + 2 R qz C
// and g = dot (dot ddw + num m2dv) + num^2 < 0 is the hit, exactly as in walk_columns.  Binary32 tier with the same error
// accounting: every coefficient and every table value carries one rounding u = 2^-24, a form is at most three fused operations,
// so |delta form| <= 5u M_form with M_form the sum of the coefficient magnitudes times the table bounds (S, C <= 1, |T| <= R);
// through dg as in walk_columns |delta g| <= (15 + 4) u S, S = Md (Md Mf + Mn Mv) + Mn^2; the band is 1.4e-6 S.  What tier 1
// cannot decide goes to the binary64 tier and the reference-order test of walk_columns, unchanged.  The rows of a column inside the
// cap (angular radius w about the piercing point F, cos w = 1 - ch^2 / 2R^2): with a = cos theta_F, b = sin theta_F cos(phi - phi_F)
// = (cph Fx + sph Fy)/R, the cap is a cos theta + b sin theta >= cos w, i.e. |theta - atan2(b, a)| <= acos(cos w / sqrt(a^2+b^2)).
// Two rows per walk step: entry i holds rows i and i+1 side by side, so that the four forms and g of BOTH candidates are packed
// binary32 operations (8 + 4 v_pk instructions for two candidates where one at a time took 4 + 5 each).  The second candidate of
// the last step of a slot with an odd row count is masked (NOT made a real candidate: one more row could reach into the range the
// line's other cap holds in the same column, and the bin would count twice -- tools/soak_cull.py found exactly that on 2- and
// 3-row grids); the table has a spare zero entry so that the last pair of the grid can be read.
struct __align__(16) RowX { float S0, S1, C0, C1, T0, T1; uint32_t pad0, pad1; };   // sin, cos, -R cos^2 of rows i, i+1
constexpr int kColPiece = 48, kColLongest = 64;
__device__ __forceinline__ int col_class(int cnt) { return cnt <= 48 ? (cnt - 1) >> 3 : 6; }   // 1-8, 9-16, ..., 41-48, 49-64

struct ColPre {     // per line and piercing point: its cap as the column producer wants it
  float fx, fy, a, cosw;   // F_xy / R, cos theta_F, cos w
  int jlo, ncol;           // first column (in [0, n_phi)) and number of columns the cap can reach
  int kind;                // 0 a cap with columns, -1 no caps: grazing line (bin_culled's box windows), -2 the line cannot hit
                           // anything, -3 this side's cap lies above every detector row
};
// side 0: the piercing point at the larger line parameter (the only one of a fast-path line), side 1: the other one.
__device__ __forceinline__ ColPre prep_cols(const GridConst& k, int n_phi, const V3& P, const V3& V, const CapShared& sh, int side) {
  ColPre o;
  o.fx = o.fy = o.a = o.cosw = 0.f; o.jlo = 0; o.ncol = 0; o.kind = sh.kind;
  if (sh.kind != 0) return o;
  float Fx, Fy, mz;
  cap_centre(k, sh, P, V, side, Fx, Fy, mz);
  if (mz * sh.iL * k.Rf - sqrt_cull(cap_chord2(k, sh)) > 0.f) { o.kind = -3; return o; }   // cap entirely above every detector row
  o.fx = Fx * sh.iL; o.fy = Fy * sh.iL; o.a = -(mz * sh.iL);
  o.cosw = sh.cosw;
  o.kind = 0;
  // columns: sin theta_F |sin(phi - phi_F)| <= sin w
  const float sinF = sqrt_cull(fmaf(o.fx, o.fx, o.fy * o.fy));
  const float sinw = sqrt_cull(fmaxf(0.f, fmaf(-o.cosw, o.cosw, 1.0f)));
  if (!(sinF > sinw * 1.01f + 1e-4f)) { o.jlo = 0; o.ncol = n_phi; return o; }   // the cap holds the pole: every column
  const float r = fminf(1.0f, sinw * rcp_cull(sinF) * 1.001f);
  const float dphi = (1.57079637f - acos_cull(r)) + 3e-3f;
  float phiF = atan2_cull(Fy, Fx);
  if (phiF < 0.f) phiF += 6.28318530718f;
  const float jc = phiF * k.inv_dphi - 0.5f, hw = dphi * k.inv_dphi + 0.02f;
  const int lo = (int)ceilf(jc - hw), hi = (int)floorf(jc + hw);
  int n = hi - lo + 1;
  if (n >= n_phi) { o.jlo = 0; o.ncol = n_phi; return o; }
  if (n < 0) n = 0;
  int j0 = lo;
  if (j0 < 0) j0 += n_phi;
  if (j0 >= n_phi) j0 -= n_phi;
  o.jlo = j0; o.ncol = n;
  return o;
}

// the rows [ilo, ilo + cnt) of column (c32, s32) that the cap of a line can reach
__device__ __forceinline__ void cap_rows(float fx, float fy, float a, float cosw, float c32, float s32, float inv_dth, int n_theta,
                                         int& ilo, int& cnt) {
#if ISX_ABL == 3
  ilo = 20; cnt = 30; return;
#endif
  ilo = 0; cnt = 0;
  const float b = fmaf(c32, fx, s32 * fy);
  const float rho2 = fmaf(a, a, b * b);
  if (!(rho2 > 1e-12f)) return;
  const float x = cosw * __builtin_amdgcn_rsqf(rho2);
  if (x > 1.0f) return;                                               // the column misses the cap
  // (a cap wider than a quarter turn in the meridian's plane -- only the cap about h^ of a grazing line can be, prep_band: the caps
  //  of prep_cols have cos w > 0.5 -- would need theta taken modulo 2 pi, and leaves out a short stretch at most: every row)
  if (x < 0.0f) { ilo = 0; cnt = n_theta; return; }
  const float dl = acos_cull(fmaxf(x, -1.0f)) + 2.5e-3f;
  const float tc = atan2_cull(b, a);
  const float tlo = tc - dl, thi = tc + dl;
  if (thi < 0.f || tlo > 1.57079637f) return;
  // rows whose centre (i + 1/2) dtheta lies in [tlo, thi] (1e-3 of a row of slack on top of the 2.5e-3 rad above)
  const int lo = max((int)ceilf(tlo * inv_dth - 0.5f - 1e-3f), 0);
  const int hi = min((int)floorf(thi * inv_dth - 0.5f + 1e-3f), n_theta - 1);
  if (hi >= lo) { ilo = lo; cnt = hi - lo + 1; }
}

// ---- GRAZING lines as column slots (round 4; until then such a line was taken one at a time, lane = row: bin_culled's box
// windows).  A line without caps passes S(O,R) at a distance h from O of about R (or its caps would merge: rho_d comparable to R).
// Every detector centre c it can hit lies within rho of the line, X(s) = H + s V (H the foot of O, |V| = 1), hence -- e = the part of
// c - H perpendicular to V, |e| <= rho -- inside two slabs:
//   (c - H).h^ >= -rho  (h^ = H/h)        =>  c.h^ >= h - rho: the CAP of angular radius acos((h - rho)/R) about the direction h^
//                                              (it bounds the LENGTH of the stretch of S(O,R) the tube crosses),
//   |(c - H).u| <= rho,  u = V x h^        =>  |c.u| <= rho: the BAND of half-width asin(rho/R) about the great circle
//                                              perpendicular to u (it bounds the stretch's WIDTH; H.u = 0),
// a "rectangle" on the sphere around the tube's footprint (4/pi of its area, like the box windows' bounding boxes).  The cap is what
// prep_cols / cap_rows already handle (F := R h^, cos w := (h - rho)/R); the band adds, per column, |N sin(theta - delta)| <= rho/R with
// N e^(i delta) = (c_phi u_x + s_phi u_y) + i u_z, i.e. theta within asin(rho / (R N)) of delta + m pi: at most two stretches of
// rows inside the cap's range.  Binary32 with explicit slack (rho: 0.1 % + 2e-3 cm as in box_line; unit vectors and R: 4e-6; angles
// 2.6e-3 rad + the cull functions' 1e-4; rows 1e-3): never decides a result, a candidate outside the tube is a miss of the exact test.
struct BandPre { float ux, uy, uz, kap; };   // u = V x h^, kappa = rho / R with its slack
__device__ __forceinline__ ColPre prep_band(const GridConst& k, int n_phi, const V3& P, const V3& V, BandPre& bp) {
  ColPre o;
  o.fx = o.fy = o.a = o.cosw = 0.f; o.jlo = 0; o.ncol = 0; o.kind = -2;
  bp.ux = bp.uy = bp.uz = 0.f; bp.kap = 2.f;
  const double wz = P.z - (double)k.portz;
  const double wv = fma(P.x, V.x, fma(P.y, V.y, wz * V.z));
  const float Hx = (float)fma(-wv, V.x, P.x), Hy = (float)fma(-wv, V.y, P.y), Hz = (float)fma(-wv, V.z, wz);
  const float Vx = (float)V.x, Vy = (float)V.y, Vz = (float)V.z;
  const float h = sqrt_cull(fmaf(Hx, Hx, fmaf(Hy, Hy, Hz * Hz)));
  if (h - k.rho > 1.001f * k.Rf) return o;                           // farther than R + rho_d from O: nothing can be hit (box_line's test)
  const float rs = fmaf(k.rho, 1.001f, 2e-3f);
  const float iR = rcp_cull(k.Rf);
  float ex, ey, ez, cosw, kap_extra = 0.f;
  if (h > 1e-3f * k.Rf) {
    const float ih = rcp_cull(h);
    ex = Hx * ih; ey = Hy * ih; ez = Hz * ih;
    cosw = fmaxf(-1.0f, (h - rs) * iR - 4e-6f);
  } else {
    // the line passes through O (to 0.1 % of R): any unit vector perpendicular to V serves as h^, and the cap is the whole sphere.
    // H.u is then no longer 0 but anything up to h, so the band |c.u| <= rho + |H.u| is wider by h / R
    kap_extra = h * iR * 1.0001f;
    const float ax = fabsf(Vx), ay = fabsf(Vy), az = fabsf(Vz);
    float tx = 0.f, ty = 0.f, tz = 0.f;
    if (ax <= ay && ax <= az) tx = 1.f; else if (ay <= az) ty = 1.f; else tz = 1.f;
    const float tv = fmaf(tx, Vx, fmaf(ty, Vy, tz * Vz));
    ex = fmaf(-tv, Vx, tx); ey = fmaf(-tv, Vy, ty); ez = fmaf(-tv, Vz, tz);
    const float ie = __builtin_amdgcn_rsqf(fmaf(ex, ex, fmaf(ey, ey, ez * ez)));
    ex *= ie; ey *= ie; ez *= ie;
    cosw = -1.0f;
  }
  o.fx = ex; o.fy = ey; o.a = -ez; o.cosw = cosw; o.kind = 0;
  bp.ux = fmaf(Vy, ez, -(Vz * ey)); bp.uy = fmaf(Vz, ex, -(Vx * ez)); bp.uz = fmaf(Vx, ey, -(Vy * ex));
  bp.kap = fmaf(rs * iR, 1.0001f, 4e-6f) + kap_extra;
  // columns the cap can reach (prep_cols): all of them if it holds the pole or is wider than a hemisphere
  const float sinF = sqrt_cull(fmaf(o.fx, o.fx, o.fy * o.fy));
  const float sinw = sqrt_cull(fmaxf(0.f, fmaf(-cosw, cosw, 1.0f)));
  if (!(cosw > 0.05f) || !(sinF > sinw * 1.01f + 1e-4f)) { o.jlo = 0; o.ncol = n_phi; return o; }
  const float r = fminf(1.0f, sinw * rcp_cull(sinF) * 1.001f);
  const float dphi = (1.57079637f - acos_cull(r)) + 3e-3f;
  float phiF = atan2_cull(o.fy, o.fx);
  if (phiF < 0.f) phiF += 6.28318530718f;
  const float jc = phiF * k.inv_dphi - 0.5f, hw = dphi * k.inv_dphi + 0.02f;
  const int lo = (int)ceilf(jc - hw), hi = (int)floorf(jc + hw);
  int n = hi - lo + 1;
  if (n >= n_phi) { o.jlo = 0; o.ncol = n_phi; return o; }
  if (n < 0) n = 0;
  int j0 = lo;
  if (j0 < 0) j0 += n_phi;
  if (j0 >= n_phi) j0 -= n_phi;
  o.jlo = j0; o.ncol = n;
  return o;
}
// the rows of column (c32, s32) inside the band, cut to [row_lo, row_hi] (what the cap allows): [ilo, ilo + cnt) and, if the
// meridian enters the band twice, [ilo_b, ilo_b + cnt_b) -- disjoint (the stretches are pi apart, each shorter than 0.9 pi)
__device__ __forceinline__ void band_rows(const BandPre& bp, float c32, float s32, float inv_dth, int row_lo, int row_hi, int& ilo, int& cnt,
                                          int& ilo_b, int& cnt_b) {
  ilo = cnt = ilo_b = cnt_b = 0;
  if (row_hi < row_lo) return;
  const float ga = fmaf(c32, bp.ux, s32 * bp.uy);
  const float N2 = fmaf(ga, ga, bp.uz * bp.uz);
  if (!(N2 > bp.kap * bp.kap * 1.03f + 1e-12f)) { ilo = row_lo; cnt = row_hi - row_lo + 1; return; }   // (nearly) the whole meridian lies in the band
  const float x = bp.kap * __builtin_amdgcn_rsqf(N2);               // < 0.986
  const float al = (1.57079637f - acos_cull(x)) + 2.6e-3f;          // asin(x) + slack: < 1.41, so the stretches are more than 0.3 rad apart
  const float de = atan2_cull(bp.uz, ga);
#pragma unroll
  for (int m = -1; m <= 1; ++m) {
    const float ce = fmaf((float)m, 3.14159274f, de);
    const float tlo = ce - al, thi = ce + al;
    if (thi < 0.f || tlo > 1.57079637f) continue;
    const int lo = max((int)ceilf(tlo * inv_dth - 0.5f - 1e-3f), row_lo);
    const int hi = min((int)floorf(thi * inv_dth - 0.5f + 1e-3f), row_hi);
    if (hi < lo) continue;
    if (cnt == 0) { ilo = lo; cnt = hi - lo + 1; }
    else { ilo_b = lo; cnt_b = hi - lo + 1; }
  }
}

// one pass: 64 column slots (record: line within the unit | column << 8 | first row << 16 | rows << 24), lane = slot
template <class D>
__device__ __forceinline__ void consume_cols(const D& d, uint32_t* __restrict__ hist, const double* __restrict__ rowt,
                                             const ColX* __restrict__ colx, const RowX* __restrict__ rowx,
                                             const double* __restrict__ lines, uint32_t rec, bool active, int lane,
                                             const SlotQueues& sq) {
  // (-DISX_DIAG, tools/diag_binning.py: wave cycles [16] batch preparation, [17] cap rows + what the compiler sinks to the push,
  //  [18] push, [19] pop, [20] slot prologue (line fetch, coefficients), [21] walk, [22] owner search, [23] the owner's cap
  //  through ds_bpermute, [24] (cap rows as placed in the source), [25] producer loop control + drain check.
  //  -DISX_DIAG_TIMING_ONLY: the timers without the per-candidate re-decision and counters, which inflate the walk tenfold)
  ISX_BD_MARK(sq, 3);
#if ISX_ABL == 2
  return;
#endif
  const int line = (int)(rec & 255u), j = (int)((rec >> 8) & 255u), ilo = (int)((rec >> 16) & 255u);
  const int len = active ? (int)(rec >> 24) : 0;
  const double* src6 = lines + 6 * line;
  float cAl = 0.f, cBe = 0.f, cVz = 0.f, cQz = 0.f, cE0 = 0.f, cE1 = 0.f, cE2 = 0.f, cF0 = 0.f, cF1 = 0.f, cF2 = 0.f;
  float band32 = 0.f;
  if (len > 0) {
    V3 P, V;
    {
      load_line(src6, P, V);
    }
    const double wz = P.z - d.portz;
    const double t0 = -fma(P.x, V.x, fma(P.y, V.y, wz * V.z));
    const double qx = fma(t0, V.x, P.x), qy = fma(t0, V.y, P.y), qz = fma(t0, V.z, wz);   // nearest to O, relative to O
    const double cph = colx[j].c, sph = colx[j].s;
    const double al = fma(V.y, cph, -(V.x * sph)), be = fma(qy, cph, -(qx * sph));
    const double ga = fma(V.x, cph, V.y * sph), de = fma(qx, cph, qy * sph);
    const double R = d.R, R2x = 2.0 * R;
    const double e0 = -2.0 * fma(qx, V.x, fma(qy, V.y, qz * V.z));
    const double f0 = fma(qx, qx, fma(qy, qy, fma(qz, qz, R * R))) - d.half_w2;
    const double e1 = R2x * ga, e2 = -(R2x * V.z), f1 = -(R2x * de), f2 = R2x * qz;
    cAl = (float)al; cBe = (float)be; cVz = (float)(-V.z); cQz = (float)(-qz);
    cE0 = (float)e0; cF0 = (float)f0; cE1 = (float)e1; cF1 = (float)f1; cE2 = (float)e2; cF2 = (float)f2;
    const double Md = fabs(al) + fabs(V.z), Mn = fabs(be) + (fabs(qz) + R);
    const double Mv = fabs(e0) + (fabs(e1) + fabs(e2)), Mf = fabs(f0) + (fabs(f1) + fabs(f2));
    const double Sg = fma(Md, fma(Md, Mf, Mn * Mv), Mn * Mn);
    band32 = Sg >= 1e-6 ? (float)(1.4e-6 * Sg) * 1.000001f + 1e-30f : __builtin_inff();
  }
  typedef __attribute__((address_space(3))) uint32_t LdsU32;
  typedef __attribute__((address_space(3))) unsigned char LdsByte;
  const uint32_t row_bytes = (uint32_t)d.n_phi * 4u;
  LdsByte* bin = reinterpret_cast<LdsByte*>((__attribute__((address_space(3))) void*)hist) + (uint32_t)j * 4u + (uint32_t)ilo * row_bytes;
#if !ISX_WALK4
  const RowX* rp = rowx + ilo;
#endif
  const isx_f2 vAl = {cAl, cAl}, vBe = {cBe, cBe}, vVz = {cVz, cVz}, vQz = {cQz, cQz}, vE0 = {cE0, cE0}, vE1 = {cE1, cE1},
               vE2 = {cE2, cE2}, vF0 = {cF0, cF0}, vF1 = {cF1, cF1}, vF2 = {cF2, cF2};
  if (len > 0) ISX_DIAG_ADD_LANES(7, len);
  ISX_DIAG_ADD(11, 1);
  ISX_BD_MARK(sq, 4);
#if ISX_ABL == 1
  if (band32 == 12345.f) hist[0] = (uint32_t)(cAl + cBe + cVz + cQz + cE0 + cE1 + cE2 + cF0 + cF1 + cF2);
  return;
#endif
#if ISX_WALK4
  // Two row pairs per step: rows k, k+1 (table entry rp) and k+2, k+3 (entry rp + 2) as two INDEPENDENT packed chains, so
  // that a wave has two LDS reads and two dependent chains in flight instead of one (the walk is latency-bound at four
  // waves per SIMD) and pays the loop control once per four candidates.  Rows past the slot's last one are masked (their
  // table entry may be one past the table and their bin up to three rows past the grid: harmless reads, atomic adds of zero).
  // (the row table as two arrays -- {S_i, S_i+1, C_i, C_i+1} at 16 bytes per row, {T_i, T_i+1} at 8 -- instead of one 32-byte
  //  entry: a 16-byte read then spreads over sixteen bank groups instead of eight, and the ablation that reads one entry for
  //  all lanes says that bank conflicts of these reads cost 1.1 of the kernel's 7.0 ms)
  const float4* rowA = reinterpret_cast<const float4*>(rowx);
  const float2* rowB = reinterpret_cast<const float2*>(rowA + (d.n_theta + 4));
  auto classify = [&](int ie, int kk, bool& h0, bool& h1) {
    const float4 sc = rowA[ie];
    const float2 tt = rowB[ie];
    const bool one = kk < len, two = kk + 1 < len;
    const isx_f2 SS = {sc.x, sc.y}, CC = {sc.z, sc.w}, TT = {tt.x, tt.y};
    const isx_f2 dot = __builtin_elementwise_fma(vAl, SS, vVz * CC);
    const isx_f2 num = __builtin_elementwise_fma(vBe, SS, __builtin_elementwise_fma(vQz, CC, TT));
    const isx_f2 mdv = __builtin_elementwise_fma(vE1, SS, __builtin_elementwise_fma(vE2, CC, vE0));
    const isx_f2 ddw = __builtin_elementwise_fma(vF1, SS, __builtin_elementwise_fma(vF2, CC, vF0));
    const isx_f2 g = __builtin_elementwise_fma(dot, __builtin_elementwise_fma(dot, ddw, num * mdv), num * num);
    bool hit0 = g.x < 0.f, hit1 = g.y < 0.f;
    if (one && !(fminf(fabsf(g.x), fabsf(g.y)) > band32)) {
      // What tier 1 cannot decide is appended to the wave's deferred list (flush_deferred) and counts as a miss here; if the list
      // is full -- a slot whose band is infinite sends every candidate this way -- it is decided on the spot.  (Positions from a
      // wave-uniform count, ballot + mbcnt, instead of the LDS atomic: 6.77 against 6.42 ms -- two more compares and ballots in
      // EVERY step.)
#pragma unroll 1
      for (int t = 0; t < 2; ++t) {
        const float gt = t == 0 ? g.x : g.y;
        if (fabsf(gt) > band32 || (t == 1 && !two)) continue;
#ifndef ISX_DIAG_TIMING_ONLY
        ISX_DIAG_ADD_LANES(12, 1);
#endif
        int ir = ilo + kk + t;
        asm volatile("" : "+v"(ir));
        bool hit = false;
        typedef __attribute__((address_space(3))) uint32_t LdsCnt;
#if defined(ISX_DIAG) && !defined(ISX_DIAG_TIMING_ONLY)
        const uint32_t pos = (uint32_t)kDeferCap;   // (the tuning build re-decides every candidate right here: nothing is put aside)
#else
        const uint32_t pos = __hip_atomic_fetch_add(reinterpret_cast<LdsCnt*>(sq.defer), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
        if (pos < (uint32_t)kDeferCap) {
          ((volatile LdsWord*)sq.defer)[4u + pos] = (uint32_t)line | ((uint32_t)j << 8) | ((uint32_t)ir << 16);
        } else {
          int jr = j;
          asm volatile("" : "+v"(jr));
          hit = decide_exact(d, rowt, colx, src6 + (jr - j), ir, jr);
        }
        if (t == 0) hit0 = hit; else hit1 = hit;
      }
    }
    h0 = hit0 && one;
    h1 = hit1 && two;
#if defined(ISX_DIAG) && !defined(ISX_DIAG_TIMING_ONLY)
    {   // tuning builds: a decision taken by tier 1 must be the reference's
      V3 P, V;
      load_line(src6, P, V);
      for (int t = 0; t < (two ? 2 : (one ? 1 : 0)); ++t) {
        const bool ref = check_intersection(d.table + 6 * (size_t)((ilo + kk + t) * d.n_phi + j), d.half_w2, P, V);
        if (ref != (t == 0 ? h0 : h1)) ISX_DIAG_ADD_LANES(14, 1);
      }
    }
#endif
  };
  int ie = ilo;
  for (int k = 0;; k += 4) {
    const bool act = k < len;
    if (__ballot(act) == 0ull) break;
#ifndef ISX_DIAG_TIMING_ONLY
    ISX_DIAG_ADD(4, 4);
#endif
    if (act) {
      bool a0, a1, b0, b1;
#if ISX_ABL == 5
      classify((k & 3), k, a0, a1);
      classify(2 + (k & 3), k + 2, b0, b1);
#else
      classify(ie, k, a0, a1);
      classify(ie + 2, k + 2, b0, b1);
#endif
      ie += 4;
#ifdef ISX_DIAG_MULT
      // (north_star's "ballot/shuffle to coalesce same-bin writes", measured before it is built: of the lanes of a wave that add
      //  to a bin in one ds_add_u32, how many name a bin that a LOWER lane of the same instruction names as well?
      //  g_diag[8]: lanes that add 1, g_diag[9]: those of them that are not the first lane on their bin)
      {
        const bool hs[4] = {a0, a1, b0, b1};
        for (int t = 0; t < 4; ++t) {
          const uint32_t addr = (uint32_t)(uintptr_t)(bin + (uint32_t)t * row_bytes);
          const unsigned long long hm = __ballot(hs[t]);
          bool dup = false;
          unsigned long long mm = hm;
          while (mm) {
            const int sl = __builtin_ctzll(mm);
            mm &= mm - 1ull;
            const uint32_t a2 = (uint32_t)__builtin_amdgcn_readlane((int)addr, sl);
            if (hs[t] && sl < lane && a2 == addr) dup = true;
          }
          const int n_add = (int)__popcll(hm), n_dup = (int)__popcll(__ballot(dup));   // (ballots outside the macro's `if (lane == 0)`)
          ISX_DIAG_ADD(8, n_add); ISX_DIAG_ADD(9, n_dup);
        }
      }
#endif
#if ISX_ATOM_BRANCH
      // (variant: only the lanes that hit take part in the atomic -- fewer lanes in the LDS bank arbitration, four exec-mask branches)
      if (a0) __hip_atomic_fetch_add(reinterpret_cast<LdsU32*>(bin), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (a1) __hip_atomic_fetch_add(reinterpret_cast<LdsU32*>(bin + row_bytes), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (b0) __hip_atomic_fetch_add(reinterpret_cast<LdsU32*>(bin + 2u * row_bytes), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (b1) __hip_atomic_fetch_add(reinterpret_cast<LdsU32*>(bin + 3u * row_bytes), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
#if ISX_ABL == 4
      if (a0 && a1 && b0 && b1 && band32 == 12345.f)
#endif
      __hip_atomic_fetch_add(reinterpret_cast<LdsU32*>(bin), a0 ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#if ISX_ABL == 4
      if (a0 && a1 && b0 && b1 && band32 == 12345.f) {
#endif
      __hip_atomic_fetch_add(reinterpret_cast<LdsU32*>(bin + row_bytes), a1 ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(reinterpret_cast<LdsU32*>(bin + 2u * row_bytes), b0 ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(reinterpret_cast<LdsU32*>(bin + 3u * row_bytes), b1 ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#if ISX_ABL == 4
      }
#endif
#endif
      bin += 4u * row_bytes;
    }
  }
  __builtin_amdgcn_wave_barrier();
#else
  for (int k = 0;; k += 2) {
    const bool act = k < len;
    if (__ballot(act) == 0ull) break;
#ifndef ISX_DIAG_TIMING_ONLY
    ISX_DIAG_ADD(4, 2);
#endif
    if (act) {
      const float4 sc = *reinterpret_cast<const float4*>(rp);
      const float2 tt = *reinterpret_cast<const float2*>(&rp->T0);
      const bool two = k + 1 < len;
      const isx_f2 SS = {sc.x, sc.y}, CC = {sc.z, sc.w}, TT = {tt.x, tt.y};
      const isx_f2 dot = __builtin_elementwise_fma(vAl, SS, vVz * CC);
      const isx_f2 num = __builtin_elementwise_fma(vBe, SS, __builtin_elementwise_fma(vQz, CC, TT));
      const isx_f2 mdv = __builtin_elementwise_fma(vE1, SS, __builtin_elementwise_fma(vE2, CC, vE0));
      const isx_f2 ddw = __builtin_elementwise_fma(vF1, SS, __builtin_elementwise_fma(vF2, CC, vF0));
      const isx_f2 g = __builtin_elementwise_fma(dot, __builtin_elementwise_fma(dot, ddw, num * mdv), num * num);
      bool hit0 = g.x < 0.f, hit1 = g.y < 0.f;
      if (!(fminf(fabsf(g.x), fabsf(g.y)) > band32)) {
        // tiers 2 and 3 of walk_columns (binary64 about the original point with its 2.1e-9 band, then the reference's own test)
#pragma unroll 1
        for (int t = 0; t < 2; ++t) {
          const float gt = t == 0 ? g.x : g.y;
          if (fabsf(gt) > band32 || (t == 1 && !two)) continue;
#ifndef ISX_DIAG_TIMING_ONLY
          ISX_DIAG_ADD_LANES(12, 1);
#endif
          int ir = ilo + k + t;
          asm volatile("" : "+v"(ir));
          bool hit = false;
          {
            const double sd = rowt[4 * ir + 0], cd = rowt[4 * ir + 1], zz = rowt[4 * ir + 2], ad = rowt[4 * ir + 3];
            int jr = j;
            asm volatile("" : "+v"(jr));
            const double cph = colx[jr].c, sph = colx[jr].s;
            V3 P, V;
            {
              load_line(src6 + (jr - j), P, V);
            }
            const double pz = P.z - zz;
            const double dotd = fma(sd * V.y, cph, fma(-(sd * V.x), sph, -(cd * V.z)));
            const double numd = fma(sd * P.y, cph, fma(-(sd * P.x), sph, -(cd * pz)));
            const double m2dv = fma(2.0 * (ad * V.x), cph, fma(2.0 * (ad * V.y), sph, -2.0 * fma(P.x, V.x, fma(P.y, V.y, pz * V.z))));
            const double f0 = fma(P.x, P.x, fma(P.y, P.y, fma(ad, ad, pz * pz)));
            const double f1c = -2.0 * (ad * P.x), f2c = -2.0 * (ad * P.y);
            const double ddwd = fma(f1c, cph, fma(f2c, sph, f0 - d.half_w2));
            const double diff = fma(dotd, fma(dotd, ddwd, numd * m2dv), numd * numd);
            const double bandc = 2.1e-9 * fma(2.0, f0 + (fabs(f1c) + fabs(f2c)), d.half_w2);
            hit = diff < 0.0;
            if (fabs(dotd) < 1e-4 || fabs(diff) <= bandc) {
              ISX_DIAG_ADD_LANES(13, 1);
              const double* tab = d.table;
              asm volatile("" : "+v"(tab));
              hit = check_intersection(tab + 6 * (size_t)(ir * d.n_phi + jr), d.half_w2, P, V);
            }
          }
          if (t == 0) hit0 = hit; else hit1 = hit;
        }
      }
      hit1 = hit1 && two;
#if defined(ISX_DIAG) && !defined(ISX_DIAG_TIMING_ONLY)
      {   // tuning builds: a decision taken by tier 1 must be the reference's
        V3 P, V;
        load_line(src6, P, V);
        for (int t = 0; t < (two ? 2 : 1); ++t) {
          const bool ref = check_intersection(d.table + 6 * (size_t)((ilo + k + t) * d.n_phi + j), d.half_w2, P, V);
          if (ref != (t == 0 ? hit0 : hit1)) ISX_DIAG_ADD_LANES(14, 1);
        }
      }
#endif
      rp += 2;
      // (0 or 1 added without a branch: two exec-mask branches less per step, 7.30 -> 7.02 ms.  The masked half of an odd slot's last
      //  pair adds 0 one row further -- at most the row past the grid, i.e. the first n_phi words of the row table that follows the
      //  bins in LDS: an atomic add of zero, no bit changes)
      __hip_atomic_fetch_add(reinterpret_cast<LdsU32*>(bin), hit0 ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(reinterpret_cast<LdsU32*>(bin + row_bytes), hit1 ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      bin += 2u * row_bytes;
    }
  }
#endif
  ISX_BD_MARK(sq, 5);
}

template <class D>
__device__ __forceinline__ void drain_cols(const D& d, uint32_t* __restrict__ hist, const double* __restrict__ rowt,
                                           const ColX* __restrict__ colx, const RowX* __restrict__ rowx,
                                           const double* __restrict__ lines, const SlotQueues& sq, int least, int lane) {
  volatile LdsInt* tl = sq.tail;
  volatile LdsInt* hd = sq.head;
  const int held = lane < kClasses ? tl[lane] - hd[lane] : 0;
  unsigned long long m = __ballot(held >= least);
  while (m) {
    const int c = __builtin_ctzll(m);
    m &= m - 1ull;
    const int h = hd[c], n = tl[c] - h;
    const int take = n < 64 ? n : 64;
    const uint32_t rec = lane < take ? ((volatile LdsWord*)sq.q)[c * kQueueCap + ((h + lane) & (kQueueCap - 1))] : 0u;
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) hd[c] = h + take;
    __builtin_amdgcn_wave_barrier();
    consume_cols(d, hist, rowt, colx, rowx, lines, rec, lane < take, lane, sq);
  }
}

// every lane hands in (at most) one column slot: rows [ilo, ilo + cnt) of column j for line `line` of the unit
template <class D>
__device__ __forceinline__ void push_cols(const D& d, uint32_t* __restrict__ hist, const double* __restrict__ rowt,
                                          const ColX* __restrict__ colx, const RowX* __restrict__ rowx,
                                          const double* __restrict__ lines, const SlotQueues& sq, int line, int j, int ilo, int cnt,
                                          int lane) {
  int i0 = ilo, rem = cnt;
  ISX_BD_MARK(sq, 1);
  for (;;) {
    const int piece = rem > kColLongest ? kColPiece : rem;
    if (piece > 0) {
      const int c = col_class(piece);
      const int pos = __hip_atomic_fetch_add(sq.tail + c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      ((volatile LdsWord*)sq.q)[c * kQueueCap + (pos & (kQueueCap - 1))] =
          (uint32_t)line | ((uint32_t)j << 8) | ((uint32_t)i0 << 16) | ((uint32_t)piece << 24);
    }
    rem -= piece;
    i0 += piece;
    __builtin_amdgcn_wave_barrier();
    ISX_BD_MARK(sq, 2);
    drain_cols(d, hist, rowt, colx, rowx, lines, sq, 64, lane);
    if (__ballot(rem > 0) == 0ull) break;
  }
}

// producer: the columns of the caps of a batch of 64 lines, packed over the lanes (owner search as in walk_lines_packed).
// Called once per pass of the batch: the two caps of a line that takes both cap passes never share a bin (caps_may_touch); `band`:
// the lines of the pass are grazing lines (prep_band), the rows of a column are those of the cap AND the band.
template <class D>
__device__ __forceinline__ void produce_cols_packed(const D& d, uint32_t* __restrict__ hist, const double* __restrict__ rowt,
                                                    const ColX* __restrict__ colx, const RowX* __restrict__ rowx,
                                                    const double* __restrict__ lines, const SlotQueues& sq, const ColPre& pre,
                                                    int excl, int incl, int total, float inv_dth, int n_theta, int first_line,
                                                    int lane, LdsInt* mark, const BandPre& bp, bool band) {
#pragma unroll 1
  for (int base = 0; base < total; base += 64) {
    const int g = base + lane;
    bool have = g < total;
    ISX_BD_MARK(sq, 9);
    volatile LdsInt* mk = mark;
    mk[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    if (pre.ncol > 0 && excl < base + 64 && incl > base) mk[(excl > base ? excl : base) - base] = lane + 1;
    __builtin_amdgcn_wave_barrier();
    const int m = mk[lane];
    const unsigned long long low = __ballot(m != 0) & (~0ull >> (63 - lane));
    const int pos = 63 - __builtin_clzll(low | 1ull);
    int owner = mk[pos] - 1;
    if (!have || owner < 0) { owner = 0; have = false; }
    ISX_BD_MARK(sq, 6);
    const int o_excl = __shfl(excl, owner, 64), o_jlo = __shfl(pre.jlo, owner, 64);
    const float fx = __shfl(pre.fx, owner, 64), fy = __shfl(pre.fy, owner, 64), a = __shfl(pre.a, owner, 64), cw = __shfl(pre.cosw, owner, 64);
    ISX_BD_MARK(sq, 7);
    int j = o_jlo + (g - o_excl);
    if (j >= d.n_phi) j -= d.n_phi;
    if (!have) j = 0;
    int ilo = 0, cnt = 0, ilo_b = 0, cnt_b = 0;
    const float c32 = colx[j].c32, s32 = colx[j].s32;
    if (have) cap_rows(fx, fy, a, cw, c32, s32, inv_dth, n_theta, ilo, cnt);
    if (band) {
      BandPre ob;
      ob.ux = __shfl(bp.ux, owner, 64); ob.uy = __shfl(bp.uy, owner, 64); ob.uz = __shfl(bp.uz, owner, 64); ob.kap = __shfl(bp.kap, owner, 64);
      const int r_lo = ilo, r_hi = ilo + cnt - 1;
      band_rows(ob, c32, s32, inv_dth, r_lo, have ? r_hi : r_lo - 1, ilo, cnt, ilo_b, cnt_b);
    }
    const int npiece = (band && __ballot(cnt_b > 0) != 0ull) ? 2 : 1;
#pragma unroll 1
    for (int w = 0; w < npiece; ++w)
      push_cols(d, hist, rowt, colx, rowx, lines, sq, first_line + owner, j, w == 0 ? ilo : ilo_b, w == 0 ? cnt : cnt_b, lane);
  }
}

// ISX_HITLINE_ORIGIN_COMPAT (fluxAtObserverFast.C:1181-1201,1285): start (0,0,0), direction last/|last|,
// normalised twice (once by hand, once by the ARay constructor); plain ops as in the reference.
__device__ __forceinline__ void hit_line_compat(V3& P, V3& V) {
  const double dx = P.x - 0.0, dy = P.y - 0.0, dz = P.z - 0.0;
  const double mag = sqrt(dx * dx + dy * dy + dz * dz);
  const double ex = dx / mag, ey = dy / mag, ez = dz / mag;
  const double m2 = sqrt(ex * ex + ey * ey + ez * ez);
  P.x = 0.0; P.y = 0.0; P.z = 0.0;
  V.x = ex / m2; V.y = ey / m2; V.z = ez / m2;
}

// ------------------------------------------------------------------ per-lane ray state
struct Ray {
  V3 p, v;
  V3 prev;          // start of the current segment (only kept for SINK_DISC)
  uint32_t ido;     // bits 0..30: ray index relative to the wave's (or launch's) first ray, id = id_base + offset();
                    // bit 31: set once the ray has been re-scattered (source_model 1) -- Philox stream 2 instead of 0.
                    // (one register for both: the BRDF kernel sits exactly at the 128-VGPR limit of a 1024-thread block)
  uint32_t j;       // mirror interactions of the current trace; track points = j + 1 (+1 once it left the box)
  int on;
  bool tgt;         // ISX_TRACE_CHORD: v holds the next wall point T, not a direction
  uint32_t k;       // lobe pipeline (assist_body<.., SURF_LOBE>): 0, or 1 + the index of the NEXT try of the rejection sampler of the
                    // interaction the ray is in (it sits at its new point p, v is still the old direction); unused elsewhere
  uint32_t cw[4];   // the Philox block this lane holds (bounce_words; unused with PH_DIRECT)
  __device__ __forceinline__ uint32_t offset() const { return ido & 0x7fffffffu; }
  __device__ __forceinline__ uint32_t stream() const { return (ido >> 31) << 1; }   // 0 primary, 2 scattered
  __device__ __forceinline__ bool scattered() const { return (ido >> 31) != 0u; }
};

template <class G>
__device__ __forceinline__ void ray_start(const G& g, Ray& r, uint32_t ido) {
  r.ido = ido; r.j = 0; r.on = K_NONE; r.tgt = false; r.k = 0;
  r.p.x = g.src[0]; r.p.y = g.src[1]; r.p.z = g.src[2];
  r.v.x = g.dir0[0]; r.v.y = g.dir0[1]; r.v.z = g.dir0[2];
}

// Second half of a step, once the boundary (kind, q) is known: advance, interact.
// CH: 0 explicit bounces only, 1 chord identity for every eligible bounce (compile time), 2 decided by h.chord.
// Returns 0 while running, else the end status of the CURRENT trace.
// SET_ON = false (the tracer waves of assist_body, which only ever arrive on the inner sphere and tell a fresh ray by its
// interaction count): Ray::on is not maintained -- one move and one compare less per bounce.
template <bool KEEP_PREV, bool LEAN, int CH, int PH = PH_DIRECT, bool SET_ON = true, int SURF = (LEAN ? SURF_LAMBERT : SURF_ANY), class G>
__device__ __forceinline__ int ray_arrive(const Hot& h, const G& g, Ray& r, uint64_t seed, uint64_t id_base, int kind, const V3& q) {
  const uint64_t rid = id_base + (uint64_t)r.offset();
  if (KEEP_PREV) r.prev = r.p;
  r.p = q;
  if (kind == K_BOX) { r.on = K_BOX; return ST_EXITED; }
  if (SET_ON) r.on = kind;
  bool alive;
  uint32_t wa, wb;
  bounce_words<PH>(seed, rid, r.j, r.stream(), r.cw, wa, wb);
  const bool eligible = (kind == K_INNER) && (SURF == SURF_LAMBERT || (SURF == SURF_ANY && h.lambertian && h.surface_model == 0));
  if (CH != 0 && eligible && (CH == 1 || h.chord)) {
    alive = interact_chord(h, r.v, wa, wb);
    r.tgt = alive;
  } else {
    alive = interact<LEAN, SURF>(h, g, kind, q, r.v, seed, rid, r.j, r.stream(), wa, wb);
  }
  r.j++;
  if (!alive) return ST_ABSORBED;
  if ((int)r.j + 1 > h.limit) return ST_SUSPENDED;  // npoints = j + 1 after this interaction
  return 0;
}

// Chord-mode arrival: the lane holds a target point T in r.v.  Returns true if T is on the mirror patch (then
// q = T, and r.v becomes the un-normalised last chord); false if T lies in the port opening (then r.v becomes
// the unit direction P->T and the generic boundary search takes over).
__device__ __forceinline__ void chord_leave(Ray& r) {   // T lies in the port opening: unit direction P -> T
  V3 d;
  d.x = r.v.x - r.p.x; d.y = r.v.y - r.p.y; d.z = r.v.z - r.p.z;
  const double mag = sqrt(dot3(d, d));
  r.v.x = d.x / mag; r.v.y = d.y / mag; r.v.z = d.z / mag;
  r.tgt = false;
}
// DEFER: the persistent kernels leave the lane as it is (r.tgt stays set, r.v keeps T) and form the direction when the
// parked lanes are flushed -- the IEEE sqrt and three divides of chord_leave ran in ~36 % of all wave-steps for the 0.8 % of
// the lanes that need them (chord mode was 2 % faster than explicit bounces, not the 10 % its shorter step promises).
template <bool DEFER = false>
__device__ __forceinline__ bool chord_arrive(const Hot& h, Ray& r, V3& q) {
  if (r.v.z >= h.zcut_in) {
    r.tgt = false;
    q = r.v;
    if (!DEFER) {   // the last chord P -> T as the ray's direction: only the end-state interface reports the direction of a ray
                    // that is absorbed at T; the persistent kernels (DEFER) never look at it, and forming it cost them three
                    // subtractions and a register shuffle per bounce
      V3 d;
      d.x = r.v.x - r.p.x; d.y = r.v.y - r.p.y; d.z = r.v.z - r.p.z;
      r.v = d;
    }
    return true;
  }
  if (!DEFER) chord_leave(r);
  return false;
}

// One full step: next boundary + interaction (chord mode decided at run time).
template <bool KEEP_PREV, class G>
__device__ __forceinline__ int ray_step(const Hot& h, const G& g, Ray& r, uint64_t seed, uint64_t id_base) {
  V3 q;
  int kind;
  if (r.tgt) {   // (a chord whose end point lies in the port opening leaves along its unit direction: generic search)
    if (chord_arrive(h, r, q)) kind = K_INNER;
    else kind = next_hit_generic(g, r.p, r.v, r.on, q);
  } else kind = next_hit(h, g, r.p, r.v, r.on, q);
  return ray_arrive<KEEP_PREV, false, 2>(h, g, r, seed, id_base, kind, q);
}

// nonLambertianFlux.C:253-268: restart from the primary's last point along a BRDF-sampled direction
template <class G>
__device__ __forceinline__ void ray_rescatter(const G& g, Ray& r, uint64_t seed, uint64_t id_base) {
  // the held Philox block belongs to the finished primary trace: dead from here on (the scattered trace starts at j = 0,
  // whose even step draws its own block); saying so frees four VGPRs across brdf_sample
  r.cw[0] = r.cw[1] = r.cw[2] = r.cw[3] = 0u;
  V3 d0; d0.x = g.dir0[0]; d0.y = g.dir0[1]; d0.z = g.dir0[2];
  const V3 normal = tv_unit(r.p);
  const V3 nd = brdf_sample(g, normal, d0, seed, id_base + (uint64_t)r.offset());
  const double mag = sqrt(nd.x * nd.x + nd.y * nd.y + nd.z * nd.z);
  r.v.x = nd.x / mag; r.v.y = nd.y / mag; r.v.z = nd.z / mag;
  r.on = (r.on == K_BOX) ? K_NONE : r.on;
  r.j = 0; r.ido |= 0x80000000u; r.tgt = false;
}

// ------------------------------------------------------------------ physical disc test (SINK_DISC)
// forward segment [0,tmax] of p+t*v enters the tube {|axial|<=h, radial<=r} about centre c, unit axis a.
// Replaces AFocalSurface(TGeoTube) + isRayHittingDetector (integratingSphereDetectorSweep.C:134-172).
__device__ __forceinline__ bool segment_hits_tube(const V3& p, const V3& v, double tmax, const double* __restrict__ ca,
                                                  double r, double h) {
  V3 c, a, w;
  c.x = ca[0]; c.y = ca[1]; c.z = ca[2]; a.x = ca[3]; a.y = ca[4]; a.z = ca[5];
  w.x = p.x - c.x; w.y = p.y - c.y; w.z = p.z - c.z;
  // cull (never decides a hit): the tube lies inside the ball of radius sqrt(r^2+h^2) about c, so a line that passes
  // farther from c misses it.  dist^2 = (|w|^2 |v|^2 - (w.v)^2)/|v|^2; the cancellation error is ~1e-11 here, the
  // margin 1e-6 relative.  Most of the 362 discs of a sweep leave through this door.
  {
    const double wv = dot3(w, v), ww = dot3(w, w), vv = dot3(v, v);
    if (fma(ww, vv, -(wv * wv)) > fma(r, r, h * h) * vv * 1.000001 + 1e-9) return false;
  }
  const double ws = dot3(w, a), vs = dot3(v, a);
  double t0 = 0.0, t1 = tmax;
  if (vs != 0.0) {
    double ta = (-h - ws) / vs, tb = (h - ws) / vs;
    if (ta > tb) { const double tmp = ta; ta = tb; tb = tmp; }
    if (ta > t0) t0 = ta;
    if (tb < t1) t1 = tb;
  } else if (fabs(ws) > h) return false;
  if (t0 > t1) return false;
  const double A = dot3(v, v) - vs * vs;
  const double B = dot3(w, v) - ws * vs;
  const double C = dot3(w, w) - ws * ws - r * r;
  if (A <= 0.0) return C <= 0.0;
  const double D = fma(B, B, -(A * C));
  if (D < 0.0) return false;
  const double sD = sqrt(D);
  const double ra = (-B - sD) / A, rb = (-B + sD) / A;
  if (ra > t0) t0 = ra;
  if (rb < t1) t1 = rb;
  return t0 <= t1;
}

// the same with the disc entry read from the device's disc list (global memory, whatever pointer type reached this point)
__device__ __forceinline__ bool segment_hits_tube_g(const V3& p, const V3& v, double tmax, const double* __restrict__ ca_global,
                                                    double r, double h) {
  double ca[6];
  load6(ca_global, ca);
  return segment_hits_tube(p, v, tmax, ca, r, h);
}

template <class DG>
__device__ __forceinline__ void bin_discs(const DG& dd, uint32_t* __restrict__ hist, const V3& P0, const V3& P1,
                                              const V3& V, int lane) {
  struct { int nbins; const double* discs; double disc_r, disc_h; } d;
  d.nbins = dd.nbins; d.discs = dd.discs; d.disc_r = dd.disc_r; d.disc_h = dd.disc_h;
  V3 dl; dl.x = P1.x - P0.x; dl.y = P1.y - P0.y; dl.z = P1.z - P0.z;
  const double tmax = dot3(dl, V);
  for (int b0 = 0; b0 < d.nbins; b0 += 64) {
    const int b = b0 + lane;
    bool hit = false;
    if (b < d.nbins) hit = segment_hits_tube_g(P0, V, tmax, d.discs + 6 * (size_t)b, d.disc_r, d.disc_h);
    if (hit) atomicAdd(&hist[b], 1u);
  }
}

// compile-time loop: f(integral_constant<int, K>) for K = FIRST .. N-1 (the step index selects code, not just data)
template <int FIRST, int N, class F>
__device__ __forceinline__ void static_steps(F&& f) {
  if constexpr (FIRST < N) {
    f(std::integral_constant<int, FIRST>());
    static_steps<FIRST + 1, N>(f);
  }
}

// ------------------------------------------------------------------ persistent trace kernel, one per sink
//   SINK_FLUX: 180x90 detector flux map (the headline path)
//   SINK_DZ  : histogram of the exit direction's z component (distributionSphereDetectorSweep.C:54,91)
//   SINK_DISC: physical disc sweep (integratingSphereDetectorSweep.C)
template <int SINK, bool LEAN = false, int CH = 2, bool RESC = !LEAN>
__device__ __forceinline__ void persistent_body(const Geom& g_arg, const DetGrid& d_arg, const Work& wk) {
  extern __shared__ __align__(16) unsigned char smem[];
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
  const int nbins = d_arg.nbins;
  const size_t off_row = ((size_t)nbins * 4 + 15) & ~(size_t)15;
  double* rowt = reinterpret_cast<double*>(smem + off_row);
  ColX* colx = reinterpret_cast<ColX*>(rowt + (SINK == SINK_FLUX ? 4 * d_arg.n_theta : 0));
  unsigned long long* sstat = reinterpret_cast<unsigned long long*>(colx + (SINK == SINK_FLUX ? 2 * d_arg.n_phi : 0));
  // LDS copies of the parameter blocks: rare paths read them on demand (volatile), the hot loop
  // keeps only `Hot` + a few scalars in SGPRs.
  Geom* g_lds = reinterpret_cast<Geom*>(sstat + 8);
  DetGrid* d_lds = reinterpret_cast<DetGrid*>(g_lds + 1);

  const int tid = threadIdx.x;
  // the workgroup size is the launch's (1024 threads for the kernels that keep the LDS histogram, 512 for the trace-only ones,
  // which then fit 6 waves per SIMD: isx_api.hip block_for())
  const int nthr = (int)blockDim.x, wpb = nthr >> 6;
  for (int b = tid; b < nbins; b += nthr) hist[b] = 0u;
  if (SINK == SINK_FLUX) {
    for (int b = tid; b < 4 * d_arg.n_theta; b += nthr) rowt[b] = d_arg.rowtab[b];
    for (int b = tid; b < 2 * d_arg.n_phi; b += nthr) {
      const int j = b < d_arg.n_phi ? b : b - d_arg.n_phi;
      ColX e;
      e.c = d_arg.coltab[2 * j]; e.s = d_arg.coltab[2 * j + 1]; e.off4 = (uint32_t)j * 4u; e.c32 = (float)e.c; e.s32 = (float)e.s; e.pad = 0u;
      colx[b] = e;
    }
  }
  if (tid < 8) sstat[tid] = 0ull;
  if (tid == (nthr > 64 ? 64 : 0)) {
    *g_lds = g_arg;
    // the first boundary of every fresh ray (Geom::q0): one evaluation per workgroup instead of one per ray
    const Hot h0 = make_hot(g_arg);
    V3 s0, d0, q0;
    s0.x = g_arg.src[0]; s0.y = g_arg.src[1]; s0.z = g_arg.src[2];
    d0.x = g_arg.dir0[0]; d0.y = g_arg.dir0[1]; d0.z = g_arg.dir0[2];
    q0 = s0;
    const bool ok = next_hit_s1<true>(h0, g_arg, s0, d0, K_NONE, q0);
    g_lds->q0[0] = q0.x; g_lds->q0[1] = q0.y; g_lds->q0[2] = q0.z;
    g_lds->q0_ok = ok ? 1 : 0;
  }
  if (tid == (nthr > 128 ? 128 : 0)) *d_lds = d_arg;
  __syncthreads();
  typedef __attribute__((address_space(3))) Geom LdsGeom;        // explicit LDS address space: ds_read, not flat_load
  typedef __attribute__((address_space(3))) DetGrid LdsDetGrid;
  const volatile LdsGeom& g = *(const volatile LdsGeom*)g_lds;
  const volatile LdsDetGrid& d = *(const volatile LdsDetGrid*)d_lds;
  const Hot h = make_hot(g_arg);
  const double portz = d_arg.portz;
  const uint64_t seed = wk.seed;

  const int lane = tid & 63;
  // the part of the launch's ray range this wave is refilling from: offsets [next, end) from wk.first (wave-uniform: SGPRs);
  // next = kDry once the launch's queue has nothing left (a launch has fewer than 2^31 rays)
  constexpr uint32_t kDry = 0xffffffffu;
  uint32_t next = 0, end = 0;
  const uint64_t range_first = wk.first;

  Ray r;
  ray_start(g, r, 0);
  r.prev = r.p;
  // lane state: `run` = has a ray that the hot path can advance; `parked` = has a ray that waits for the generic boundary
  // search; neither = no ray (never had one, or its ray ended)
  bool run = false, parked = false;
  uint32_t iter = 0;
  uint32_t n_wall = 0;                                                           // per lane: mirror interactions of its finished rays
  uint32_t n_exited = 0, n_counted = 0, n_susp = 0, n_ended = 0;                 // per wave (ballot counts: scalar registers)
  unsigned long long n_inc = 0;                                                  // per wave (SINK_LOG; the histogram sinks count at flush)
  uint32_t n_taken = 0;                                                          // per wave: rays taken off the queue
  uint32_t reg_slot = 0, reg_left = 0, reg_id = 0xffffffffu;                     // per wave (SINK_REC): cursor in the open region
  ISX_TD_DECL;

  for (;;) {
    // ---- refill dead lanes from this wave's range
    const unsigned long long dead = __ballot(!(run || parked));
    if (dead) {
      if (next == end) {   // this wave's sub-range is used up: the next one off the launch's queue
        const uint32_t n32 = (uint32_t)wk.n;
        const uint32_t share = (n32 - end) / (2u * (uint32_t)wpb * gridDim.x);   // (end: where the queue stood at the last visit)
        const uint32_t want = share >= wk.sub ? wk.sub : (share > 64u ? share : 64u);
        unsigned long long b64 = 0;
        if (lane == 0) b64 = atomicAdd(reinterpret_cast<unsigned long long*>(wk.ctr + Q_RAYS), (unsigned long long)want);
        const uint32_t bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(b64 >> 32));
        const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b64);
        if (bhi == 0u && b < n32) {
          next = b; end = n32 - b > want ? b + want : n32;
        } else next = kDry;
      }
      if (next < end) {
        const uint32_t left = end - next;
        const uint32_t rank =
            __builtin_amdgcn_mbcnt_hi((uint32_t)(dead >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dead, 0u));
        if (!(run || parked) && rank < left) {
          ray_start(g, r, next + rank);
          run = true;
        }
        const uint32_t want = (uint32_t)__popcll(dead);
        const uint32_t take = want < left ? want : left;
        ISX_TD_ADD(15, take);
        next += take; n_taken += take;
      }
      if (__ballot(run || parked) == 0ull) break;   // (nothing left in the queue either: a wave with dead lanes has just asked)
    }
    ISX_TD_MARK(0);
    ISX_TD_ADD(7, 1); ISX_TD_ADD(8, __popcll(__ballot(run))); ISX_TD_ADD(9, __popcll(__ballot(parked)));
    if (next == kDry) { ISX_TD_ADD(10, 1); ISX_TD_ADD(11, __popcll(__ballot(run))); }
    // ---- one boundary + interaction per live lane.  The hot boundary search (rule S1) runs every
    // iteration; the generic search (port transits, rim, box: ~0.75 % of lane-steps but ~40 % of
    // wave-iterations if run eagerly) is BATCHED: a lane that needs it parks until several lanes
    // need it, every 4th loop trip, or nothing else is left to do.  Scheduling only - a ray's
    // history never depends on it.
    bool bin_me = false;
    int pend = 0;     // end status of the ray this lane finished during the trip, 0 if none
    // what happens to a lane once its boundary (kind, q) is known
    // Philox block shared by two bounces (bounce_words): the lean kernels alternate compute / reuse steps
    constexpr bool kShare = LEAN;
    static_assert(!kShare || (kStepsPerTrip % 2) == 0, "block sharing needs an even number of steps per trip");
    auto arrive = [&](int kind, const V3& q, auto ph) {
      constexpr int PH = kShare ? decltype(ph)::value : PH_DIRECT;
      const int st = ray_arrive<SINK == SINK_DISC || SINK == SINK_DISCPOS, LEAN, CH, PH>(h, g, r, seed, range_first, kind, q);
      if (st != 0) { run = false; pend = st; }   // census / re-scatter once per trip (below), not per bounce
    };
    // hot boundary search of one lane: true if it arrived on the inner mirror patch, else the lane parks
    auto hot_search = [&](V3& q) -> bool {
      if (CH != 0 && r.tgt) return chord_arrive<true>(h, r, q);
      return next_hit_s1<false>(h, g, r.p, r.v, r.on, q);
    };
    {
      V3 q;
      int kind = K_NONE;
      bool arrived = false;
      if (run) {
        // a fresh ray (rays enter on step 0 only) goes straight to the launch's common first boundary (Geom::q0); a
        // re-scattered one that starts on the world box (on == K_NONE as well) is outside the ball: generic search
        const bool fresh = r.on == K_NONE && r.j == 0u && !r.scattered();
        if (fresh) {
          if (g.q0_ok) { q.x = g.q0[0]; q.y = g.q0[1]; q.z = g.q0[2]; kind = K_INNER; arrived = true; }
          else { parked = true; run = false; }
        } else if (hot_search(q)) { kind = K_INNER; arrived = true; }
        else { parked = true; run = false; }
      }
      ISX_TD_MARK(1);
      const unsigned long long pm = __ballot(parked);
      if (pm) {
        const int sched_min = g.sched_min, sched_mask = g.sched_mask;   // rare: read from the LDS copy
        const bool flush = ((int)__popcll(pm) >= sched_min) || ((iter & (uint32_t)sched_mask) == (uint32_t)sched_mask) ||
                           (__ballot(run) == 0ull);
        if (flush) { ISX_TD_ADD(13, 1); ISX_TD_ADD(14, __popcll(pm)); }
        if (flush && parked) {
          if (CH != 0 && r.tgt) chord_leave(r);   // a chord whose end point lies in the port opening (chord_arrive<DEFER>)
          else if (r.on == K_INNER) unit_dir(r.v);   // it left rule S1' (next_hit_s1): the generic search takes a unit direction
          kind = next_hit_generic(g, r.p, r.v, r.on, q);
          arrived = true;
          parked = false;
          run = true;
        }
      }
      iter++;
      ISX_TD_MARK(2);
      if (arrived) arrive(kind, q, std::integral_constant<int, PH_EVEN>());
      ISX_TD_MARK(3);
    }
    // extra bounces per loop trip (hot search only): amortises refill / flush / exit bookkeeping
    static_steps<1, kStepsPerTrip>([&](auto rep) {
      V3 q;
      bool arrived = false;
      if (run) {
        if (hot_search(q)) arrived = true;
        else { parked = true; run = false; }
      }
      if (arrived) arrive(K_INNER, q, std::integral_constant<int, (decltype(rep)::value & 1) ? PH_ODD : PH_EVEN>());
    });
    ISX_TD_MARK(4);
    // ---- census of the rays that ended in this trip (a dead lane stays dead until the next refill, so each ended
    // ray is seen exactly once, with its final point and direction still in place)
    if (RESC && pend != 0 && h.source_model == 1 && !r.scattered()) {
      // nonLambertianFlux.C:253-268: the primary trace is over (whatever its status); the ray restarts from its last
      // point along a BRDF-sampled direction.  One copy of this code per trip instead of one per bounce.
      n_wall += r.j;
      ray_rescatter(g, r, seed, range_first);
      run = true;
      pend = 0;
    }
    {
      const bool ended = pend != 0, exited = pend == ST_EXITED;
      const bool below = exited && (r.p.z < portz);   // isRayPassingThroughExitPort, fluxAtObserver.C:162-166
      if (ended) {
        n_wall += r.j;
        if (n_wall > 0x7fffffffu) { atomicAdd(&sstat[6], (unsigned long long)n_wall); n_wall = 0; }
      }
      bin_me = (SINK == SINK_DISC || SINK == SINK_DISCPOS) ? exited : below;
      const unsigned long long me = __ballot(ended);
      if (me) {
        n_ended += (uint32_t)__popcll(me);
        n_exited += (uint32_t)__popcll(__ballot(exited));
        n_counted += (uint32_t)__popcll(__ballot(below));
        n_susp += (uint32_t)__popcll(__ballot(pend == ST_SUSPENDED));
        ISX_TD_ADD(12, __popcll(me));
      }
    }
    ISX_TD_MARK(5);
    if (SINK == SINK_LOG) {
      // wave-aggregated append: one atomic on the cursor per wave-step, 32-byte records
      const unsigned long long m = __ballot(bin_me);
      if (m) {
        const int leader = __builtin_ctzll(m);
        unsigned long long base = 0;
        if (lane == leader) base = atomicAdd(d_arg.log_count, (unsigned long long)__popcll(m));
        base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(base >> 32), leader) << 32) |
               (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
        if (bin_me) {
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          const unsigned long long slot = base + rank;
          if (slot < d_arg.log_cap) {
            double4 rec;
            rec.x = __longlong_as_double((long long)(range_first + (uint64_t)r.offset())); rec.y = r.v.x; rec.z = r.v.y; rec.w = r.v.z;
            reinterpret_cast<double4*>(d_arg.log_rec)[slot] = rec;
          }
        }
        n_inc += (unsigned long long)__popcll(m);
      }
    } else if (SINK == SINK_REC) {
      const unsigned long long m = __ballot(bin_me);
      if (m) {
        const uint32_t cnt = (uint32_t)__popcll(m);
        if (cnt > reg_left) {   // close the open region, reserve the next one (kRegion above)
          if (lane == 0 && reg_id != 0xffffffffu) d_arg.rec_counts[reg_id] = kRegion - reg_left;
          uint32_t id = 0;
          if (lane == 0) id = atomicAdd(&wk.ctr[Q_REGIONS], 1u);
          reg_id = (uint32_t)__builtin_amdgcn_readfirstlane((int)id);
          reg_slot = 0; reg_left = kRegion;
        }
        if (bin_me) {
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          double2* dst = reinterpret_cast<double2*>(d_arg.rec_lines + 6ull * ((uint64_t)reg_id * kRegion + (uint64_t)(reg_slot + rank)));
          dst[0] = make_double2(r.p.x, r.p.y); dst[1] = make_double2(r.p.z, r.v.x); dst[2] = make_double2(r.v.y, r.v.z);
        }
        reg_slot += cnt; reg_left -= cnt;
      }
    } else if (SINK == SINK_PERPOS) {
      // per-lane: the ray's own detector group only (one or two exact tests)
      bool hit0 = false, hit1 = false;
      int b0 = 0, b1 = 0;
      if (bin_me) {
        const uint64_t map_first = d.map_first, rpg = d.rays_per_group;
        const double* table = d.table;
        const double half_w2 = d.half_w2;
        V3 lp = r.p, lv = r.v;
        if (d.hit_line_mode == 1) hit_line_compat(lp, lv);
        const uint64_t rel = (range_first + (uint64_t)r.offset()) - map_first;
        uint64_t grp = (uint64_t)((double)rel / (double)rpg);
        if (grp * rpg > rel) grp--;
        else if ((grp + 1) * rpg <= rel) grp++;
        if (d.fold == 2) {
          const int nphi = d.n_phi, half = nphi / 2;
          const int i = (int)(grp / (uint64_t)half), j = (int)(grp % (uint64_t)half);
          b0 = i * nphi + j;
          b1 = b0 + half;
          hit1 = check_intersection(table + 6 * (size_t)b1, half_w2, lp, lv);
        } else {
          b0 = (int)grp;
        }
        hit0 = check_intersection(table + 6 * (size_t)b0, half_w2, lp, lv);
      }
      if (hit0) atomicAdd(&hist[b0], 1u);
      if (hit1) atomicAdd(&hist[b1], 1u);
    } else if (SINK == SINK_DISCPOS) {
      // per-lane: the forward exit segment against the ray's own disc only
      bool hit = false;
      int b = 0;
      if (bin_me) {
        const uint64_t map_first = d.map_first, rpg = d.rays_per_group;
        const uint64_t rel = (range_first + (uint64_t)r.offset()) - map_first;
        uint64_t grp = (uint64_t)((double)rel / (double)rpg);
        if (grp * rpg > rel) grp--;
        else if ((grp + 1) * rpg <= rel) grp++;
        b = (int)grp;
        V3 dl; dl.x = r.p.x - r.prev.x; dl.y = r.p.y - r.prev.y; dl.z = r.p.z - r.prev.z;
        const double tmax = dot3(dl, r.v);
        hit = segment_hits_tube_g(r.prev, r.v, tmax, d.discs + 6 * (size_t)b, d.disc_r, d.disc_h);
      }
      if (hit) atomicAdd(&hist[b], 1u);
    } else if (SINK == SINK_DZ) {
      // per-lane: TH1D(nbins,-1,1)->Fill(dz)
      bool hit = false;
      int b = 0;
      if (bin_me) {
        const double f = (r.v.z + 1.0) * 0.5 * (double)nbins;
        b = (int)floor(f);
        hit = b >= 0 && b < nbins;
      }
      if (hit) atomicAdd(&hist[b], 1u);
    } else {
      // ---- wave-cooperative binning of every line that left in this trip
      unsigned long long em = __ballot(bin_me);
      constexpr bool kFast = (SINK == SINK_FLUX) && LEAN;   // lean kernels: hit line = last segment, no compat mode
      const int bin_mode = em ? d.bin_mode : 1;   // wave-uniform scalars of the binning: read from the LDS copy when needed
      const int hit_line_mode = (!LEAN && em) ? d.hit_line_mode : 0;
      const bool fast = kFast && em && d.rec_stage && bin_mode == 1;
      // per-wave staging of the exit-line records in LDS (16 B + 4 B per lane, after the parameter blocks)
      float4* rec4 = nullptr;
      int* reci = nullptr;
      LdsInt* spl = nullptr;
      if (fast && em) {
        // (offsets from `smem`, not pointer-to-integer casts: the accesses must stay ds_read/ds_write with 32-bit addresses)
        const uint32_t off_rec = ((uint32_t)(reinterpret_cast<unsigned char*>(d_lds + 1) - smem) + 15u) & ~15u;
        float4* base = reinterpret_cast<float4*>(smem + off_rec);
        // (the wave's slot is re-derived here, behind a compiler barrier, and portz re-read from the LDS copy: hoisted out of
        //  the trace loop these two loop invariants cost the BRDF kernel two VGPRs it does not have -> 12 B/lane of scratch)
        uint32_t wslot = (uint32_t)tid >> 6;
        asm volatile("" : "+v"(wslot));
        rec4 = base + wslot * 64u;
        reci = reinterpret_cast<int*>(base + wpb * 64) + wslot * 64u;
        spl = (LdsInt*)(reinterpret_cast<int*>(base + wpb * 64) + ((uint32_t)wpb + wslot) * 64u);   // long-row list of walk_rows
        if (bin_me) {   // the exiting lanes prepare their own lines, all at once, and park the result in LDS
          GridConst k;
          k.Rf = (float)d.R; k.rho = (float)d.rho_d; k.portz = (float)d.portz; k.n_theta = d.n_theta;
          k.inv_dphi = (float)d.n_phi * 0.15915494309f;
          k.inv_dth = (float)k.n_theta * 0.63661977237f;
          const RecPre pre = prep_record(k, r.p, r.v);
          rec4[lane] = make_float4(pre.Fz, pre.AF, pre.jf, pre.ch2);
          reci[lane] = pre.rows;
        }
      }
      while (em) {
        const int src = __builtin_ctzll(em);
        em &= em - 1ull;
        V3 P, V;
        P.x = readlane_f64(r.p.x, src); P.y = readlane_f64(r.p.y, src); P.z = readlane_f64(r.p.z, src);
        V.x = readlane_f64(r.v.x, src); V.y = readlane_f64(r.v.y, src); V.z = readlane_f64(r.v.z, src);
        if (SINK == SINK_DISC) {
          V3 P0;
          P0.x = readlane_f64(r.prev.x, src); P0.y = readlane_f64(r.prev.y, src); P0.z = readlane_f64(r.prev.z, src);
          bin_discs(d, hist, P0, P, V, lane);
        } else {
          if (!LEAN && hit_line_mode == 1) hit_line_compat(P, V);
          if (bin_mode == 0) bin_brute(d, hist, P, V, lane);
          else if (bin_mode == 1) {
            const int rows = fast ? __builtin_amdgcn_readfirstlane(reci[src]) : -1;   // same address in every lane
            if (rows == -2) { ISX_DIAG_ADD(3, 1); continue; }
            if (rows >= 0) {
              const float4 q4 = rec4[src];
              struct { int n_phi; double half_w2, portz; const double* table; } dfast;
              dfast.n_phi = d.n_phi; dfast.half_w2 = d.half_w2; dfast.portz = portz; dfast.table = d.table;
              CapWin wfast;
              wfast.inv_dphi = (float)dfast.n_phi * 0.15915494309f;
              wfast.Fz = q4.x; wfast.AF = q4.y; wfast.AF2 = q4.y * q4.y; wfast.jf = q4.z; wfast.ch2 = q4.w;
              ISX_DIAG_ADD(0, 1);
              walk_rows<MODE_CAP>(dfast, hist, rowt, colx, P, V, lane, rows & 0xffff, rows >> 16, wfast, BoxLine(), spl, 0);
            } else {
              bin_culled<false>(d, hist, rowt, colx, P, V, lane, spl);
            }
          }
          // bin_mode 2: diagnostic only (trace without binning; results are NOT a flux map)
        }
      }
    }
    ISX_TD_MARK(6);
  }
  ISX_TD_FLUSH();

  // ---- census + histogram flush
  if (SINK == SINK_REC && lane == 0 && reg_id != 0xffffffffu) d_arg.rec_counts[reg_id] = kRegion - reg_left;
  atomicAdd(&sstat[6], (unsigned long long)n_wall);
  if (lane == 0) {
    atomicAdd(&sstat[1], (unsigned long long)n_exited);
    atomicAdd(&sstat[2], (unsigned long long)n_counted);
    atomicAdd(&sstat[3], (unsigned long long)(n_ended - n_exited - n_susp));  // absorbed
    atomicAdd(&sstat[4], (unsigned long long)n_susp);
    atomicAdd(&sstat[0], (unsigned long long)n_taken);                         // launched
    atomicAdd(&sstat[5], n_inc);
  }
  __syncthreads();
  unsigned long long flushed = 0;   // increments of this block = sum of its LDS bins
  for (int b = tid; b < nbins; b += nthr) {
    const uint32_t c = hist[b];
    if (c) { global_add_u64(&wk.hist[b], (unsigned long long)c); flushed += c; }
  }
  if (SINK != SINK_LOG && flushed) atomicAdd(&sstat[5], flushed);
  __syncthreads();
  {
    // (the index is re-derived behind a compiler barrier: kept alive from the `sstat[tid] = 0` at the top of the kernel,
    //  this one address cost a VGPR across the whole trace loop -- 8 B/lane of scratch at the 128-VGPR limit)
    uint32_t t2 = threadIdx.x;
    asm volatile("" : "+v"(t2));
    if (t2 < 7u) {
      const unsigned long long c = sstat[t2];
      if (c) atomicAdd(&wk.stats[t2], c);
    }
  }
}

extern "C" __global__ void ISX_KERNEL_ATTR
isx_trace_bin_kernel(const Geom g, const DetGrid d, const Work wk) { persistent_body<SINK_FLUX, true, 0>(g, d, wk); }
// the same path with ISX_TRACE_CHORD compiled in (next wall point sampled directly)
extern "C" __global__ void ISX_KERNEL_ATTR
isx_trace_bin_chord_kernel(const Geom g, const DetGrid d, const Work wk) { persistent_body<SINK_FLUX, true, 1>(g, d, wk); }
// the lean path plus the BRDF re-scatter of nonLambertianFlux.C (BASELINE.json configs[2])
extern "C" __global__ void ISX_KERNEL_ATTR
isx_trace_bin_brdf_kernel(const Geom g, const DetGrid d, const Work wk) { persistent_body<SINK_FLUX, true, 0, true>(g, d, wk); }
// every surface / source / hit-line model (BRDF re-scatter, cos^2 lobe, rough specular, origin-compat line)
extern "C" __global__ void ISX_KERNEL_ATTR
isx_trace_bin_full_kernel(const Geom g, const DetGrid d, const Work wk) { persistent_body<SINK_FLUX, false, 2>(g, d, wk); }
extern "C" __global__ void ISX_KERNEL_ATTR
isx_trace_dz_kernel(const Geom g, const DetGrid d, const Work wk) { persistent_body<SINK_DZ>(g, d, wk); }
extern "C" __global__ void ISX_KERNEL_ATTR
isx_trace_dz_lean_kernel(const Geom g, const DetGrid d, const Work wk) { persistent_body<SINK_DZ, true, 0>(g, d, wk); }
extern "C" __global__ void ISX_KERNEL_ATTR
isx_trace_disc_kernel(const Geom g, const DetGrid d, const Work wk) { persistent_body<SINK_DISC>(g, d, wk); }
extern "C" __global__ void ISX_KERNEL_ATTR
isx_trace_disc_lean_kernel(const Geom g, const DetGrid d, const Work wk) { persistent_body<SINK_DISC, true, 0>(g, d, wk); }
extern "C" __global__ void ISX_KERNEL_ATTR
isx_trace_discpos_kernel(const Geom g, const DetGrid d, const Work wk) { persistent_body<SINK_DISCPOS>(g, d, wk); }
extern "C" __global__ void ISX_KERNEL_ATTR
isx_trace_discpos_lean_kernel(const Geom g, const DetGrid d, const Work wk) { persistent_body<SINK_DISCPOS, true, 0>(g, d, wk); }
extern "C" __global__ void ISX_KERNEL_ATTR
isx_trace_perpos_kernel(const Geom g, const DetGrid d, const Work wk) { persistent_body<SINK_PERPOS>(g, d, wk); }
// the reference's main sweep (per-position maps) in the headline configuration: lean trace, one exact test per exiting ray
extern "C" __global__ void ISX_KERNEL_ATTR
isx_trace_perpos_lean_kernel(const Geom g, const DetGrid d, const Work wk) { persistent_body<SINK_PERPOS, true, 0>(g, d, wk); }
// Two-kernel pipeline of the headline flux map: this kernel only traces and writes the exit lines (no histogram, no binning
// state: 0 B of LDS histogram), isx_bin_lines_kernel bins them.
extern "C" __global__ void ISX_KERNEL_ATTR
isx_trace_rec_kernel(const Geom g, const DetGrid d, const Work wk) { persistent_body<SINK_REC, true, 0>(g, d, wk); }
// the same for the optional chord mode and for the BRDF source model (configs[2]): only the trace differs, the lines are binned
// by the same isx_bin_lines_kernel
extern "C" __global__ void ISX_KERNEL_ATTR
isx_trace_rec_chord_kernel(const Geom g, const DetGrid d, const Work wk) { persistent_body<SINK_REC, true, 1>(g, d, wk); }
extern "C" __global__ void ISX_KERNEL_ATTR
isx_trace_rec_brdf_kernel(const Geom g, const DetGrid d, const Work wk) { persistent_body<SINK_REC, true, 0, true>(g, d, wk); }
extern "C" __global__ void ISX_KERNEL_ATTR
isx_trace_log_kernel(const Geom g, const DetGrid d, const Work wk) { persistent_body<SINK_LOG>(g, d, wk); }
extern "C" __global__ void ISX_KERNEL_ATTR
isx_trace_log_lean_kernel(const Geom g, const DetGrid d, const Work wk) { persistent_body<SINK_LOG, true, 0>(g, d, wk); }

// ------------------------------------------------------------------ trace kernel of the pipeline with an ASSIST wave
// (default for the lean flux maps since round 3).  In persistent_body a lane whose ray leaves rule S1' -- it heads for the port
// opening: 0.75 % of the lane-steps -- parks until a dozen lanes of its wave wait, and the generic boundary search (port,
// rim, outer sphere, world box: ~400 instructions with three IEEE square roots) then runs for ~9 of 64 lanes: 10 % of the
// trace kernel's time, and 5 of 64 lanes parked at any moment (tools/diag_trace.py).  Here the LAST wave of every workgroup
// traces nothing: it serves the others.  A tracer lane that leaves rule S1' hands its ray (64 bytes: point, direction, index,
// interaction count) to a workgroup-wide LDS queue at the end of the loop trip and is free for the next ray at once; the
// assist wave takes up to 64 queued rays at a time -- lane = ray -- and does for them everything that is not a bounce off the
// inner sphere: the generic search, the interaction with the rim or the outer sphere, the end of the ray at the world box
// with the port census and the exit line for the binning kernel (SINK_REC's regions), the BRDF re-scatter of a primary that
// left.  A ray that is back on the inner sphere returns through a second LDS queue, from which the tracers refill before
// they take fresh rays.  Scheduling only: a ray's history is a function of (seed, index), so histogram and census are those
// of persistent_body<SINK_REC> bit for bit.
//
// Queue discipline (all in LDS, workgroup scope):
//   pending  kPendCap slots, many producers (the tracers: slots reserved by compare-and-swap, so the ring never overflows; a
//            wave that finds no room keeps its rays and tries again a trip later), one consumer (the assist wave, which only
//            takes what is published: res == pub);
//   resume   kResumeCap slots, many consumers (the tracers: compare-and-swap on the head) and, since round 4, several
//            producers (slots reserved by compare-and-swap, taken only once everything reserved is published: pub == res, as
//            on the pending ring): the assist wave, which waits for room -- the tracers never wait for it, so there is no
//            cycle -- and the tracer waves that give their last rays away at the end of a launch (below);
//   busy     rays that are in neither a tracer lane nor ended; a tracer wave with no ray left leaves when the launch's ray
//            queue is dry and busy == 0, the assist wave when every tracer has left.
// The end of a launch (round 4).  Once the launch's ray queue is dry a wave only loses rays: after ~20 loop trips it runs a
// dozen lanes of 64 and keeps doing so for another ~40 trips, until its longest ray has ended -- 3 % of all trips of a 5e7-ray
// launch of the headline, 8 % at 2e7 rays, 12 % of the shared-ray disc sweep (reflectance 1: 144 bounces per ray), all of them
// at a fifth of the lanes.  So a wave that is down to kDonateMax rays after the queue ran dry hands them to the resume ring --
// the records the assist wave writes there, with the two Philox words an odd interaction count needs taken from the lane's own
// block -- and leaves; the waves that stay fill their dead lanes with them.  `alive` keeps the last wave from leaving that
// way, and a wave that leaves because it sees no work (busy == 0) takes itself out of `alive` FIRST and looks again: a donor
// announces its rays in `busy` before it takes itself out, so one of the two always sees the other.
// Scheduling only: a ray's history is a function of (seed, index).
// Every wait is bounded (kSpinLimit polls during which NO wave of the workgroup made progress -- AssistQueues::beat; a long
// stretch without hand-overs, e.g. a tiny port opening with a large bounce limit or a down-clocked device, is not a failure):
// a wave that gives up raises stats[7] and the host reports ISX_ERR_HIP instead of hanging the device.
constexpr uint32_t kPendCap = 512, kResumeCap = 128;
#ifndef ISX_DONATE_MAX
#define ISX_DONATE_MAX 32     // a tracer wave with at most this many rays left after the launch's queue ran dry gives them away (0: never;
                              // measured, trace kernel of 5e7 headline rays / of the 1e7-ray disc sweep: 0: 11.35 / 7.13 ms, 16: 11.27 / 6.79,
                              // 24: 11.29 / 6.76, 32: 11.23 / 6.71, 48: 11.31 / 6.85)
#endif
constexpr uint32_t kDonateMax = ISX_DONATE_MAX;
constexpr uint32_t kSpinLimit = 1u << 22;
struct AssistQueues {   // LDS, one per workgroup
  uint32_t pend_res, pend_pub, pend_head, resume_pub, resume_head, busy, tracers_done, failed;
  uint32_t drain, beat, resume_res, alive;   // drain: a tracer wave has no ray left and none to get (the assist wave stops waiting for full batches)
                                      // resume_res: slots of the resume ring reserved (it has several producers since round 4)
                                      // alive: tracer waves that have neither left nor given their last rays away (set to n_tracers)
                                      // beat: bumped by every wave of the workgroup that makes progress (a tracer's loop trip, a batch of the
                                      // assist wave): the bounded waits below count polls WITHOUT a beat, not idle time
};
enum : uint32_t { IDO_SCATTERED = 0x80000000u, IDO_TARGET = 0x40000000u };   // Ray::ido flags in a queue record (offsets < 2^30)

// A ray as a 64-byte record of the LDS queues: point, direction, {index | flags, interaction count, two more words} -- the
// surface it sits on and a spare for the pending ring, words 2-3 of its Philox block for the resume ring.
__device__ __forceinline__ void ray_pack(uint4* dst, const Ray& r, uint32_t wa, uint32_t wb) {
  const unsigned long long px = (unsigned long long)__double_as_longlong(r.p.x), py = (unsigned long long)__double_as_longlong(r.p.y),
                           pz = (unsigned long long)__double_as_longlong(r.p.z), vx = (unsigned long long)__double_as_longlong(r.v.x),
                           vy = (unsigned long long)__double_as_longlong(r.v.y), vz = (unsigned long long)__double_as_longlong(r.v.z);
  dst[0] = make_uint4((uint32_t)px, (uint32_t)(px >> 32), (uint32_t)py, (uint32_t)(py >> 32));
  dst[1] = make_uint4((uint32_t)pz, (uint32_t)(pz >> 32), (uint32_t)vx, (uint32_t)(vx >> 32));
  dst[2] = make_uint4((uint32_t)vy, (uint32_t)(vy >> 32), (uint32_t)vz, (uint32_t)(vz >> 32));
  dst[3] = make_uint4(r.ido | (r.tgt ? IDO_TARGET : 0u), r.j, wa, wb);
}
__device__ __forceinline__ void ray_unpack(const uint4& a, const uint4& b, const uint4& c, const uint4& e, Ray& r) {
  r.p.x = __longlong_as_double(((long long)a.y << 32) | a.x); r.p.y = __longlong_as_double(((long long)a.w << 32) | a.z);
  r.p.z = __longlong_as_double(((long long)b.y << 32) | b.x); r.v.x = __longlong_as_double(((long long)b.w << 32) | b.z);
  r.v.y = __longlong_as_double(((long long)c.y << 32) | c.x); r.v.z = __longlong_as_double(((long long)c.w << 32) | c.z);
  r.ido = e.x & ~IDO_TARGET; r.tgt = (e.x & IDO_TARGET) != 0u; r.j = e.y; r.k = 0u;
}

// DISC (the shared-ray physical-disc sweep, integratingSphereDetectorSweep.C:134-172 / SINK_DISC): the assist wave writes, for
// EVERY ray that leaves for the world box, its forward exit segment -- start point, direction, length (8 doubles per slot) --
// and isx_bin_discs_kernel tests the segments against the discs.
// PP (the per-position sinks, one launch for all positions): 1 = SINK_PERPOS (a ray is tested against the detector(s) of its own
// group only), 2 = SINK_DISCPOS (its forward exit segment against its own disc only) -- per-lane work at the exit, so the
// assist wave does it on the spot and adds the rare hit to the global bins; no exit lines, no second kernel.
// SURF (round 5): the border's model -- SURF_LAMBERT (the lean path), SURF_ROUGH (ROBAST's rough-specular border: the same steps
// with another interact()), SURF_LOBE (the cos^2-lobe rejection sampler of "nonLambertianFlux copy.C":31-70, the de-facto
// CustomMirror).  Rule S1' holds for ANY direction that leaves a point of the inner sphere inwards, so only the emission differs.
// The lobe's rejection loop is not run inside a step (its length is geometric with acceptance 0.707: the 64 lanes of a wave would
// wait for the longest of 64 loops, ~4.4 tries per step where 1.41 are needed); a lane takes ONE try per step instead -- a lane
// whose try was rejected stays at its new point (Ray::k counts its tries) and tries again in the next step while its neighbours
// go on to their next wall point.  The tries of an interaction are a function of (seed, ray, interaction, try index) alone, so
// the schedule changes nothing: bit-equal to lobe_sample()'s loop (oracle: isxo lobe_sample).
template <int CH, bool RESC, bool DISC = false, int PP = 0, int SURF = SURF_LAMBERT>
__device__ __forceinline__ void assist_body(const Geom& g_arg, const DetGrid& d_arg, const Work& wk) {
  constexpr bool LEAN = SURF == SURF_LAMBERT;
  static_assert(LEAN || (CH == 0 && !RESC), "the chord identity and the BRDF re-scatter pipeline are built for the Lambertian border");
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned long long* sstat = reinterpret_cast<unsigned long long*>(smem);
  Geom* g_lds = reinterpret_cast<Geom*>(sstat + 8);
  DetGrid* d_lds = reinterpret_cast<DetGrid*>(g_lds + 1);
  AssistQueues* Q = reinterpret_cast<AssistQueues*>(smem + ((reinterpret_cast<unsigned char*>(d_lds + 1) - smem + 15) & ~(size_t)15));
  uint4* resume_q = reinterpret_cast<uint4*>(Q + 1);               // [kResumeCap][4]
  uint4* pend_q = resume_q + 4 * kResumeCap;                       // [kPendCap][4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int nthr = (int)blockDim.x, wpb = nthr >> 6;
  const int n_tracers = wpb - 1;
  if (tid < 8) sstat[tid] = 0ull;
  if (tid == 64) {
    *g_lds = g_arg;
    const Hot h0 = make_hot(g_arg);
    V3 s0, d0, q0;
    s0.x = g_arg.src[0]; s0.y = g_arg.src[1]; s0.z = g_arg.src[2];
    d0.x = g_arg.dir0[0]; d0.y = g_arg.dir0[1]; d0.z = g_arg.dir0[2];
    q0 = s0;
    const bool ok = next_hit_s1<true>(h0, g_arg, s0, d0, K_NONE, q0);
    g_lds->q0[0] = q0.x; g_lds->q0[1] = q0.y; g_lds->q0[2] = q0.z;
    g_lds->q0_ok = ok ? 1 : 0;
  }
  if (tid == 0) { *d_lds = d_arg; AssistQueues z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, (uint32_t)n_tracers}; *Q = z; }
  __syncthreads();
  typedef __attribute__((address_space(3))) Geom LdsGeom;
  const volatile LdsGeom& g = *(const volatile LdsGeom*)g_lds;
  const Hot h = make_hot(g_arg);
  const double portz = d_arg.portz;
  const uint64_t seed = wk.seed, first = wk.first;
  auto ld = [](uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); };
  auto add = [](uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); };
  uint32_t n_wall = 0;                                   // per lane
  uint32_t n_exited = 0, n_counted = 0, n_susp = 0, n_ended = 0, n_taken = 0;   // per wave
  unsigned long long n_inc = 0;                           // per wave: bin increments of the per-position sinks

  if ((tid >> 6) < n_tracers) {
    // =============================================================== tracer waves
    constexpr uint32_t kDry = 0xffffffffu;
    uint32_t next = 0, end = 0, spins = 0, beat_seen = 0;
    bool drained = false;
    Ray r;
    ray_start(g, r, 0);
    bool run = false, hand = false;    // hand: the lane's ray waits to be handed to the assist wave
    // (-DISX_DIAG, tools/diag_trace.py: cycles [16] refill, [17] bounce steps, [18] census, [19] hand-over; [23] trips, [24] lanes
    //  running / [25] still waiting to be handed over at the top of a trip, [26] hand-overs refused for lack of room, [27] rays
    //  taken back from the assist wave, [28] rays ended here, [29] waits for the workgroup's last rays)
    ISX_TD_DECL;
    for (;;) {
      // ---- refill: rays that come back from the assist wave first, then fresh ones off the launch's queue
      unsigned long long dead = __ballot(!(run || hand));
      if (dead) {
        // optimistic pop: read the candidates, then move the head by compare-and-swap -- it succeeds only if nobody else took
        // them, and the assist wave overwrites a slot only after the head has passed it, so what was read is what was won
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(dead >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dead, 0u));
        for (;;) {
          // (pub before res: pub(t1) <= res(t1) <= res(t2), so pub == res proves every slot below res is written)
          const uint32_t pubq = ld(&Q->resume_pub), resq = ld(&Q->resume_res), hd = ld(&Q->resume_head);
          const uint32_t avail = pubq == resq ? resq - hd : 0u;
          if (avail == 0u) break;
          const uint32_t want = (uint32_t)__popcll(dead);
          const uint32_t take = want < avail ? want : avail;
          const bool mine = !(run || hand) && rank < take;
          uint4 a = make_uint4(0u, 0u, 0u, 0u), b = a, c = a, e = a;
          if (mine) {
            const uint4* src = resume_q + 4 * ((hd + rank) & (kResumeCap - 1));
            a = src[0]; b = src[1]; c = src[2]; e = src[3];
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
          uint32_t won = 0;
          if (lane == 0) {
            uint32_t expect = hd;
            won = __hip_atomic_compare_exchange_strong(&Q->resume_head, &expect, hd + take, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
          }
          won = (uint32_t)__builtin_amdgcn_readfirstlane((int)won);
          if (!won) continue;
          if (mine) {
            ray_unpack(a, b, c, e, r);
            r.on = K_INNER;
            r.cw[0] = 0u; r.cw[1] = 0u; r.cw[2] = e.z; r.cw[3] = e.w;
            run = true;
          }
          if (lane == 0) __hip_atomic_fetch_sub(&Q->busy, take, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          ISX_TD_ADD(11, take);
          dead = __ballot(!(run || hand));
          break;
        }
      }
      if (dead) {
        if (next == end) {   // this wave's sub-range is used up: the next one off the launch's queue (persistent_body)
          const uint32_t n32 = (uint32_t)wk.n;
          const uint32_t share = (n32 - end) / (2u * (uint32_t)n_tracers * gridDim.x);
          const uint32_t want = share >= wk.sub ? wk.sub : (share > 64u ? share : 64u);
          unsigned long long b64 = 0;
          if (lane == 0) b64 = atomicAdd(reinterpret_cast<unsigned long long*>(wk.ctr + Q_RAYS), (unsigned long long)want);
          const uint32_t bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(b64 >> 32));
          const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b64);
          if (bhi == 0u && b < n32) { next = b; end = n32 - b > want ? b + want : n32; }
          else next = kDry;
        }
        if (next < end) {
          const uint32_t left = end - next;
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(dead >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dead, 0u));
          if (!(run || hand) && rank < left) { ray_start(g, r, next + rank); run = true; }
          const uint32_t want = (uint32_t)__popcll(dead);
          const uint32_t take = want < left ? want : left;
          next += take; n_taken += take;
        }
        if (__ballot(run || hand) == 0ull) {
          // no ray in this wave and none to be had from the launch; rays of this workgroup may still come back
          if (ld(&Q->busy) == 0u) {
            // out of `alive` first, then look again: a wave that gives its rays away has announced them in `busy` before it
            // took itself out, so either it found this wave still counted, or this wave finds its rays
            uint32_t gone = 0;
            if (lane == 0) {
              __hip_atomic_fetch_sub(&Q->alive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
              if (ld(&Q->busy) == 0u) gone = 1u;
              else add(&Q->alive, 1u);
            }
            if (__builtin_amdgcn_readfirstlane((int)gone)) break;
          }
          if (!drained) { drained = true; if (lane == 0) __hip_atomic_store(&Q->drain, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
          if (ld(&Q->failed) != 0u) break;                                                             // a wave of this workgroup gave up: the launch has failed
          { const uint32_t bt = ld(&Q->beat); if (bt != beat_seen) { beat_seen = bt; spins = 0u; } }   // somebody is still at work
          if (++spins > kSpinLimit) { if (lane == 0) add(&Q->failed, 1u); break; }
          ISX_TD_ADD(13, 1);
          __builtin_amdgcn_s_sleep(8);
          continue;
        }
      }
      spins = 0u;
      if (lane == 0) __hip_atomic_fetch_add(&Q->beat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      ISX_TD_MARK(0);
      ISX_TD_ADD(7, 1); ISX_TD_ADD(8, __popcll(__ballot(run))); ISX_TD_ADD(9, __popcll(__ballot(hand)));
      // ---- kStepsPerTrip bounces off the inner sphere per live lane (rule S1'); anything else is the assist wave's
      // (the end of a ray as two flags -- lane masks in scalar registers -- instead of a status word per lane; a tracer lane's ray
      //  sits on the inner sphere unless it is fresh, so Ray::on is neither read nor written in the steps)
      bool ended = false, susp = false;
      auto arrive = [&](const V3& q, auto ph) {
        const int st = ray_arrive<false, LEAN, CH, decltype(ph)::value, false, SURF>(h, g, r, seed, first, K_INNER, q);
        if (st != 0) { run = false; ended = true; susp = st == ST_SUSPENDED; }
      };
      // SURF_LOBE: one TRY of the lobe's rejection sampler for every lane that is in an interaction (Ray::k != 0)
      auto lobe_try_pending = [&]() {
        if (run && r.k != 0u) {
          V3 n, sc;
          n.x = r.p.x * h.ninv_rin; n.y = r.p.y * h.ninv_rin; n.z = r.p.z * h.ninv_rin;   // surface_normal(K_INNER)
          const uint32_t idx = r.k - 1u;
          const bool acc = lobe_try(n, seed, first + (uint64_t)r.offset(), r.j - 1u, r.stream(), idx, sc);   // (j was counted at the arrival)
          if (acc || idx + 1u == kLobeTries) {
            lobe_hemisphere(n, sc);
            finish_direction(n, sc);
            r.v = sc; r.k = 0u;
          } else r.k++;
        }
      };
      // SURF_LOBE: one step = the next wall point for the lanes that have a direction (ray_arrive's order: move, absorb test on the
      // interaction's word b, count, bounce limit -- the direction of an absorbed or suspended ray is never formed: nothing reads it),
      // then one try for every lane that needs a direction
      auto lobe_step = [&](bool step0) {
        if (run && r.k == 0u) {
          V3 q;
          bool arrived = false;
          const bool fresh = step0 && r.j == 0u && !r.scattered();
          if (fresh) {
            if (g.q0_ok) { q.x = g.q0[0]; q.y = g.q0[1]; q.z = g.q0[2]; arrived = true; }
            else { hand = true; run = false; }
          } else if (next_hit_s1<false>(h, g, r.p, r.v, K_INNER, q)) arrived = true;
          else { hand = true; run = false; }
          if (arrived) {
            r.p = q;
            uint32_t w4[4];
            draw_block(seed, first + (uint64_t)r.offset(), r.j >> 1, r.stream(), w4);
            const uint32_t wb = (r.j & 1u) ? w4[3] : w4[1];
            r.j++;
            if (!((unsigned long long)wb < h.rho_thr)) { run = false; ended = true; }
            else if ((int)r.j + 1 > h.limit) { run = false; ended = true; susp = true; }
            else r.k = 1u;
          }
        }
        lobe_try_pending();
      };
      auto hot_search = [&](V3& q) -> bool {
        if (CH != 0 && r.tgt) return chord_arrive<true>(h, r, q);
        return next_hit_s1<false>(h, g, r.p, r.v, K_INNER, q);
      };
      if (next == kDry) { ISX_TD_ADD(14, 1); ISX_TD_ADD(15, __popcll(__ballot(run))); }   // (-DISX_DIAG: trips after the launch's queue ran dry)
      if constexpr (SURF == SURF_LOBE) {
        ISX_TD_ADD(4, __popcll(__ballot(run)));
        lobe_step(true);
        static_steps<1, kStepsPerTrip>([&](auto rep) {
          (void)rep;
          ISX_TD_ADD(4, __popcll(__ballot(run)));
          lobe_step(false);
        });
      } else {
      {
        V3 q;
        bool arrived = false;
        ISX_TD_ADD(4, __popcll(__ballot(run)));   // (-DISX_DIAG: lanes that attempt a bounce in this step)
        if (run) {
          const bool fresh = r.j == 0u && !r.scattered();   // (a ray that returns from the assist wave has interactions behind it)
          if (fresh) {
            if (g.q0_ok) { q.x = g.q0[0]; q.y = g.q0[1]; q.z = g.q0[2]; arrived = true; }
            else { hand = true; run = false; }
          } else if (hot_search(q)) arrived = true;
          else { hand = true; run = false; }
        }
        if (arrived) arrive(q, std::integral_constant<int, PH_EVEN>());
      }
      static_steps<1, kStepsPerTrip>([&](auto rep) {
        V3 q;
        bool arrived = false;
        ISX_TD_ADD(4, __popcll(__ballot(run)));
        if (run) {
          if (hot_search(q)) arrived = true;
          else { hand = true; run = false; }
        }
        if (arrived) arrive(q, std::integral_constant<int, (decltype(rep)::value & 1) ? PH_ODD : PH_EVEN>());
      });
      }
      ISX_TD_MARK(1);
      // ---- rays that ended on the inner sphere (absorbed, suspended): census, or the BRDF re-scatter of a primary
      if (RESC && ended && h.source_model == 1 && !r.scattered()) {
        n_wall += r.j;
        r.on = K_INNER;                  // (a primary ends on the inner sphere here: the re-scattered ray starts there and runs on)
        ray_rescatter(g, r, seed, first);
        run = true;
        ended = false; susp = false;
      }
      {
        if (ended) {
          n_wall += r.j;
          if (n_wall > 0x7fffffffu) { atomicAdd(&sstat[6], (unsigned long long)n_wall); n_wall = 0; }
        }
        const unsigned long long me = __ballot(ended);
        if (me) {
          n_ended += (uint32_t)__popcll(me);
          n_susp += (uint32_t)__popcll(__ballot(susp));
          ISX_TD_ADD(12, __popcll(me));
        }
      }
      ISX_TD_MARK(2);
      // ---- hand the rays that left rule S1' to the assist wave
      const unsigned long long hm = __ballot(hand);
      if (hm) {
        const uint32_t cnt = (uint32_t)__popcll(hm);
        uint32_t base = 0, ok = 0;
        if (lane == 0) {
          for (;;) {
            const uint32_t res = ld(&Q->pend_res), hd = ld(&Q->pend_head);
            if (res + cnt - hd > kPendCap) break;                      // no room: keep them for a trip
            uint32_t expect = res;
            if (__hip_atomic_compare_exchange_strong(&Q->pend_res, &expect, res + cnt, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WORKGROUP)) { base = res; ok = 1u; break; }
          }
        }
        ok = (uint32_t)__builtin_amdgcn_readfirstlane((int)ok);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (ok) {
          if (hand) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
            // (the surface the ray sits on: none for a fresh ray -- Geom::q0_ok = 0 --, else the inner sphere)
            ray_pack(pend_q + 4 * ((base + rank) & (kPendCap - 1)), r, (r.j == 0u && !r.scattered()) ? (uint32_t)K_NONE : (uint32_t)K_INNER, 0u);
            hand = false;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) { add(&Q->busy, cnt); add(&Q->pend_pub, cnt); }
        } else {
          ISX_TD_ADD(10, 1);
          if (ld(&Q->failed) != 0u) break;   // no room and the assist wave has given up: nobody will make any
        }
      }
      ISX_TD_MARK(3);
      // ---- the end of the launch: a wave that is down to a few rays gives them to the waves that stay, and leaves
      if (kDonateMax != 0u && next == kDry) {
        const unsigned long long live = __ballot(run);
        const uint32_t cnt = (uint32_t)__popcll(live);
        if (cnt != 0u && cnt <= kDonateMax && __ballot(hand) == 0ull) {
          uint32_t ok = 0, base = 0;
          if (lane == 0) {
            add(&Q->busy, cnt);                                          // announced before this wave takes itself out
            for (;;) {
              const uint32_t al = ld(&Q->alive);
              if (al < 2u) break;                                        // the last wave keeps its rays
              uint32_t expect = al;
              if (__hip_atomic_compare_exchange_strong(&Q->alive, &expect, al - 1u, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP)) { ok = 1u; break; }
            }
            if (ok) {
              for (;;) {
                const uint32_t rs = ld(&Q->resume_res), h2 = ld(&Q->resume_head);
                if (rs + cnt - h2 > kResumeCap) { ok = 0u; break; }      // no room: another trip with these rays
                uint32_t expect = rs;
                if (__hip_atomic_compare_exchange_strong(&Q->resume_res, &expect, rs + cnt, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_WORKGROUP)) { base = rs; break; }
              }
              if (!ok) add(&Q->alive, 1u);
            }
            if (!ok) __hip_atomic_fetch_sub(&Q->busy, cnt, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          ok = (uint32_t)__builtin_amdgcn_readfirstlane((int)ok);
          base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
          if (ok) {
            // (a queue record has no room for a try count: a lane that is in the middle of an interaction finishes its tries first)
            if constexpr (SURF == SURF_LOBE) { while (__ballot(run && r.k != 0u) != 0ull) lobe_try_pending(); }
            if (run) {
              const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0u));
              // (the record the assist wave writes for a ray that returns: at the top of a trip a lane with an odd interaction
              //  count holds words 2-3 of its block j/2 in cw[2..3] -- bounce_words)
              ray_pack(resume_q + 4 * ((base + rank) & (kResumeCap - 1)), r, r.cw[2], r.cw[3]);
              run = false;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) {
              add(&Q->resume_pub, cnt);
              __hip_atomic_store(&Q->drain, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (no more waiting for full batches)
            }
            ISX_TD_ADD(5, cnt); ISX_TD_ADD(6, 1);
            break;
          }
        }
      }
    }
    ISX_TD_FLUSH();
    if (lane == 0) add(&Q->tracers_done, 1u);
  } else {
    // =============================================================== the assist wave
    uint32_t reg_slot = 0, reg_left = 0, reg_id = 0xffffffffu;       // cursor in the open region of exit lines (SINK_REC)
    uint32_t spins = 0, lazy = 0, beat_seen = 0;
    // (-DISX_DIAG, g_diag[32..]: cycles [0] waiting for rays, [1] at work; [2] batches, [3] rays in them, [4] rays sent to the back
    //  of the pending queue, [5] rays returned to the tracers, [6] rays that ended here)
    ISX_TD_DECL;
    // one wave serves eleven: it goes first whenever it has something to do (its SIMD's five tracers take every other slot)
    __builtin_amdgcn_s_setprio(3);
    for (;;) {
      // pend_pub is a SUM of published counts, not an ordered cursor: it is read BEFORE pend_res, so that pub(t1) <= res(t1) <=
      // res(t2) and pub == res proves that every reservation below res was published at t1.  (Read the other way round, a batch
      // that reserves, writes and publishes between the two loads could make up for an earlier one that has reserved but not yet
      // published, and this wave would read the earlier batch's unwritten slots.)
      const uint32_t pub = ld(&Q->pend_pub), res = ld(&Q->pend_res), hd = ld(&Q->pend_head);
      const uint32_t n = pub == res ? res - hd : 0u;                 // everything below res is written once pub has caught up
      if (n == 0u) {
        if (res == hd && ld(&Q->tracers_done) == (uint32_t)n_tracers) break;
        if (ld(&Q->failed) != 0u && ld(&Q->tracers_done) == (uint32_t)n_tracers) break;             // (a failed launch: whatever is reserved stays unpublished)
        { const uint32_t bt = ld(&Q->beat); if (bt != beat_seen) { beat_seen = bt; spins = 0u; } }   // the tracers are at work
        if (++spins > kSpinLimit) { if (lane == 0) add(&Q->failed, 1u); break; }
        __builtin_amdgcn_s_sleep(4);
        continue;
      }
      // A few rays only: wait a little for more.  A batch costs the same ~500 instructions whether it holds 8 rays or 64, and
      // taken as they came the batches held ~8 (this wave has priority on its SIMD and was back at the queue before the eleven
      // tracers had handed over more).  Bounded (ISX_ASSIST_LAZY polls of 512 cycles: the last rays of a launch are held back
      // by 27 us at most per generation), and over as soon as a tracer wave of the workgroup stands idle -- no ray left, none
      // to get from the launch's queue: from then on every batch that waits keeps a wave waiting.  (Ending it earlier, when a
      // tracer finds the launch's ray queue empty, was measured: 11.48 against 11.35 ms -- the waves still hold sub-ranges
      // then.)  Scheduling only.
      if (n < (uint32_t)ISX_ASSIST_MIN && lazy < (uint32_t)ISX_ASSIST_LAZY && ld(&Q->drain) == 0u) {
        ++lazy;
        __builtin_amdgcn_s_sleep(8);
        continue;
      }
      lazy = 0;
      spins = 0;
      if (lane == 0) __hip_atomic_fetch_add(&Q->beat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t take = n < 64u ? n : 64u;
      ISX_TD_MARK(0);
      ISX_TD_ADD(2, 1); ISX_TD_ADD(3, take);
      const bool have = (uint32_t)lane < take;
      Ray r;
      ray_start(g, r, 0);
      if (have) {
        const uint4* src = pend_q + 4 * ((hd + (uint32_t)lane) & (kPendCap - 1));
        const uint4 a = src[0], b = src[1], c = src[2], e = src[3];
        ray_unpack(a, b, c, e, r);
        r.on = (int)e.z;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // (the slots are free once they have been read)
      if (lane == 0) add(&Q->pend_head, take);
      // ---- one step of everything that is not a bounce off the inner sphere.  A ray that is then neither ended nor back on
      // the inner sphere (it sits on the rim, on the outer sphere, or starts its re-scattered life on the world box) goes to
      // the back of the pending queue: its next step is taken with a full wave again, not with the two or three lanes that
      // need one.  (No room there: it takes its steps here.)
      int st = 0;
      bool go = have;
      for (;;) {
        if (go) {
          if (CH != 0 && r.tgt) chord_leave(r);
          else if (r.on == K_INNER) unit_dir(r.v);   // handed over by a tracer whose rule S1' failed: unit direction from here on
          V3 q;
          const int kind = next_hit_generic(g, r.p, r.v, r.on, q);
          st = ray_arrive<DISC || PP == 2, LEAN, CH, PH_DIRECT, true, SURF>(h, g, r, seed, first, kind, q);   // (DISC: r.prev = start of this segment)
          if (RESC && st != 0 && h.source_model == 1 && !r.scattered()) {   // nonLambertianFlux.C:253-268
            n_wall += r.j;
            ray_rescatter(g, r, seed, first);
            st = 0;
          }
          if (st != 0 || r.on == K_INNER) go = false;
        }
        const unsigned long long gm = __ballot(go);
        if (gm == 0ull) break;
        const uint32_t cnt = (uint32_t)__popcll(gm);
        uint32_t base = 0, ok = 0;
        if (lane == 0) {
          for (;;) {
            const uint32_t rs = ld(&Q->pend_res), h2 = ld(&Q->pend_head);
            if (rs + cnt - h2 > kPendCap) break;
            uint32_t expect = rs;
            if (__hip_atomic_compare_exchange_strong(&Q->pend_res, &expect, rs + cnt, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WORKGROUP)) { base = rs; ok = 1u; break; }
          }
        }
        ok = (uint32_t)__builtin_amdgcn_readfirstlane((int)ok);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (!ok) continue;
        if (go) {
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(gm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)gm, 0u));
          ray_pack(pend_q + 4 * ((base + rank) & (kPendCap - 1)), r, (uint32_t)r.on, 0u);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) add(&Q->pend_pub, cnt);
        break;
      }
      const bool requeued = go;   // (still counted in `busy`)
      // ---- census + exit lines of the rays that ended
      const bool ended = have && st != 0, exited = have && st == ST_EXITED;
      const bool below = exited && (r.p.z < portz);                   // isRayPassingThroughExitPort, fluxAtObserver.C:162-166
      if (ended) n_wall += r.j;
      if (n_wall > 0x7fffffffu) { atomicAdd(&sstat[6], (unsigned long long)n_wall); n_wall = 0; }
      const uint32_t c_ended = (uint32_t)__popcll(__ballot(ended));
      n_ended += c_ended;
      n_exited += (uint32_t)__popcll(__ballot(exited));
      n_susp += (uint32_t)__popcll(__ballot(have && st == ST_SUSPENDED));
      n_counted += (uint32_t)__popcll(__ballot(below));
      const bool keep = (DISC || PP == 2) ? exited : below;           // what goes to the binning kernel (or is binned here: PP)
      if (PP != 0) {
        // persistent_body's SINK_PERPOS / SINK_DISCPOS, per lane: the ray's own detector group (disc) only
        bool hit0 = false, hit1 = false;
        int b0 = 0, b1 = 0;
        if (keep) {
          const uint64_t rel = (first + (uint64_t)r.offset()) - d_arg.map_first, rpg = d_arg.rays_per_group;
          uint64_t grp = (uint64_t)((double)rel / (double)rpg);
          if (grp * rpg > rel) grp--;
          else if ((grp + 1) * rpg <= rel) grp++;
          if (PP == 1) {
            const double* table = d_arg.table;
            const V3 lp = r.p, lv = r.v;
            if (d_arg.fold == 2) {
              const int nphi = d_arg.n_phi, half = nphi / 2;
              const int i = (int)(grp / (uint64_t)half), j = (int)(grp % (uint64_t)half);
              b0 = i * nphi + j;
              b1 = b0 + half;
              hit1 = check_intersection(table + 6 * (size_t)b1, d_arg.half_w2, lp, lv);
            } else {
              b0 = (int)grp;
            }
            hit0 = check_intersection(table + 6 * (size_t)b0, d_arg.half_w2, lp, lv);
          } else {
            b0 = (int)grp;
            V3 dl; dl.x = r.p.x - r.prev.x; dl.y = r.p.y - r.prev.y; dl.z = r.p.z - r.prev.z;
            hit0 = segment_hits_tube_g(r.prev, r.v, dot3(dl, r.v), d_arg.discs + 6 * (size_t)b0, d_arg.disc_r, d_arg.disc_h);
          }
        }
        if (hit0) atomicAdd(&wk.hist[b0], 1ull);
        if (hit1) atomicAdd(&wk.hist[b1], 1ull);
        n_inc += (unsigned long long)__popcll(__ballot(hit0)) + (unsigned long long)__popcll(__ballot(hit1));
      }
      const unsigned long long m = PP != 0 ? 0ull : __ballot(keep);
      if (m) {
        const uint32_t cnt = (uint32_t)__popcll(m);
        if (cnt > reg_left) {   // close the open region, reserve the next one (kRegion)
          if (lane == 0 && reg_id != 0xffffffffu) d_arg.rec_counts[reg_id] = kRegion - reg_left;
          uint32_t id = 0;
          if (lane == 0) id = atomicAdd(&wk.ctr[Q_REGIONS], 1u);
          reg_id = (uint32_t)__builtin_amdgcn_readfirstlane((int)id);
          reg_slot = 0; reg_left = kRegion;
        }
        if (keep) {
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          const uint64_t slot = (uint64_t)reg_id * kRegion + (uint64_t)(reg_slot + rank);
          if (DISC) {   // forward exit segment: start, direction, length (bin_discs' own expression)
            V3 dl; dl.x = r.p.x - r.prev.x; dl.y = r.p.y - r.prev.y; dl.z = r.p.z - r.prev.z;
            double2* dst = reinterpret_cast<double2*>(d_arg.rec_lines + 8ull * slot);
            dst[0] = make_double2(r.prev.x, r.prev.y); dst[1] = make_double2(r.prev.z, r.v.x); dst[2] = make_double2(r.v.y, r.v.z);
            dst[3] = make_double2(dot3(dl, r.v), 0.0);
          } else {
            // (ISX_HITLINE_ORIGIN_COMPAT: isx_compat_lines_kernel rewrites the lines between this kernel and the binning kernel)
            const V3 lp = r.p, lv = r.v;
            double2* dst = reinterpret_cast<double2*>(d_arg.rec_lines + 6ull * slot);
            dst[0] = make_double2(lp.x, lp.y); dst[1] = make_double2(lp.z, lv.x); dst[2] = make_double2(lv.y, lv.z);
          }
        }
        reg_slot += cnt; reg_left -= cnt;
      }
      // ---- the rays that go on: back to the tracers
      const bool back = have && st == 0 && !requeued;
      const unsigned long long bm = __ballot(back);
      if (bm) {
        const uint32_t cnt = (uint32_t)__popcll(bm);
        // room?  (the tracers never wait for this wave; slots are reserved by compare-and-swap: waves that give their last rays
        // away write to this ring as well)
        uint32_t pubr = 0, w = 0, gave_up = 0;
        if (lane == 0) {
          for (;;) {
#ifdef ISX_TEST_GIVEUP   // (test build only, tests/test_gpu_round5.py: the first batch of workgroup 0 finds "no room, no progress")
            if (blockIdx.x == 0u) { add(&Q->failed, 1u); gave_up = 1u; break; }
#endif
            const uint32_t rs = ld(&Q->resume_res);
            if (rs + cnt - ld(&Q->resume_head) <= kResumeCap) {
              uint32_t expect = rs;
              if (__hip_atomic_compare_exchange_strong(&Q->resume_res, &expect, rs + cnt, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP)) { pubr = rs; break; }
              continue;
            }
            { const uint32_t bt = ld(&Q->beat); if (bt != beat_seen) { beat_seen = bt; w = 0u; } }
            // gives up: nothing is reserved, nothing written, nothing published (the ring stays consistent); the launch is
            // reported as failed and every wave of the workgroup leaves as soon as it sees the flag
            if (++w > kSpinLimit) { add(&Q->failed, 1u); gave_up = 1u; break; }
            __builtin_amdgcn_s_sleep(2);
          }
        }
        gave_up = (uint32_t)__builtin_amdgcn_readfirstlane((int)gave_up);
        pubr = (uint32_t)__builtin_amdgcn_readfirstlane((int)pubr);
        if (gave_up) break;
        if (back) {
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
          uint32_t cw2 = 0u, cw3 = 0u;
          if (r.j & 1u) {   // the tracer's next step finds words (2,3) of block j/2 in the lane (bounce_words, PH_EVEN)
            uint32_t wv[4];
            draw_block(seed, first + (uint64_t)r.offset(), r.j >> 1, r.stream(), wv);
            cw2 = wv[2]; cw3 = wv[3];
          }
          ray_pack(resume_q + 4 * ((pubr + rank) & (kResumeCap - 1)), r, cw2, cw3);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) add(&Q->resume_pub, cnt);
      }
      if (c_ended && lane == 0) __hip_atomic_fetch_sub(&Q->busy, c_ended, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      ISX_TD_ADD(4, __popcll(__ballot(requeued))); ISX_TD_ADD(5, __popcll(bm)); ISX_TD_ADD(6, c_ended);
      ISX_TD_MARK(1);
    }
    ISX_TD_FLUSH_AT(32);
    if (lane == 0 && reg_id != 0xffffffffu) d_arg.rec_counts[reg_id] = kRegion - reg_left;
  }

  // ---- census (persistent_body's epilogue)
  atomicAdd(&sstat[6], (unsigned long long)n_wall);
  if (lane == 0) {
    if (PP != 0 && n_inc) atomicAdd(&sstat[5], n_inc);
    atomicAdd(&sstat[1], (unsigned long long)n_exited);
    atomicAdd(&sstat[2], (unsigned long long)n_counted);
    atomicAdd(&sstat[3], (unsigned long long)(n_ended - n_exited - n_susp));  // absorbed
    atomicAdd(&sstat[4], (unsigned long long)n_susp);
    atomicAdd(&sstat[0], (unsigned long long)n_taken);
  }
  __syncthreads();
  if (tid == 0 && Q->failed) atomicAdd(&wk.stats[7], (unsigned long long)Q->failed);
  {
    uint32_t t2 = threadIdx.x;
    asm volatile("" : "+v"(t2));
    if (t2 < 7u) {
      const unsigned long long c = sstat[t2];
      if (c) atomicAdd(&wk.stats[t2], c);
    }
  }
}

#ifndef ISX_ASSIST_BLOCK
#define ISX_ASSIST_BLOCK 768
#endif
#define ISX_ASSIST_ATTR __launch_bounds__(ISX_ASSIST_BLOCK) __attribute__((amdgpu_waves_per_eu(6, 6)))
extern "C" __global__ void ISX_ASSIST_ATTR
isx_trace_assist_kernel(const Geom g, const DetGrid d, const Work wk) { assist_body<0, false>(g, d, wk); }
extern "C" __global__ void ISX_ASSIST_ATTR
isx_trace_assist_chord_kernel(const Geom g, const DetGrid d, const Work wk) { assist_body<1, false>(g, d, wk); }
extern "C" __global__ void ISX_ASSIST_ATTR
isx_trace_assist_brdf_kernel(const Geom g, const DetGrid d, const Work wk) { assist_body<0, true>(g, d, wk); }
extern "C" __global__ void ISX_ASSIST_ATTR
isx_trace_assist_disc_kernel(const Geom g, const DetGrid d, const Work wk) { assist_body<0, false, true>(g, d, wk); }
extern "C" __global__ void ISX_ASSIST_ATTR
isx_trace_assist_perpos_kernel(const Geom g, const DetGrid d, const Work wk) { assist_body<0, false, false, 1>(g, d, wk); }
extern "C" __global__ void ISX_ASSIST_ATTR
isx_trace_assist_discpos_kernel(const Geom g, const DetGrid d, const Work wk) { assist_body<0, false, false, 2>(g, d, wk); }
// the other border models on the same pipeline (round 5; until then round 1's fused isx_trace_bin_full_kernel served them):
// the cos^2 lobe of "nonLambertianFlux copy.C":31-70,188-221 and ROBAST's rough-specular border (EnableLambertian(false))
#ifndef ISX_LOBE_WAVES
#define ISX_LOBE_WAVES 4
#endif
#ifndef ISX_ROUGH_WAVES
#define ISX_ROUGH_WAVES 4
#endif
extern "C" __global__ void __launch_bounds__(ISX_ASSIST_BLOCK) __attribute__((amdgpu_waves_per_eu(ISX_LOBE_WAVES, ISX_LOBE_WAVES)))
isx_trace_assist_lobe_kernel(const Geom g, const DetGrid d, const Work wk) { assist_body<0, false, false, 0, SURF_LOBE>(g, d, wk); }
extern "C" __global__ void __launch_bounds__(ISX_ASSIST_BLOCK) __attribute__((amdgpu_waves_per_eu(ISX_ROUGH_WAVES, ISX_ROUGH_WAVES)))
isx_trace_assist_rough_kernel(const Geom g, const DetGrid d, const Work wk) { assist_body<0, false, false, 0, SURF_ROUGH>(g, d, wk); }
// ... and as per-position sinks: the macros that own these borders sweep one detector position at a time with fresh rays
// ("nonLambertianFlux copy.C":306-345: 45 x 20 positions x 1e5 rays), i.e. isx_fluxmap_per_position / isx_trace_rays_detector
extern "C" __global__ void __launch_bounds__(ISX_ASSIST_BLOCK) __attribute__((amdgpu_waves_per_eu(ISX_LOBE_WAVES, ISX_LOBE_WAVES)))
isx_trace_assist_perpos_lobe_kernel(const Geom g, const DetGrid d, const Work wk) { assist_body<0, false, false, 1, SURF_LOBE>(g, d, wk); }
extern "C" __global__ void __launch_bounds__(ISX_ASSIST_BLOCK) __attribute__((amdgpu_waves_per_eu(ISX_ROUGH_WAVES, ISX_ROUGH_WAVES)))
isx_trace_assist_perpos_rough_kernel(const Geom g, const DetGrid d, const Work wk) { assist_body<0, false, false, 1, SURF_ROUGH>(g, d, wk); }

// ISX_HITLINE_ORIGIN_COMPAT on the pipeline (round 5): what fluxAtObserverFast.C:1181-1201,1285-1288 effectively tested is the line
// from the origin along lastPoint/|lastPoint| (hit_line_compat).  The trace kernels write last point + final direction as always;
// this kernel rewrites the exit lines of a chunk in place -- one thread per line, region by region -- before the binning kernel
// reads them, so neither the trace kernels nor the binning kernels carry the switch (2 x 48 B per line: ~0.3 ms per 5e7 rays).
extern "C" __global__ void __launch_bounds__(256)
isx_compat_lines_kernel(double* __restrict__ rec_lines, const uint32_t* __restrict__ rec_counts, const uint32_t* __restrict__ ctr) {
  const uint32_t n_regions = ctr[Q_REGIONS];
  for (uint32_t reg = blockIdx.x; reg < n_regions; reg += gridDim.x) {
    const uint32_t cnt = rec_counts[reg];
    for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
      double* p6 = rec_lines + 6ull * ((uint64_t)reg * kRegion + i);
      V3 P, V;
      load_line(p6, P, V);
      hit_line_compat(P, V);
      double2* dst = reinterpret_cast<double2*>(p6);
      dst[0] = make_double2(P.x, P.y); dst[1] = make_double2(P.z, V.x); dst[2] = make_double2(V.y, V.z);
    }
  }
}

// ------------------------------------------------------------------ disc-binning kernel of the shared-ray disc sweep
// Persistent waves take quarter regions of exit segments (isx_trace_assist_disc_kernel) off the launch's queue, 64 segments at a
// time.  SINK_DISC tested every exit against every disc ("one exit at a time, lane = disc": VALU issue 0.70, most lanes culled);
// here, per batch:
//   1. lane = segment: the line of the segment against the ball of every CLUSTER of eight neighbouring discs (binary32, with a
//      margin that covers its rounding: never drops a hit); a (segment, cluster) pair that passes goes to a wave-private list;
//   2. lane = (pair, disc of the cluster): the exact test of SINK_DISC (segment_hits_tube: bounding-ball cull in binary64, then
//      the tube) on the segment read back from wave-private LDS; a hit increments the disc's bin.
// ~100 pairs per batch instead of 64 x 362 tests.  Same decisions, so the same counts.
// (LDS sets the occupancy here: with 1024 pairs per wave and 512-thread workgroups ONE workgroup fitted a CU -- 2 waves per SIMD,
//  1.45 ms for configs[3]; 256 pairs and 256-thread workgroups: four workgroups, 4 waves per SIMD, 0.96 ms.  -D to re-tune.)
#ifndef ISX_PAIR_CAP
#define ISX_PAIR_CAP 256
#endif
#ifndef ISX_DISC_BIN_BLOCK
#define ISX_DISC_BIN_BLOCK 256
#endif
constexpr int kPairCap = ISX_PAIR_CAP, kDiscsInLds = 1024, kDiscBinBlock = ISX_DISC_BIN_BLOCK;
static_assert(kPairCap >= 128 && kPairCap % 2 == 0, "a batch adds up to 64 pairs per cluster before the list is flushed");
extern "C" __global__ void __launch_bounds__(kDiscBinBlock)
isx_bin_discs_kernel(const DetGrid d_arg, const Work wk) {
  extern __shared__ __align__(16) unsigned char smem[];
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
  const int nbins = d_arg.nbins;
  const int tid = threadIdx.x, lane = tid & 63;
  const int nthr = (int)blockDim.x;
  // the cluster table (read once per cluster and batch by every wave: from LDS, not through a global-memory round trip per
  // iteration), then per wave the batch's 64 segments (7 doubles each) and the pair list
  // up to kDiscsInLds discs also keep their ordered list and permutation in LDS (the exact tests of phase 2 then wait for no
  // global-memory round trip: measured 70 % of the wave cycles before); longer lists are read from global memory
  const int n_clusters = d_arg.n_clusters;
  const bool staged = nbins <= kDiscsInLds;
  // (LDS-typed pointers throughout: through generic ones the wave-private lists became flat_load / flat_store with system
  //  scope -- volatile -- and the kernel waited on them for 70 % of its cycles)
  typedef __attribute__((address_space(3))) float LdsF32;
  typedef __attribute__((address_space(3))) double LdsF64;
  typedef __attribute__((address_space(3))) int LdsI32;
  typedef __attribute__((address_space(3))) uint32_t LdsU32;
  LdsF32* clusters = (LdsF32*)(smem + (((size_t)nbins * 4 + 15) & ~(size_t)15));   // [n_clusters][4]
  LdsF64* discs_lds = (LdsF64*)(clusters + 4 * n_clusters);
  LdsI32* perm_lds = (LdsI32*)(discs_lds + (staged ? 6 * nbins : 0));
  LdsF64* seg = (LdsF64*)(perm_lds + (staged ? ((nbins + 1) & ~1) : 0)) + (size_t)(tid >> 6) * (64 * 7 + kPairCap / 2);
  LdsU32* pairs = (LdsU32*)(seg + 64 * 7);
  for (int b = tid; b < nbins; b += nthr) hist[b] = 0u;
  for (int b = tid; b < 4 * n_clusters; b += nthr) clusters[b] = d_arg.clusters[b];
  if (staged) {
    for (int b = tid; b < 6 * nbins; b += nthr) discs_lds[b] = d_arg.discs[b];
    for (int b = tid; b < nbins; b += nthr) perm_lds[b] = d_arg.disc_perm[b];
  }
  __syncthreads();
  const double* __restrict__ discs = d_arg.discs;
  const int* __restrict__ perm = d_arg.disc_perm;
  const double disc_r = d_arg.disc_r, disc_h = d_arg.disc_h;
  const uint32_t n_regions = wk.ctr[Q_REGIONS];
  const uint32_t ushift = 2u + wk.pad;            // a work unit = 1024 >> ushift exit lines of one region: a quarter region; a sixteenth for small launches (Work::pad)
  // phase 2: the exact test for every (pair, disc of its cluster), 64 at a time
  auto flush = [&](int n_pairs) {
#pragma unroll 1
    for (int base = 0; base < n_pairs * 8; base += 64) {
      const int item = base + lane;
      const uint32_t pr = item < n_pairs * 8 ? pairs[item >> 3] : 0u;
      const int j = (int)(pr >> 8) * 8 + (item & 7);                 // disc in spatial order
      bool hit = false;
      int bin = 0;
      if (item < n_pairs * 8 && j < nbins) {
        const LdsF64* sp = seg + 7 * (int)(pr & 63u);
        V3 P0, V;
        P0.x = sp[0]; P0.y = sp[1]; P0.z = sp[2]; V.x = sp[3]; V.y = sp[4]; V.z = sp[5];
        double ca[6];   // (two branches so that the staged list is read with ds_read, not through a generic pointer)
        if (staged) { for (int q = 0; q < 6; ++q) ca[q] = discs_lds[6 * j + q]; bin = perm_lds[j]; }
        else { for (int q = 0; q < 6; ++q) ca[q] = discs[6 * (size_t)j + q]; bin = perm[j]; }
        hit = segment_hits_tube(P0, V, sp[6], ca, disc_r, disc_h);
      }
      if (hit) atomicAdd(&hist[bin], 1u);
    }
  };
#pragma unroll 1
  for (;;) {
    uint32_t unit = 0;
    if (lane == 0) unit = atomicAdd(&wk.ctr[Q_BIN], 1u);
    unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit);
    const uint32_t region = unit >> ushift;
    if (region >= n_regions) break;
    const uint32_t r_lines = (uint32_t)__builtin_amdgcn_readfirstlane((int)d_arg.rec_counts[region]);
    const uint32_t q_first = (unit & ((1u << ushift) - 1u)) * (kRegion >> ushift);
    const uint32_t n_lines = r_lines < q_first + (kRegion >> ushift) ? r_lines : q_first + (kRegion >> ushift);
    const double* rec = d_arg.rec_lines + 8ull * ((uint64_t)region * kRegion);
#pragma unroll 1
    for (uint32_t b0 = q_first; b0 < n_lines; b0 += 64u) {
      const bool have = b0 + (uint32_t)lane < n_lines;
      float px = 0.f, py = 0.f, pz = 0.f, vx = 0.f, vy = 0.f, vz = -1.f;
      if (have) {
        const GlbD2* src = (const GlbD2*)(rec + 8ull * (b0 + (uint32_t)lane));
        const isx_d2 a = src[0], b = src[1], c = src[2], e = src[3];
        LdsF64* sp = seg + 7 * lane;
        sp[0] = a.x; sp[1] = a.y; sp[2] = b.x; sp[3] = b.y; sp[4] = c.x; sp[5] = c.y; sp[6] = e.x;
        px = (float)a.x; py = (float)a.y; pz = (float)b.x; vx = (float)b.y; vy = (float)c.x; vz = (float)c.y;
      }
      __builtin_amdgcn_wave_barrier();
      const float vv = fmaf(vx, vx, fmaf(vy, vy, vz * vz));
      int n_pairs = 0;                                               // wave-uniform
#pragma unroll 1
      for (int k = 0; k < n_clusters; ++k) {
        float4 c;                                                    // (same LDS address in every lane: a broadcast read)
        c.x = clusters[4 * k]; c.y = clusters[4 * k + 1]; c.z = clusters[4 * k + 2]; c.w = clusters[4 * k + 3];
        const float wx = px - c.x, wy = py - c.y, wz = pz - c.z;
        const float wv = fmaf(wx, vx, fmaf(wy, vy, wz * vz)), ww = fmaf(wx, wx, fmaf(wy, wy, wz * wz));
        // line-to-centre distance^2 * vv = ww vv - wv^2 against R^2 vv, with 0.2 % on the radius^2 and 4e-6 of the two
        // cancelling terms for the binary32 rounding of inputs and products (2e-6 of them by the count of roundings)
        const float lhs = fmaf(ww, vv, -(wv * wv)), rhs = fmaf(c.w * c.w * 1.002f, vv, 4e-6f * fmaf(ww, vv, wv * wv));
        const bool near = have && !(lhs > rhs);
        const unsigned long long nm = __ballot(near);
        if (nm) {
          if (near) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(nm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nm, 0u));
            pairs[n_pairs + (int)rank] = (uint32_t)lane | ((uint32_t)k << 8);
          }
          n_pairs += (int)__popcll(nm);
          if (n_pairs > kPairCap - 64) { __builtin_amdgcn_wave_barrier(); flush(n_pairs); n_pairs = 0; }
        }
      }
      __builtin_amdgcn_wave_barrier();
      flush(n_pairs);
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  unsigned long long flushed = 0;
  for (int b = tid; b < nbins; b += nthr) {
    const uint32_t c = hist[b];
    if (c) { global_add_u64(&wk.hist[b], (unsigned long long)c); flushed += c; }
  }
  if (flushed) atomicAdd(&wk.stats[5], flushed);
}

// ------------------------------------------------------------------ binning kernel of the two-kernel pipeline
// Persistent waves take the regions of exit lines the trace kernel filled (kRegion slots each, rec_counts[region] lines in
// them) off the launch's queue, ctr[Q_BIN], and bin them 64 lines at a time: lane = line for the per-line preparation
// (prep_record), then the fast-path lines' rows packed over the lanes (walk_lines_packed) and every other line broadcast
// and binned by the whole wave exactly as in the fused kernel (walk_rows / bin_culled: same cull, same exact decision, so
// the histogram is the same).  No ray state lives here; the grid is what is resident (isx_api.hip), one histogram flush each.
#ifndef ISX_BIN_WAVES_PER_EU
#define ISX_BIN_WAVES_PER_EU 4
#endif
extern "C" __global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(ISX_BIN_WAVES_PER_EU, ISX_BIN_WAVES_PER_EU)))
isx_bin_lines_kernel(const DetGrid d_arg, const Work wk) {
  extern __shared__ __align__(16) unsigned char smem[];
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
  const int nbins = d_arg.nbins;
  const size_t off_row = ((size_t)nbins * 4 + 15) & ~(size_t)15;
  double* rowt = reinterpret_cast<double*>(smem + off_row);
  ColX* colx = reinterpret_cast<ColX*>(rowt + 4 * d_arg.n_theta);
  DetGrid* d_lds = reinterpret_cast<DetGrid*>(colx + 2 * d_arg.n_phi);
  int* split_all = reinterpret_cast<int*>(d_lds + 1);
  const int tid = threadIdx.x, lane = tid & 63;
  const int nthr = (int)blockDim.x;
  for (int b = tid; b < nbins; b += nthr) hist[b] = 0u;
  for (int b = tid; b < 4 * d_arg.n_theta; b += nthr) rowt[b] = d_arg.rowtab[b];
  for (int b = tid; b < 2 * d_arg.n_phi; b += nthr) {
    const int j = b < d_arg.n_phi ? b : b - d_arg.n_phi;
    ColX e;
    e.c = d_arg.coltab[2 * j]; e.s = d_arg.coltab[2 * j + 1]; e.off4 = (uint32_t)j * 4u; e.c32 = (float)e.c; e.s32 = (float)e.s; e.pad = 0u;
    colx[b] = e;
  }
  if (tid == (nthr > 128 ? 128 : 0)) *d_lds = d_arg;
  __syncthreads();
  typedef __attribute__((address_space(3))) DetGrid LdsDetGrid;
  const volatile LdsDetGrid& d = *(const volatile LdsDetGrid*)d_lds;
  LdsInt* spl = (LdsInt*)(split_all + (tid >> 6) * 128);   // per wave: 64 ints of long-row list + 64 ints of owner marks
  LdsInt* mrk = spl + 64;

  const int bin_mode = d_arg.bin_mode;
  const uint32_t n_regions = wk.ctr[Q_REGIONS];   // (the trace kernel of this launch has completed)
  const uint32_t ushift = 2u + wk.pad;            // a work unit = 1024 >> ushift exit lines of one region: a quarter region; a sixteenth for small launches (Work::pad)
#pragma unroll 1
  for (;;) {
    // (wave-uniform: through readfirstlane so that the region pointer lives in scalar registers)
    // (a quarter of a region at a time: the waves finish closer together than with whole regions)
    uint32_t unit = 0;
    if (lane == 0) unit = atomicAdd(&wk.ctr[Q_BIN], 1u);
    unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit);
    const uint32_t region = unit >> ushift;
    if (region >= n_regions) break;
    const uint32_t r_lines = (uint32_t)__builtin_amdgcn_readfirstlane((int)d_arg.rec_counts[region]);
    const uint32_t q_first = (unit & ((1u << ushift) - 1u)) * (kRegion >> ushift);
    const uint32_t n_lines = r_lines < q_first + (kRegion >> ushift) ? r_lines : q_first + (kRegion >> ushift);
    const double* rec = d_arg.rec_lines + 6ull * ((uint64_t)region * kRegion);
#pragma unroll 1
    for (uint32_t b0 = q_first; b0 < n_lines; b0 += 64u) {
      const bool have = b0 + (uint32_t)lane < n_lines;
      // (wave-uniform constants of the per-line preparation, derived again for every batch from the LDS copy of the grid: kept
      //  across the walks below they would sit in VGPRs -- gfx950 has no scalar float unit -- and push three values to scratch)
      GridConst k;
      k.Rf = (float)d.R; k.rho = (float)d.rho_d; k.portz = (float)d.portz; k.n_theta = d.n_theta;
      k.inv_dphi = (float)d.n_phi * 0.15915494309f;
      k.inv_dth = (float)k.n_theta * 0.63661977237f;
      V3 lp, lv;
      lp.x = lp.y = lp.z = 0.0; lv.x = lv.y = 0.0; lv.z = -1.0;
      RecPre pre;
      pre.Fz = pre.AF = pre.jf = pre.ch2 = 0.f; pre.rows = -1;
      if (have) {
        load_line(rec + 6ull * (b0 + (uint32_t)lane), lp, lv);
        if (bin_mode == 1) pre = prep_record(k, lp, lv);
      }
      struct { int n_phi; double half_w2, portz; const double* table; } dfast;
      dfast.n_phi = d.n_phi; dfast.half_w2 = d.half_w2; dfast.portz = d_arg.portz; dfast.table = d.table;
      if (bin_mode == 0) {   // brute-force reference-order test of every bin (tests)
        unsigned long long em = __ballot(have);
        while (em) {
          const int src = __builtin_ctzll(em);
          em &= em - 1ull;
          V3 P, V;
          P.x = readlane_f64(lp.x, src); P.y = readlane_f64(lp.y, src); P.z = readlane_f64(lp.z, src);
          V.x = readlane_f64(lv.x, src); V.y = readlane_f64(lv.y, src); V.z = readlane_f64(lv.z, src);
          bin_brute(d, hist, P, V, lane);
        }
        continue;
      }
      // (lines that cannot hit anything -- pre.rows == -2 -- end here, 64 at a time)
      { const int n_far = (int)__popcll(__ballot(have && pre.rows == -2)); (void)n_far; ISX_DIAG_ADD(3, n_far); }
      // lines off the fast path: one at a time, lane = row (cap or box windows)
      unsigned long long em = __ballot(have && pre.rows == -1);
      while (em) {
        const int src = __builtin_ctzll(em);
        em &= em - 1ull;
        V3 P, V;
        P.x = readlane_f64(lp.x, src); P.y = readlane_f64(lp.y, src); P.z = readlane_f64(lp.z, src);
        V.x = readlane_f64(lv.x, src); V.y = readlane_f64(lv.y, src); V.z = readlane_f64(lv.z, src);
        bin_culled<true>(d, hist, rowt, colx, P, V, lane, spl);
      }
      // fast-path lines: their rows packed over the lanes
      const int nrow = (have && pre.rows >= 0) ? ((pre.rows >> 16) - (pre.rows & 0xffff) + 1) : 0;
      int incl = nrow;
#pragma unroll
      for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const int o = __shfl_up(incl, dlt, 64);
        if (lane >= dlt) incl += o;
      }
      const int total = __builtin_amdgcn_readlane(incl, 63);
      { const int n_fast = (int)__popcll(__ballot(nrow > 0)); (void)n_fast; ISX_DIAG_ADD(0, n_fast); }
      walk_lines_packed(dfast, hist, rowt, colx, rec + 6ull * b0, pre, nrow, incl - nrow, incl, total, k.inv_dphi, lane, mrk, spl);
    }
  }
  __syncthreads();
  unsigned long long flushed = 0;
  unsigned long long* ghist = wk.hist;
  asm volatile("" : "+s"(ghist));   // (no VGPR copy of this pointer held from the prologue; global_add_u64: global_atomic, not flat)
  for (int b = tid; b < nbins; b += nthr) {
    const uint32_t c = hist[b];
    if (c) { global_add_u64(&ghist[b], (unsigned long long)c); flushed += c; }
  }
  if (flushed) atomicAdd(&wk.stats[5], flushed);
}

// ------------------------------------------------------------------ binning kernel with slot queues (default for grids
// of at most 256 x 255 bins whose LDS need fits): as isx_bin_lines_kernel, but the windows of all rows go through the
// wave's length-class queues (SlotQueues above) before they are walked.  Work unit = a quarter region = up to 256 lines.
extern "C" __global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(ISX_BIN_WAVES_PER_EU, ISX_BIN_WAVES_PER_EU)))
isx_bin_slots_kernel(const DetGrid d_arg, const Work wk) {
  extern __shared__ __align__(16) unsigned char smem[];
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
  const int nbins = d_arg.nbins;
  const size_t off_row = ((size_t)nbins * 4 + 15) & ~(size_t)15;
  double* rowt = reinterpret_cast<double*>(smem + off_row);
  ColX* colx = reinterpret_cast<ColX*>(rowt + 4 * d_arg.n_theta);
  ColP* colp = reinterpret_cast<ColP*>(colx + 2 * d_arg.n_phi);
  DetGrid* d_lds = reinterpret_cast<DetGrid*>(colp + 2 * d_arg.n_phi);
  uint32_t* wave_all = reinterpret_cast<uint32_t*>(d_lds + 1);
  const int tid = threadIdx.x, lane = tid & 63;
  const int nthr = (int)blockDim.x;
  for (int b = tid; b < nbins; b += nthr) hist[b] = 0u;
  for (int b = tid; b < 4 * d_arg.n_theta; b += nthr) rowt[b] = d_arg.rowtab[b];
  for (int b = tid; b < 2 * d_arg.n_phi; b += nthr) {
    const int j = b < d_arg.n_phi ? b : b - d_arg.n_phi;
    const int j1 = j + 1 < d_arg.n_phi ? j + 1 : 0;
    ColX e;
    e.c = d_arg.coltab[2 * j]; e.s = d_arg.coltab[2 * j + 1]; e.off4 = (uint32_t)j * 4u; e.c32 = (float)e.c; e.s32 = (float)e.s; e.pad = 0u;
    colx[b] = e;
    ColP q;
    q.c0 = e.c32; q.s0 = e.s32; q.c1 = (float)d_arg.coltab[2 * j1]; q.s1 = (float)d_arg.coltab[2 * j1 + 1];
    q.off0 = (uint32_t)j * 4u; q.off1 = (uint32_t)j1 * 4u; q.pad0 = q.pad1 = 0u;
    colp[b] = q;
  }
  if (tid == (nthr > 128 ? 128 : 0)) *d_lds = d_arg;
  // per wave: the class queues, their 8 + 8 counters, 64 owner marks (kSlotWaveWords 32-bit words)
  uint32_t* mine = wave_all + (size_t)(tid >> 6) * kSlotWaveWords;
  SlotQueues sq;
  sq.colp = colp;
  sq.q = (LdsWord*)mine;
  sq.tail = (LdsInt*)(mine + kClasses * kQueueCap);
  sq.head = sq.tail + 8;
  LdsInt* mrk = sq.head + 8;
  sq.defer = nullptr;   // (a deferred list for tiers 2 and 3 as in isx_bin_cols_kernel was measured here: 50.93 against 50.91 ms on the
                        //  BRDF source -- this kernel is VALU-bound, a diverged lane's wait for its line is covered by the other waves)
  if (lane < 16) sq.tail[lane] = 0;
  __syncthreads();
  ISX_BD_INIT(sq);
  typedef __attribute__((address_space(3))) DetGrid LdsDetGrid;
  const volatile LdsDetGrid& d = *(const volatile LdsDetGrid*)d_lds;

  const uint32_t n_regions = wk.ctr[Q_REGIONS];   // (the trace kernel of this launch has completed)
  const uint32_t ushift = 2u + wk.pad;            // a work unit = 1024 >> ushift exit lines of one region: a quarter region; a sixteenth for small launches (Work::pad)
#pragma unroll 1
  for (;;) {
    uint32_t unit = 0;
    if (lane == 0) unit = atomicAdd(&wk.ctr[Q_BIN], 1u);
    unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit);
    const uint32_t region = unit >> ushift;
    if (region >= n_regions) break;
    const uint32_t r_lines = (uint32_t)__builtin_amdgcn_readfirstlane((int)d_arg.rec_counts[region]);
    const uint32_t q_first = (unit & ((1u << ushift) - 1u)) * (kRegion >> ushift);
    const uint32_t n_lines = r_lines < q_first + (kRegion >> ushift) ? r_lines : q_first + (kRegion >> ushift);
    const double* lines = d_arg.rec_lines + 6ull * ((uint64_t)region * kRegion + q_first);   // the unit's lines
    ISX_BD_MARK(sq, 5);
    struct { int n_phi; double half_w2, portz; const double* table; } dfast;
    dfast.n_phi = d.n_phi; dfast.half_w2 = d.half_w2; dfast.portz = d_arg.portz; dfast.table = d.table;
#pragma unroll 1
    for (uint32_t b0 = q_first; b0 < n_lines; b0 += 64u) {
      const bool have = b0 + (uint32_t)lane < n_lines;
      const int first_line = (int)(b0 - q_first);
      // Three passes over the batch, one producer: the caps around the first piercing points (every line that has caps), the caps
      // around the second piercing points (the few lines whose second cap reaches detector rows), the box windows of the grazing
      // lines (no caps: a third of the BRDF source's lines).  Nothing but wave-uniform masks lives from one pass to the next: a
      // pass reads its lines (again: L2) and the grid constants (LDS) where it needs them.
      unsigned long long m_second = 0ull, m_box = 0ull;
#pragma unroll 1
      for (int pass = 0; pass < 3; ++pass) {
        const unsigned long long pm = pass == 0 ? __ballot(have) : (pass == 1 ? m_second : m_box);
        if (pm == 0ull) continue;
        const bool part = ((pm >> lane) & 1ull) != 0ull;
        BoxLine own;
        own.smax = own.smin = own.vxy = own.avz = own.ivz = own.dn = own.Hm = own.Hz = own.phin = own.rs = own.sig = 0.f;
        int ilo = 0, ihi = -1;
        bool low2 = false, box = false, far = false;
        GridConst k;   // (volatile LDS reads: derived again in every pass, never held across one)
        k.Rf = (float)d.R; k.rho = (float)d.rho_d; k.portz = (float)d.portz; k.n_theta = d.n_theta;
        k.inv_dphi = (float)d.n_phi * 0.15915494309f;
        k.inv_dth = (float)k.n_theta * 0.63661977237f;
        if (part) {
          uint32_t li = (uint32_t)(first_line + lane);
          asm volatile("" : "+v"(li));   // (the address is formed here, not hoisted out of the batch loop and spilled)
          V3 lp, lv;
          load_line(lines + 6 * li, lp, lv);
          if (pass < 2) {
            CapShared sh = prep_shared(k, lp, lv);
            if (pass == 0) {
              if (sh.kind == 0) {
                low2 = cap_is_low(k, sh, lp, lv, 1);
                if (low2 && caps_may_touch(k, sh)) { sh.kind = -1; low2 = false; }   // the whole line through the box windows
              }
              box = sh.kind == -1;
              far = sh.kind == -2;
            }
            if (sh.kind == 0) cap_rows_pre(k, lp, lv, sh, pass, own, ilo, ihi);
          } else {
            struct { int n_theta; double rho_d, R, portz; } dg;
            dg.n_theta = d.n_theta; dg.rho_d = d.rho_d; dg.R = d.R; dg.portz = d.portz;
            if (!box_line(dg, lp, lv, k.inv_dth, own, ilo, ihi)) { ilo = 0; ihi = -1; far = true; }
          }
        }
        if (pass == 0) {
          m_second = __ballot(low2);
          m_box = __ballot(box);
          ISX_BD_MARK(sq, 0);
          { const int n_fast = (int)__popcll(__ballot(part && !box && !far)); (void)n_fast; ISX_DIAG_ADD(0, n_fast); }
        }
        if (pass == 2) { const int n_box = (int)__popcll(__ballot(part && !far)); (void)n_box; ISX_DIAG_ADD(2, n_box); }
        { const int n_far = (int)__popcll(__ballot(far)); (void)n_far; ISX_DIAG_ADD(3, n_far); }
        const int nrow = (part && ihi >= ilo) ? ihi - ilo + 1 : 0;
        int incl = nrow;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
          const int o = __shfl_up(incl, dlt, 64);
          if (lane >= dlt) incl += o;
        }
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (total > 0)
          produce_rows_packed(dfast, hist, rowt, colx, lines, sq, own, ilo, nrow, incl - nrow, incl, total, pass == 2, k.inv_dphi, k.portz,
                              first_line, lane, mrk);
      }
    }
    drain_slots(dfast, hist, rowt, colx, lines, sq, 1, lane);   // the unit's leftovers, class by class
  }
  __syncthreads();
  unsigned long long flushed = 0;
  unsigned long long* ghist = wk.hist;
  asm volatile("" : "+s"(ghist));
  for (int b = tid; b < nbins; b += nthr) {
    const uint32_t c = hist[b];
    if (c) { global_add_u64(&ghist[b], (unsigned long long)c); flushed += c; }
  }
  if (flushed) atomicAdd(&wk.stats[5], flushed);
}

// ------------------------------------------------------------------ binning kernel with COLUMN slots (default for the pencil
// source on grids of at most 256 x 255 bins whose LDS need fits): the fast-path lines go through (line, column) slots and the
// wave's length-class queues (RowX / ColPre above); the few lines off the fast path take bin_culled (one line at a time, lane =
// row) as in isx_bin_lines_kernel.  Work unit = a quarter region = up to 256 lines.
extern "C" __global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(ISX_BIN_WAVES_PER_EU, ISX_BIN_WAVES_PER_EU)))
isx_bin_cols_kernel(const DetGrid d_arg, const Work wk) {
  extern __shared__ __align__(16) unsigned char smem[];
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
  const int nbins = d_arg.nbins;
  const size_t off_row = ((size_t)nbins * 4 + 15) & ~(size_t)15;
  double* rowt = reinterpret_cast<double*>(smem + off_row);
  ColX* colx = reinterpret_cast<ColX*>(rowt + 4 * d_arg.n_theta);
  RowX* rowx = reinterpret_cast<RowX*>(colx + 2 * d_arg.n_phi);
  DetGrid* d_lds = reinterpret_cast<DetGrid*>(rowx + d_arg.n_theta + 4);   // (the walk reads up to three entries past its slot's rows)
  uint32_t* wave_all = reinterpret_cast<uint32_t*>(d_lds + 1);
  const int tid = threadIdx.x, lane = tid & 63;
  const int nthr = (int)blockDim.x;
  for (int b = tid; b < nbins; b += nthr) hist[b] = 0u;
  for (int b = tid; b < 4 * d_arg.n_theta; b += nthr) rowt[b] = d_arg.rowtab[b];
  for (int b = tid; b <= d_arg.n_theta; b += nthr) {
    RowX e;
    e.S0 = e.S1 = e.C0 = e.C1 = e.T0 = e.T1 = 0.f; e.pad0 = e.pad1 = 0u;
    if (b < d_arg.n_theta) {
      const double sd = d_arg.rowtab[4 * b], cd = d_arg.rowtab[4 * b + 1];
      e.S0 = (float)sd; e.C0 = (float)cd; e.T0 = (float)(-(d_arg.R * (cd * cd)));
    }
    if (b + 1 < d_arg.n_theta) {
      const double sd = d_arg.rowtab[4 * b + 4], cd = d_arg.rowtab[4 * b + 5];
      e.S1 = (float)sd; e.C1 = (float)cd; e.T1 = (float)(-(d_arg.R * (cd * cd)));
    }
#if ISX_WALK4
    reinterpret_cast<float4*>(rowx)[b] = make_float4(e.S0, e.S1, e.C0, e.C1);
    reinterpret_cast<float2*>(reinterpret_cast<float4*>(rowx) + (d_arg.n_theta + 4))[b] = make_float2(e.T0, e.T1);
#else
    rowx[b] = e;
#endif
  }
  for (int b = tid; b < 2 * d_arg.n_phi; b += nthr) {
    const int j = b < d_arg.n_phi ? b : b - d_arg.n_phi;
    ColX e;
    e.c = d_arg.coltab[2 * j]; e.s = d_arg.coltab[2 * j + 1]; e.off4 = (uint32_t)j * 4u; e.c32 = (float)e.c; e.s32 = (float)e.s; e.pad = 0u;
    colx[b] = e;
  }
  if (tid == (nthr > 128 ? 128 : 0)) *d_lds = d_arg;
  // per wave: the class queues, their 8 + 8 counters, 64 owner marks, the deferred list
  uint32_t* mine = wave_all + (size_t)(tid >> 6) * kColWaveWords;
  SlotQueues sq;
  sq.colp = nullptr;
  sq.q = (LdsWord*)mine;
  sq.tail = (LdsInt*)(mine + kClasses * kQueueCap);
  sq.head = sq.tail + 8;
  LdsInt* mrk = sq.head + 8;
  sq.defer = (LdsWord*)(mrk + 64);
  if (lane < 16) sq.tail[lane] = 0;
  if (lane == 0) sq.defer[0] = 0u;
  __syncthreads();
  typedef __attribute__((address_space(3))) DetGrid LdsDetGrid;
  const volatile LdsDetGrid& d = *(const volatile LdsDetGrid*)d_lds;

  const uint32_t n_regions = wk.ctr[Q_REGIONS];   // (the trace kernel of this launch has completed)
  const uint32_t ushift = 2u + wk.pad;            // a work unit = 1024 >> ushift exit lines of one region: a quarter region; a sixteenth for small launches (Work::pad)
  ISX_BD_INIT(sq);
#pragma unroll 1
  for (;;) {
    uint32_t unit = 0;
    if (lane == 0) unit = atomicAdd(&wk.ctr[Q_BIN], 1u);
    unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit);
    const uint32_t region = unit >> ushift;
    if (region >= n_regions) break;
    const uint32_t r_lines = (uint32_t)__builtin_amdgcn_readfirstlane((int)d_arg.rec_counts[region]);
    const uint32_t q_first = (unit & ((1u << ushift) - 1u)) * (kRegion >> ushift);
    const uint32_t n_lines = r_lines < q_first + (kRegion >> ushift) ? r_lines : q_first + (kRegion >> ushift);
    const double* lines = d_arg.rec_lines + 6ull * ((uint64_t)region * kRegion + q_first);   // the unit's lines
    struct { int n_phi, n_theta; double half_w2, portz, R; const double* table; } dcol;
    dcol.n_phi = d.n_phi; dcol.n_theta = d.n_theta; dcol.half_w2 = d.half_w2; dcol.portz = d_arg.portz; dcol.R = d_arg.R; dcol.table = d.table;
#pragma unroll 1
    for (uint32_t b0 = q_first; b0 < n_lines; b0 += 64u) {
      const bool have = b0 + (uint32_t)lane < n_lines;
      const int first_line = (int)(b0 - q_first);
      // Three passes over the batch, one producer: the caps around the first piercing points (every line that has caps), the caps
      // around the second piercing points (the few lines whose second cap reaches detector rows: 3 % of the headline's), cap AND
      // band of the grazing lines (no caps; none in the headline, a third of the BRDF source's).
      // Nothing but wave-uniform masks (scalar registers) lives from one pass to the next: until round 4 the line (12 VGPRs), the
      // first cap (7) and the wave-uniform binary32 constants of the preparation (gfx950 has no scalar float unit: VGPRs as well)
      // were kept across the whole inlined consumer for the benefit of the second pass -- the kernel sits at its 128-VGPR limit,
      // so they went to scratch and back, 60 bytes per lane and batch: 0.8 GB of HBM writes per 5e7-ray launch next to 129.6 KB
      // of algorithmic output.  A pass now reads its lines (again: L2) and the constants (LDS) where it needs them, and the
      // second pass no longer looks at the first cap (caps_may_touch).
      unsigned long long m_second = 0ull, m_band = 0ull;
#pragma unroll 1
      for (int pass = 0; pass < 3; ++pass) {
        const unsigned long long pm = pass == 0 ? __ballot(have) : (pass == 1 ? m_second : m_band);
        if (pm == 0ull) continue;
        const bool part = ((pm >> lane) & 1ull) != 0ull;
        ColPre pre;
        pre.fx = pre.fy = pre.a = pre.cosw = 0.f; pre.jlo = 0; pre.ncol = 0; pre.kind = -2;
        BandPre bp;
        bp.ux = bp.uy = bp.uz = 0.f; bp.kap = 2.f;
        bool low2 = false, graze = false;
        if (part) {
          GridConst k;   // (volatile LDS reads: derived again here, never held across a pass)
          k.Rf = (float)d.R; k.rho = (float)d.rho_d; k.portz = (float)d.portz; k.n_theta = d.n_theta;
          k.inv_dphi = (float)d.n_phi * 0.15915494309f;
          k.inv_dth = 0.f;
          uint32_t li = (uint32_t)(first_line + lane);
          asm volatile("" : "+v"(li));   // (the address is formed here, not hoisted out of the batch loop and spilled)
          V3 lp, lv;
          load_line(lines + 6 * li, lp, lv);
          if (pass < 2) {
            CapShared sh = prep_shared(k, lp, lv);
            if (pass == 0 && sh.kind == 0) {
              low2 = cap_is_low(k, sh, lp, lv, 1);
              if (low2 && caps_may_touch(k, sh)) { sh.kind = -1; low2 = false; }   // the whole line as a grazing line
            }
            graze = sh.kind == -1;
            pre = prep_cols(k, dcol.n_phi, lp, lv, sh, pass);
          } else {
            pre = prep_band(k, dcol.n_phi, lp, lv, bp);
          }
        }
        if (pass == 0) {
          ISX_BD_MARK(sq, 0);
          m_second = __ballot(low2);
          m_band = __ballot(graze);
          { const int n_fast = (int)__popcll(__ballot(part && pre.kind == 0 && pre.ncol > 0)); (void)n_fast; ISX_DIAG_ADD(0, n_fast); }
        }
        if (pass == 2) { const int n_box = (int)__popcll(__ballot(part && pre.kind == 0)); (void)n_box; ISX_DIAG_ADD(2, n_box); }
        { const int n_far = (int)__popcll(__ballot(part && pre.kind == -2)); (void)n_far; ISX_DIAG_ADD(3, n_far); }
        const int ncol = (part && pre.kind == 0) ? pre.ncol : 0;
        int incl = ncol;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
          const int o = __shfl_up(incl, dlt, 64);
          if (lane >= dlt) incl += o;
        }
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (total == 0) continue;
        ColPre pc = pre;
        pc.ncol = ncol;
        const int n_theta = d.n_theta;
        produce_cols_packed(dcol, hist, rowt, colx, rowx, lines, sq, pc, incl - ncol, incl, total, (float)n_theta * 0.63661977237f, n_theta,
                            first_line, lane, mrk, bp, pass == 2);
      }
      // what the walks of this batch put aside for tiers 2 and 3: decided as soon as a wave-full has collected (and at the unit's end)
      if (((volatile LdsWord*)sq.defer)[0] >= kDeferFlushAt) flush_deferred(dcol, hist, rowt, colx, lines, sq, lane);
    }
    drain_cols(dcol, hist, rowt, colx, rowx, lines, sq, 1, lane);   // the unit's leftovers, class by class
    flush_deferred(dcol, hist, rowt, colx, lines, sq, lane);        // ... and what its walks put aside for tiers 2 and 3
  }
  __syncthreads();
  unsigned long long flushed = 0;
  unsigned long long* ghist = wk.hist;
  asm volatile("" : "+s"(ghist));
  for (int b = tid; b < nbins; b += nthr) {
    const uint32_t c = hist[b];
    if (c) { global_add_u64(&ghist[b], (unsigned long long)c); flushed += c; }
  }
  if (flushed) atomicAdd(&wk.stats[5], flushed);
}

// ------------------------------------------------------------------ per-ray end states (parity tests)
extern "C" __global__ void __launch_bounds__(256)
isx_endstates_kernel(const Geom g, uint64_t seed, uint64_t first, uint64_t n, int32_t* __restrict__ status,
                     int32_t* __restrict__ npts, double* __restrict__ lp, double* __restrict__ dir) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Hot h = make_hot(g);
  Ray r;
  ray_start(g, r, (uint32_t)i);
  int st;
  for (;;) {
    st = ray_step<false>(h, g, r, seed, first);
    if (st != 0 && h.source_model == 1 && !r.scattered()) { ray_rescatter(g, r, seed, first); st = 0; }
    if (st != 0) break;
  }
  status[i] = st;
  npts[i] = (int)r.j + 1 + (st == ST_EXITED ? 1 : 0);
  lp[3 * i] = r.p.x; lp[3 * i + 1] = r.p.y; lp[3 * i + 2] = r.p.z;
  dir[3 * i] = r.v.x; dir[3 * i + 1] = r.v.y; dir[3 * i + 2] = r.v.z;
}

// ------------------------------------------------------------------ device-side self test of the numeric contract
// out[k] for k in [0,n): op 0 sqrt(a), 1 a/b, 2 fma(a,b,c), 3 log_pos(a), 4/5 sincos2pi(a), 6/7 sincos_cw(a),
// 8 sqrt_unit(a), 9 neg_rcp_unit(a), 10/11 circle_point(a) cos/sin
extern "C" __global__ void isx_mathprobe_kernel(int op, const double* __restrict__ a, const double* __restrict__ b,
                                                const double* __restrict__ c, double* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s, co;
  switch (op) {
    case 0: out[i] = sqrt(a[i]); break;
    case 1: out[i] = a[i] / b[i]; break;
    case 2: out[i] = fma(a[i], b[i], c[i]); break;
    case 3: out[i] = log_pos(a[i]); break;
    case 4: sincos2pi(a[i], s, co); out[i] = s; break;
    case 5: sincos2pi(a[i], s, co); out[i] = co; break;
    case 6: sincos_cw(a[i], s, co); out[i] = s; break;
    case 8: out[i] = sqrt_unit(a[i]); break;
    case 9: out[i] = neg_rcp_unit(a[i]); break;
    case 10: circle_point(a[i], co, s); out[i] = co; break;
    case 11: circle_point(a[i], co, s); out[i] = s; break;
    case 12: case 13: case 14: { V3 v; v.x = a[i]; v.y = b[i]; v.z = c[i]; const V3 r = tv_unit_n(v); out[i] = op == 12 ? r.x : op == 13 ? r.y : r.z; break; }
    default: sincos_cw(a[i], s, co); out[i] = co; break;
  }
}

}  // namespace isx
