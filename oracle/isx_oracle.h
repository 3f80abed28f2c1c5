/*
 * isx_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference hot path of bdagnillo/altair-raytracing
 * (SURVEY.md §8a): trace loop, port test, detector test, flux accumulation.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  Nothing under altair-raytracing_amd/ includes, links or calls it.
 *
 * PARITY STATUS: "parity unpinned" at the bit level against ROOT/ROBAST — the
 * arithmetic of AOpticsManager::TraceNonSequential lives in ROBAST (un-vendored,
 * un-versioned checkout, see flux_at_observer/nonLambertianFlux_C.d) and the
 * reference holds no tests or seeded vectors.  The restatement is pinned
 * STATISTICALLY by the reference's committed result files (tests/golden/,
 * SURVEY.md §4); Detector::setPosition/checkIntersection, which ARE in the
 * reference, are restated operation for operation.
 *
 * The struct layouts deliberately equal include/isx.h so one ctypes definition
 * serves both libraries; the code is independent.
 */
#ifndef ISX_ORACLE_H
#define ISX_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct isxo_config {
  uint32_t struct_size, reserved0;   /* layout twin of isx_config (ABI v3); not interpreted by the oracle */
  double r_in, r_out, theta_max_deg, reflectance, roughness_rad, box_half;
  int32_t lambertian, max_points;
  double src[3], dir[3];
  int32_t n_theta, n_phi;
  double det_diameter, det_distance, exit_port_z;
  int32_t source_model, surface_model;
  double brdf[3];
  int32_t hit_line_mode, trace_mode;
} isxo_config;

typedef struct isxo_stats {
  uint64_t launched, exited, counted_below_z, absorbed, suspended, bin_increments, wall_hits;
  double t_kernel_ms; /* wall time of the call, ms */
} isxo_stats;

enum { ISXO_EXITED = 1, ISXO_ABSORBED = 2, ISXO_SUSPENDED = 3 };

void isxo_default_config(isxo_config* cfg);

/* RNG + math primitives, exported so tests can pin them (KATs, libm comparison). */
void isxo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double isxo_u01(uint32_t w);
double isxo_log(double x);                            /* x in (0,1], normal            */
void isxo_sincos2pi(double u, double* s, double* c);  /* sin/cos(2*pi*u), u in [0,1)   */
void isxo_circle_point(double u, double* c, double* s); /* uniform point of the unit circle, u in (0,1) */
void isxo_sincos(double x, double* s, double* c);     /* |x| < 1e5                     */

/* Detector::setPosition (fluxAtObserver.C:49-68) for the whole grid: out[(i*n_phi+j)*6]. */
int isxo_detector_table(const isxo_config* cfg, double* out);
/* Detector::checkIntersection (fluxAtObserver.C:70-107); det = x,y,z,nx,ny,nz. */
int isxo_check_intersection(const double det[6], double width, const double last_point[3],
                            const double direction[3]);

/* Per-ray end states (for source_model 1: of the SCATTERED ray). */
int isxo_trace_endstates(const isxo_config* cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray,
                         int32_t* status, int32_t* n_points, double* last_point, double* direction);

/* Flux map: hits[n_theta*n_phi] zeroed then filled. nthreads<=0: all cores. */
int isxo_fluxmap(const isxo_config* cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray,
                 uint64_t* hits, isxo_stats* stats, int nthreads);

/* Per-position maps (fluxAtObserverOptimize.C:542-579; fold=2: fluxAtObserverFast.C:336-408). */
int isxo_fluxmap_per_position(const isxo_config* cfg, uint64_t rays_per_position, int32_t fold, uint64_t first_group,
                              uint64_t n_groups, uint64_t seed, uint64_t first_ray, uint64_t* hits, isxo_stats* stats,
                              int nthreads);
/* traceRays() against one detector (fluxAtObserver.C:169-228). */
int isxo_trace_rays_detector(const isxo_config* cfg, const double* detector, double width, uint64_t n_rays,
                             uint64_t seed, uint64_t first_ray, uint64_t* hit_count, isxo_stats* stats);

/* Physical disc sweep (integratingSphereDetectorSweep.C:134-172). */
int isxo_disc_sweep(const isxo_config* cfg, const double* centers_axes, int32_t n_disc, double radius,
                    double half_thick, uint64_t n_rays, uint64_t seed, uint64_t first_ray, uint64_t* hits,
                    isxo_stats* stats, int nthreads);
/* The same with every disc position seeing only its own rays [first_ray + k*rays_per_position, +rays_per_position)
 * (the loop integratingSphereDetectorSweep.C:54-77 as written). */
int isxo_disc_sweep_per_position(const isxo_config* cfg, const double* centers_axes, int32_t n_disc, double radius,
                                 double half_thick, uint64_t rays_per_position, uint64_t seed, uint64_t first_ray,
                                 uint64_t* hits, isxo_stats* stats, int nthreads);

/* Exit-direction by-products (distributionSphereDetectorSweep.C:61-103): for rays counted
 * below z, histogram of dz into nbins over [-1,1) and optional direction log. */
int isxo_exit_dz_hist(const isxo_config* cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray,
                      int32_t nbins, uint64_t* hist, isxo_stats* stats, int nthreads);

/* Un-binned exit log (3dRayLog.txt): ray index + final direction of every ray counted below z, in ray order. */
int isxo_exit_directions(const isxo_config* cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray, uint64_t capacity,
                         uint64_t* ray_ids, double* directions, uint64_t* count);

/* One bounce at a time (replay of a reference-side dump, tests/test_robast_dump.py): next boundary from p along v when the
 * ray sits on boundary `on` (0 none, 1 inner sphere, 2 outer sphere, 3 rim cone, 4 box) -> kind, point (v_out, if not NULL:
 * the direction the step was taken along -- v itself, or its unit vector if the step left rule S1'); the surface normal
 * there; the cosine-law emission from surface point q for the two Philox words of an interaction. */
int isxo_next_boundary(const isxo_config* cfg, const double p[3], const double v[3], int on, double q_out[3], double v_out[3]);
int isxo_surface_normal(const isxo_config* cfg, int kind, const double q[3], double n_out[3]);
int isxo_cosine_emission(const isxo_config* cfg, int kind, const double q[3], uint32_t wa, uint32_t wb, double w_out[3]);

int isxo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
