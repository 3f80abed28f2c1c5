/*
 * isx.h — C ABI of libisx, the MI355X-native integrating-sphere ray tracer.
 *
 * This is the drop-in boundary for ONE hot path of bdagnillo/altair-raytracing:
 * the per-ray trace loop + port-escape test + detector flux histogram
 * (SURVEY.md §8).  The reference has no FFI of its own; the entry points below
 * are what a maintainer's ROOT-macro (or cgo/ctypes) stub would bind in place of
 *
 *   AOpticsManager::TraceNonSequential(ARay&/ARayArray*)   flux_at_observer/fluxAtObserverOptimize.C:254,295
 *   isRayPassingThroughExitPort()                          flux_at_observer/fluxAtObserver.C:162-166
 *   Detector::setPosition / Detector::checkIntersection    flux_at_observer/fluxAtObserver.C:49-107
 *   hitCount++ / fraction = hit/n                          flux_at_observer/fluxAtObserverOptimize.C:309-312,571
 *   trace-once endpoint binning                            flux_at_observer/fluxAtObserverFast.C:1269-1315
 *   BRDF::SampleDirection + second trace                   flux_at_observer/nonLambertianFlux.C:147-208,253-268
 *   addDetectorDisk / isRayHittingDetector                 integratingSphereDetectorSweep.C:134-172
 *
 * Rules of the boundary: plain C types only, caller owns every host buffer,
 * the library owns device memory between isx_init()/isx_shutdown(), calls are
 * blocking unless they take a stream, no exception ever crosses the ABI, every
 * function returns 0 or a negative isx_status.  There is NO CPU fallback: if no
 * gfx950 device (or the HIP runtime) is available every compute entry point
 * returns ISX_ERR_NO_DEVICE.
 *
 * Process model: the library is a process-wide singleton bound to ONE device by isx_init()
 * (one process per GPU, as the reference is one process per run); it is not re-entrant and
 * not thread-safe.  isx_init() with another device ordinal shuts the first binding down.
 */
#ifndef ISX_H
#define ISX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: isx_config starts with struct_size (a config whose size is not the library's is refused, so a field added by a
 *    later ABI can never be mis-read silently). */
#define ISX_ABI_VERSION 3
/* Results are a pure function of (configuration, seed, ray indices) AND of the random-number layout below; a build with
 * another ISX_STREAM_VERSION gives different (equally valid) histograms for the same seed.
 * 3: Philox4x32-10, counter (ray lo, ray hi, block, stream); interaction j takes words (2(j&1), 2(j&1)+1) of block j/2;
 *    absorption and azimuth share one word; Householder cosine emission.
 * 4: the same generator and word layout; the cosine emission is n + s with s a uniform point of the unit sphere in world
 *    coordinates (first word: its z, second word: its azimuth), left un-normalised on the inner sphere, and the next wall point
 *    is p - 2 (p.v)/(v.v) v.  The explicit and the chord trace mode visit the same wall points since. */
#define ISX_STREAM_VERSION 4

typedef enum isx_status {
  ISX_OK = 0,
  ISX_ERR_NO_DEVICE = -1,   /* HIP runtime/device missing: the product never falls back to the CPU */
  ISX_ERR_BAD_CONFIG = -2,  /* geometry/grid parameters out of the supported domain */
  ISX_ERR_BAD_ARG = -3,     /* null pointer, zero size ... */
  ISX_ERR_HIP = -4,         /* a HIP call failed; isx_last_hip_error() has the code */
  ISX_ERR_NOT_INIT = -5,
  ISX_ERR_TOO_LARGE = -6    /* n_rays per call above ISX_MAX_RAYS_PER_CALL (or records above ISX_MAX_LOG_RECORDS) */
} isx_status;

/* One call may trace at most this many rays (the library cuts a call into launches of at most 2^30 rays: a lane keeps a
 * 31-bit offset from its launch's first ray, a workgroup counts in 32-bit LDS bins).  Split larger jobs over calls.
 * Ray indices are 64-bit; first_ray + n_rays must not exceed 2^64 - 1 (ISX_ERR_BAD_ARG). */
#define ISX_MAX_RAYS_PER_CALL (1ull << 40)
/* isx_exit_directions keeps at most this many 32-byte records per call (8 GiB of device memory) */
#define ISX_MAX_LOG_RECORDS (1ull << 28)

/* source_model */
#define ISX_SOURCE_PENCIL 0 /* fluxAtObserver*.C: identical rays from src along dir            */
#define ISX_SOURCE_BRDF 1   /* nonLambertianFlux.C:235-304: primary trace, BRDF re-scatter, 2nd trace */

/* surface_model */
#define ISX_SURFACE_ROBAST 0 /* ABorderSurfaceCondition: Lambertian if `lambertian`, else rough specular   */
#define ISX_SURFACE_LOBE 1   /* "nonLambertianFlux copy.C":31-70,188-221 NonLambertianSurface: cos^2 lobe
                                within 60 deg of the normal by rejection sampling (the de-facto CustomMirror) */
/* trace_mode: how a Lambertian bounce off the INNER SPHERE finds the next wall point */
#define ISX_TRACE_EXPLICIT 0 /* sample a cosine-law direction, intersect the ray with the sphere              */
#define ISX_TRACE_CHORD 1    /* integrating-sphere identity: for cosine-law emission from a point of a sphere
                                the far intersection is UNIFORM over the sphere's area, so the next wall point
                                is sampled directly (1 sqrt + 1 sincos, no direction, no intersection); the
                                direction is only formed when the point falls in the port opening.  Same
                                distribution, different random history; rim/outer-sphere/non-Lambertian
                                interactions are always explicit. */
/* hit_line_mode: which line Detector::checkIntersection sees */
#define ISX_HITLINE_LAST_SEGMENT 0 /* last point + final direction (fluxAtObserverOptimize.C:309; canonical)   */
#define ISX_HITLINE_ORIGIN_COMPAT 1 /* what fluxAtObserverFast.C:1181-1201,1285-1288 effectively used because
                                       GetPoint(nPoints-2, buf) never fills buf: start (0,0,0), direction
                                       lastPoint/|lastPoint| - reproduces the old fluxmap_traceonce_* files */

/*
 * Geometry + surface + source + detector grid.  Field meaning follows the
 * reference constants: fluxAtObserverOptimize.C:33-41 (THETA_MAX, MAX_REFLECTIONS,
 * INNER/OUTER_RADIUS, REFLECTANCE, ROUGHNESS), :192-230 (setupOpticsManager),
 * :456-461,495 (n, exitPortZ, bins, detector size), fluxAtObserver.C:352-358 (grid).
 * Lengths in cm (AOpticsManager::cm() == 1).
 */
typedef struct isx_config {
  uint32_t struct_size;  /* sizeof(isx_config) of the caller's ABI; set by isx_default_config().  Every entry
                            point refuses (ISX_ERR_BAD_CONFIG) a config whose size is not the library's.  */
  uint32_t reserved0;    /* 0 */
  double r_in;           /* TGeoSphere rmin (100.1)                                   */
  double r_out;          /* TGeoSphere rmax (101)                                     */
  double theta_max_deg;  /* TGeoSphere theta2: shell spans polar angle [0,theta_max]  */
  double reflectance;    /* AMirror::SetReflectance                                   */
  double roughness_rad;  /* ABorderSurfaceCondition::SetGaussianRoughness (sigma)     */
  double box_half;       /* TGeoBBox half edge                                        */
  int32_t lambertian;    /* ABorderSurfaceCondition::EnableLambertian                 */
  int32_t max_points;    /* AOpticsManager::SetLimit                                  */
  double src[3];         /* ARay start point                                          */
  double dir[3];         /* ARay direction (normalised by the library)                */
  int32_t n_theta;       /* detector grid rows: theta_i=(i+.5)*90/n_theta              */
  int32_t n_phi;         /* detector grid cols: phi_j=(j+.5)*360/n_phi                 */
  double det_diameter;   /* Detector::width (used as a DIAMETER, fluxAtObserver.C:106) */
  double det_distance;   /* Detector::setPosition radius (100)                        */
  double exit_port_z;    /* exitPortZ (-100); also the point the detectors face       */
  int32_t source_model;  /* ISX_SOURCE_*                                              */
  int32_t surface_model; /* ISX_SURFACE_*                                             */
  double brdf[3];        /* BRDF(roughness, specular, diffuse) nonLambertianFlux.C:211 */
  int32_t hit_line_mode; /* ISX_HITLINE_*                                             */
  int32_t trace_mode;    /* ISX_TRACE_*                                               */
} isx_config;

/* Ray census of one call (all ranks' census add up). */
typedef struct isx_stats {
  uint64_t launched;
  uint64_t exited;          /* left the world box (ARayArray::GetExited)                  */
  uint64_t counted_below_z; /* exited with lastPoint.z < exit_port_z ("rays exiting port") */
  uint64_t absorbed;
  uint64_t suspended;       /* more than max_points track points                          */
  uint64_t bin_increments;  /* sum of the histogram this call added                        */
  uint64_t wall_hits;       /* mirror interactions (bounces) traced                        */
  double t_kernel_ms;       /* HIP-event time of the kernels of this call, on their stream */
} isx_stats;

/* Fill cfg with the reference's constants for src(-60,0,-75), dir(5,0,0), port 170 deg
 * (fluxAtObserverOptimize.C:33-41,892-896; 180x90 grid, 40 cm detector at 100 cm). */
void isx_default_config(isx_config* cfg);

/* Select device `device` (ordinal as seen by HIP), create stream + workspaces.      */
int isx_init(int device);
void isx_shutdown(void);
const char* isx_strerror(int status);
int isx_last_hip_error(void);
int isx_abi_version(void);
int isx_stream_version(void);
/* Name/arch/CU count of the bound device (buf may be NULL). Returns CU count or <0. */
int isx_device_info(char* buf, int buflen);

/*
 * Trace rays [first_ray, first_ray+n_rays) of stream `seed` and add, for every
 * detector position (i,j), the number of traced rays whose final line hits that
 * detector (Detector::checkIntersection) into hits[i*n_phi+j].
 * Replaces the per-position loop fluxAtObserverOptimize.C:542-579 / the
 * trace-once loops fluxAtObserverFast.C:1143-1315.
 * hits: caller-owned host buffer [n_theta*n_phi], ZEROED by the callee.
 */
int isx_fluxmap(const isx_config* cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray,
                uint64_t* hits, isx_stats* stats);

/*
 * Same, but ACCUMULATES (+=) into a device-resident histogram d_hits
 * [n_theta*n_phi] of uint64 (e.g. a torch tensor the caller will all-reduce
 * over RCCL) on the library's stream; returns after the kernels are enqueued.
 * isx_sync() waits; isx_take_stats() then returns the census + event time of
 * everything enqueued since the previous isx_take_stats().
 */
int isx_fluxmap_device(const isx_config* cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray,
                       uint64_t* d_hits);
int isx_sync(void);
int isx_take_stats(isx_stats* stats);
/* The hipStream_t the library launches on (so callers can order work after it). */
void* isx_stream(void);
/* HIP-event times (ms) of the kernels collected by the last blocking call / isx_take_stats(), by kind: single-kernel
 * launches, the trace kernel of the two-kernel flux-map pipeline, its binning kernel.  Any pointer may be NULL. */
int isx_last_kernel_ms(double* single_ms, double* trace_ms, double* bin_ms);

/* Tuning/diagnostic switches.  None of them changes any result (a ray's history is a function of seed and index).
 *   "bin_mode"     1 (default) culled + classified binning, 0 brute-force reference-order test of every detector position
 *   "pipeline"     1 (default): the lean flux maps (headline, chord mode, BRDF source) run as two kernels -- a trace kernel
 *                  writes the exit lines (48 B per counted ray) to an HBM workspace, a binning kernel reads them; 0: one fused kernel
 *   "pipeline_chunk"  rays per trace / binning pair, default and maximum 2^26.  WORKSPACE: exit lines live in regions of 1024
 *                  slots; a chunk of n rays traced by W waves is given n/961 + W + 1 regions of 48 KB (3.4 GB for 2^26 rays),
 *                  allocated once and kept until isx_shutdown(); "overlap" keeps three of them
 *   "assist"       1 (default): trace kernels with an assist wave per workgroup -- the flux-map pipelines, the
 *                  shared-ray disc sweep and the per-position sinks (isx_fluxmap_per_position, isx_disc_sweep_per_position);
 *                  0: round 2's kernels;
 *                  "assist_block" = their workgroup size (128..768, default 768 = 11 tracer waves + 1 assist wave)
 *   "bin_slots"    1 (default): binning kernels with slot queues by window length (grids up to 256 x 255); 0: round 2's
 *   "bin_cols"     1 (default; 2 is accepted and means the same): (line, COLUMN) slots for every source -- caps of the lines that
 *                  pass near the detector sphere's centre, cap-and-band windows for grazing lines; 0: (line, row) slots
 *   "surface_pipeline"  1 (default, round 5): the cos^2-lobe and rough-specular borders (pencil source) and the origin-compat hit
 *                  line run on the assist-wave pipeline too (isx_trace_assist_lobe_kernel / ..._rough_kernel; the compat lines are
 *                  rewritten by isx_compat_lines_kernel between the trace and the binning kernel); 0: round 1's fused
 *                  isx_trace_bin_full_kernel
 *   "bin_block", "bin_blocks_per_cu"  shape of round 2's binning kernel (0 workgroups per CU = what is resident)
 *                  (round 5: "assist_block" 0 = the default again -- 768 threads, 256 for launches below 1e6 rays, 512 for the lobe /
 *                  rough-specular kernels; "rays_per_lane" > 0 sizes every grid for that many rays per tracer lane, 0 = by launch
 *                  size: 1 below 1.5e5 rays, 2 below 3e5, else 4 -- a small launch is bound by its longest ray, not by throughput)
 *   "ray_sub"      rays a wave takes off a launch's ray queue at a time (0 = default: 128)
 *   "overlap", "overlap_trace_streams"  cut a flux-map call into k chunks, binning of chunk i on a second stream while chunk
 *                  i+1 is traced (measured slower on MI355X, default 0)
 *   "disc_pipeline"  1 (default): the shared-ray disc sweep as trace kernel + disc-binning kernel (discs clustered by eight on the
 *                  host, exit segments in HBM); 0: one fused kernel
 *   "blocks_per_cu", "grid_blocks" (0 = auto), "trace_block" (64..1024, default 512), "trace_blocks_per_cu" (0 = resident)
 *                  launch shapes; "sched_mask", "sched_min": batching of the generic boundary search in the kernels without
 *                  an assist wave (every (mask+1)-th loop trip or when `min` lanes wait). */
int isx_set_option(const char* key, int64_t value);

/* Device-side probe of the numeric contract (tests): out[i] = op(a[i],b[i],c[i]) with
 * op 0 sqrt, 1 a/b, 2 fma, 3 log, 4/5 sin/cos(2*pi*a), 6/7 sin/cos(a), 8 the hot loop's sqrt for operands
 * in [2^-33,1], 9 its -1/x for |x| in [1,2] (both must equal the IEEE results),
 * 10/11 cos/sin of the emission azimuth (oracle: isxo_circle_point), 12/13/14 the x/y/z of TVector3(a,b,c).Unit() as the lobe
 * sampler forms it (must equal the plain-operation result). */
int isx_mathprobe(int op, const double* a, const double* b, const double* c, double* out, int32_t n);

/*
 * Per-ray end states, for parity tests against the oracle: status (isx_ray_status),
 * last point, final direction, number of track points.  Host buffers sized n_rays.
 */
typedef enum isx_ray_status { ISX_RAY_EXITED = 1, ISX_RAY_ABSORBED = 2, ISX_RAY_SUSPENDED = 3 } isx_ray_status;
int isx_trace_endstates(const isx_config* cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray,
                        int32_t* status, int32_t* n_points, double* last_point /*[n][3]*/,
                        double* direction /*[n][3]*/);

/*
 * Physical-disc sweep (integratingSphereDetectorSweep.C:31-105,145-172): n_disc
 * discs (centre[3], unit axis[3]) of radius `radius`, half thickness `half_thick`;
 * hits[k] = number of rays whose forward exit segment enters disc k's volume.
 */
int isx_disc_sweep(const isx_config* cfg, const double* centers_axes /*[n_disc][6]*/, int32_t n_disc,
                   double radius, double half_thick, uint64_t n_rays, uint64_t seed, uint64_t first_ray,
                   uint64_t* hits, isx_stats* stats);

/*
 * The same sweep as the reference WRITES it (integratingSphereDetectorSweep.C:54-77): every disc position k gets its
 * own `rays_per_position` fresh rays [first_ray + k*rays_per_position, +rays_per_position) and only those are tested
 * against disc k -- all positions in ONE launch (362 positions x 1e5 rays for the macro's defaults).
 */
int isx_disc_sweep_per_position(const isx_config* cfg, const double* centers_axes /*[n_disc][6]*/, int32_t n_disc,
                                double radius, double half_thick, uint64_t rays_per_position, uint64_t seed,
                                uint64_t first_ray, uint64_t* hits, isx_stats* stats);

/*
 * The reference's PER-POSITION maps: every detector position gets its own
 * `rays_per_position` fresh rays (fluxAtObserverOptimize.C:542-579, n=50000 => 8.1e8 rays
 * for 180x90).  Rays [first_ray + g*rays_per_position, +rays_per_position) belong to
 * detector group g and are tested against that group only.  fold=1: group g is bin g
 * (theta-major).  fold=2 is the "twofold" variant (fluxAtObserverFast.C:336-408,518-865):
 * group g = (i, j<n_phi/2) feeds the two detectors (i,j) and (i,j+n_phi/2).
 * Groups [first_group, first_group+n_groups) are traced (so ranks can split a map);
 * hits: host buffer [n_theta*n_phi], zeroed by the callee.
 */
int isx_fluxmap_per_position(const isx_config* cfg, uint64_t rays_per_position, int32_t fold, uint64_t first_group,
                             uint64_t n_groups, uint64_t seed, uint64_t first_ray, uint64_t* hits, isx_stats* stats);

/*
 * int traceRays(AOpticsManager*, int n, double exitPortZ, Detector&, bool) (fluxAtObserver.C:169,
 * fluxAtObserverOptimize.C:239,281): n rays against ONE detector given as x,y,z,nx,ny,nz
 * (what Detector::setPosition left in the struct) and its width.
 */
int isx_trace_rays_detector(const isx_config* cfg, const double* detector /*[6]*/, double width, uint64_t n_rays,
                            uint64_t seed, uint64_t first_ray, uint64_t* hit_count, isx_stats* stats);

/*
 * Exit-direction by-product (distributionSphereDetectorSweep.C:54,91): histogram of the z
 * component of the final direction of every ray counted below exit_port_z,
 * TH1D(nbins,-1,1) binning.  hist: host buffer [nbins], zeroed by the callee.
 */
int isx_exit_dz_hist(const isx_config* cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray, int32_t nbins,
                     uint64_t* hist, isx_stats* stats);

/*
 * Un-binned exit log (the committed 3dRayLog.txt, "# dx dy dz"): ray index and final unit
 * direction of every ray counted below exit_port_z, sorted by ray index.  capacity = room in
 * ray_ids[capacity] / directions[capacity][3] (at most ISX_MAX_LOG_RECORDS are kept, more is ISX_ERR_TOO_LARGE;
 * a capacity above n_rays is treated as n_rays); *count = number of such rays (may exceed capacity:
 * the surplus is dropped).  This is the one sink with real HBM output (32 B per exiting ray).
 */
int isx_exit_directions(const isx_config* cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray, uint64_t capacity,
                        uint64_t* ray_ids, double* directions, uint64_t* count, isx_stats* stats);

/*
 * Series driver (sweepSeries, fluxAtObserverOptimize.C:892-921 / fluxAtObserverFast.C:1641-1673):
 * n_cfg configurations sharing one detector grid, traced back to back on the device with ONE
 * host synchronisation; hits[n_cfg][n_theta*n_phi], stats[n_cfg] (t_kernel_ms = whole series).
 * Configuration k uses ray indices [first_ray + k*n_rays, +n_rays).
 */
int isx_fluxmap_series(const isx_config* cfgs, int32_t n_cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray,
                       uint64_t* hits, isx_stats* stats);

/* Host-side detector table exactly as Detector::setPosition builds it
 * (fluxAtObserver.C:49-68): out[(i*n_phi+j)*6] = x,y,z,nx,ny,nz.  No GPU needed. */
int isx_detector_table(const isx_config* cfg, double* out);

#ifdef __cplusplus
}
#endif
#endif /* ISX_H */
