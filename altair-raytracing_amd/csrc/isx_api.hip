// isx_api.hip — C ABI of libisx (include/isx.h) over the gfx950 kernels.
//
// Host-side "geometry setup" here replaces setupOpticsManager()
// (fluxAtObserverOptimize.C:192-230) and Detector::setPosition (fluxAtObserver.C:49-68):
// it only evaluates closed-form constants and the detector table; all ray work is on the GPU.
// There is no CPU compute path: without a device every entry point returns ISX_ERR_NO_DEVICE.
#include "../../include/isx.h"
#include "isx_kernels.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <tuple>
#include <vector>

using namespace isx;

namespace {

struct State {
  bool init = false;
  int device = -1;
  int cu_count = 0;
  size_t lds_limit = 64 * 1024;   // per-workgroup LDS the device grants (160 KiB on gfx950)
  hipStream_t stream = nullptr;
  int last_hip = 0;
  // detector tables (device) + the config they were built for
  isx_config tab_cfg{};
  bool have_tab = false;
  double* d_table = nullptr;
  double* d_rowtab = nullptr;
  double* d_coltab = nullptr;
  size_t cap_bins = 0, cap_rows = 0, cap_cols = 0;
  unsigned long long* d_hist = nullptr;
  size_t cap_hist = 0;
  unsigned long long* d_stats = nullptr;  // [8]
  double* d_aux = nullptr;                // caller-supplied detector / disc lists of the current call (grown, never shrunk)
  size_t cap_aux = 0;
  // two-kernel pipeline of the headline flux map: exit lines in HBM, in regions of kRegion 48-byte slots + lines per region
  static constexpr int kRecBufs = 3;      // workspaces: chunk k uses buffer k mod 3 (the overlapped pipeline keeps up to 3 in flight)
  double* d_rec[kRecBufs] = {nullptr, nullptr, nullptr};
  uint32_t* d_rec_counts[kRecBufs] = {nullptr, nullptr, nullptr};
  size_t cap_regions[kRecBufs] = {0, 0, 0};
  // overlap > 1: a flux-map call is cut into that many chunks and the binning kernel of chunk k runs on a second stream
  // while the trace kernel of chunk k+1 runs on the first
  int overlap = 0;
  int overlap_trace_streams = 1;          // 2: consecutive trace kernels alternate between two streams (chunk k+1 fills the tail of chunk k)
  hipStream_t stream2 = nullptr, stream3 = nullptr;
  int pipeline = 1;                       // 1 (default): trace kernel -> HBM -> binning kernel (lean flux map); 0: fused kernel
  uint64_t pipe_chunk = 1ull << 26;       // rays per trace/bin pair (3.4 GB of exit-line workspace at most)
  // work-queue counters of the launches (Work::ctr): a ring of Q_WORDS-word blocks, one per launch, zeroed on the stream
  // right before its launch
  uint32_t* d_ctr = nullptr;
  size_t ctr_next = 0;
  static constexpr size_t kCtrRing = 256;
  int ray_sub = 0;                        // rays a wave takes off the queue at a time (0: by launch size)
  int bin_block = 512, bin_blocks_per_cu = 0;   // binning kernel: workgroup size, workgroups per CU in the grid (0: what is resident)
  int assist_block = ISX_ASSIST_BLOCK;          // its workgroup size: (assist_block / 64 - 1) tracer waves + 1 assist wave
  bool assist_block_set = false;                // set through isx_set_option: then it holds for small launches too (small_shape)
  int rays_per_lane = 0;                        // 0: by launch size (small_shape); > 0: the grid is sized for this many rays per tracer lane
  int disc_pipeline = 1;                        // 1 (default): the shared-ray disc sweep as assist-wave trace kernel + isx_bin_discs_kernel; 0: fused SINK_DISC kernel
  int assist = 1;                               // 1: trace kernels with an assist wave per workgroup (assist_body)
  int bin_cols = 1;                             // 1 (2: the same): isx_bin_cols_kernel ((line, column) slots) where bin_slots applies; 0: row slots
  int bin_slots = 1;                            // 1: isx_bin_slots_kernel (slot queues by window length) where the grid allows it
  int surface_pipeline = 1;                     // 1 (default): the lobe / rough-specular borders and the origin-compat hit line on the assist-wave
                                                // pipeline (round 5); 0: round 1's fused isx_trace_bin_full_kernel
  // options
  int bin_mode = 1;
  int blocks_per_cu = 1;   // 1024-thread blocks: 16 waves/CU, 4 per SIMD
  int grid_blocks = 0;  // 0 = auto
  int sched_mask = 3, sched_min = 12;   // generic boundary search: every 4th loop trip (24 bounces) or when 12 lanes wait
  // timing of enqueued-but-not-collected launches
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  struct Span { size_t a, b; int kind; };   // kind 0: single-kernel launch, 1: trace kernel of the pipeline, 2: its binning kernel
  std::vector<Span> spans;
  double last_ms[3] = {0, 0, 0};           // per kind, of the launches collected by the last collect_stats()
  // workgroup shape of the kernels that keep no LDS histogram: 512 threads (6 waves per SIMD at <= 80 VGPRs); the grid is what
  // is resident (the waves share one ray queue, so late workgroups have nothing to even out); trace_blocks_per_cu > 0 overrides
  int trace_block = 512, trace_blocks_per_cu = 0;
  // hipFuncSetAttribute(MaxDynamicSharedMemorySize) is made once per kernel and size, not once per launch
  std::map<const void*, size_t> attr_lds;
  // ... and so is the occupancy query (resident_per_cu): (kernel, workgroup size, LDS bytes) -> workgroups per CU
  std::map<std::tuple<const void*, int, size_t>, int> occ;
  // small calls are bound by host round trips: results and census go through ONE pinned staging buffer
  // (asynchronous copies, one synchronisation per call), and a blocking call that finds nothing enqueued before it skips the
  // census round trip at its start
  unsigned char* h_pin = nullptr;
  size_t cap_pin = 0;
  bool pending = false;                    // launches enqueued since the last collect_stats()
} S;

// Scratch device allocation of one call: freed on every return path.
template <class T>
struct DevBuf {
  T* p = nullptr;
  hipError_t alloc(size_t n) { return hipMalloc(&p, n * sizeof(T)); }
  ~DevBuf() { if (p) (void)hipFree(p); }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

#define HIPCHK(expr)                                 \
  do {                                               \
    hipError_t e_ = (expr);                          \
    if (e_ != hipSuccess) {                          \
      S.last_hip = (int)e_;                          \
      return ISX_ERR_HIP;                            \
    }                                                \
  } while (0)

// ABI v3: the caller's struct must be the library's (isx.h: struct_size)
bool config_abi_ok(const isx_config* c) { return c->struct_size == (uint32_t)sizeof(isx_config); }

int prepare_geom(const isx_config* c, Geom* g) {
  if (!config_abi_ok(c)) return ISX_ERR_BAD_CONFIG;
  if (!(c->r_in > 0) || !(c->r_out > c->r_in)) return ISX_ERR_BAD_CONFIG;
  if (!(c->theta_max_deg > 90.0) || !(c->theta_max_deg < 180.0)) return ISX_ERR_BAD_CONFIG;
  if (!(c->box_half > c->r_out)) return ISX_ERR_BAD_CONFIG;
  if (c->max_points < 1) return ISX_ERR_BAD_CONFIG;
  if (c->source_model != ISX_SOURCE_PENCIL && c->source_model != ISX_SOURCE_BRDF) return ISX_ERR_BAD_CONFIG;
  g->rin2 = c->r_in * c->r_in;
  g->rout2 = c->r_out * c->r_out;
  const double th = c->theta_max_deg * M_PI / 180.0;
  const double ct = std::cos(th);
  const double tt = std::tan(th);
  g->zcut_in = c->r_in * ct;
  g->zcut_out = c->r_out * ct;
  g->k2 = tt * tt;
  g->ninv_rin = -1.0 / c->r_in;
  g->inv_rout = 1.0 / c->r_out;
  g->H = c->box_half;
  g->rho = c->reflectance;
  {
    // (w + 0.5) * 2^-32 < rho  <=>  w < rho * 2^32 - 0.5 =: x (both scalings exact)  <=>  w < ceil(x) for integer w
    const double x = std::ldexp(c->reflectance, 32) - 0.5;
    g->rho_thr = !(x > 0.0) ? 0ull : (x >= 4294967296.0 ? 4294967296ull : (unsigned long long)std::ceil(x));
    g->inv_thr = g->rho_thr ? 1.0 / (double)g->rho_thr : 0.0;
    g->psi_k1 = g->inv_thr * 1.57079632679489655800e+00;
    g->psi_k0 = (0.5 * g->inv_thr - 0.5) * 1.57079632679489655800e+00;
  }
  g->sigma = c->roughness_rad;
  g->lambertian = c->lambertian;
  g->limit = c->max_points;
  g->source_model = c->source_model;
  if (c->surface_model != ISX_SURFACE_ROBAST && c->surface_model != ISX_SURFACE_LOBE) return ISX_ERR_BAD_CONFIG;
  if (c->hit_line_mode != ISX_HITLINE_LAST_SEGMENT && c->hit_line_mode != ISX_HITLINE_ORIGIN_COMPAT) return ISX_ERR_BAD_CONFIG;
  g->surface_model = c->surface_model;
  if (c->trace_mode != ISX_TRACE_EXPLICIT && c->trace_mode != ISX_TRACE_CHORD) return ISX_ERR_BAD_CONFIG;
  g->chord = c->trace_mode; g->pad2 = 0;
  g->r_in = c->r_in;
  g->sched_mask = S.sched_mask;
  g->sched_min = S.sched_min;
  for (int k = 0; k < 3; ++k) g->src[k] = c->src[k];
  const double dx = c->dir[0], dy = c->dir[1], dz = c->dir[2];
  const double mag = std::sqrt(dx * dx + dy * dy + dz * dz);
  if (!(mag > 0)) return ISX_ERR_BAD_CONFIG;
  g->dir0[0] = dx / mag; g->dir0[1] = dy / mag; g->dir0[2] = dz / mag;
  g->brdf_theta_scale = c->brdf[0] * M_PI / 6;
  const double sum = c->brdf[1] + c->brdf[2];
  g->brdf_spec = sum != 0 ? c->brdf[1] / sum : 0.0;
  return ISX_OK;
}

// Detector::setPosition, operation for operation (fluxAtObserver.C:49-68)
void det_set_position(double theta, double phi, double radius, double portz, double* d) {
  const double theta_rad = theta * M_PI / 180.0;
  const double phi_rad = phi * M_PI / 180.0;
  // the reference is built by g++ -O2 (ACLiC), which turns sin(a),cos(a) into ONE sincos(a)
  double st, ct, sp, cp;
  ::sincos(theta_rad, &st, &ct);
  ::sincos(phi_rad, &sp, &cp);
  const double x = radius * st * cp;
  const double y = radius * st * sp;
  const double z = portz - radius * ct;
  const double dx = x - 0;
  const double dy = y - 0;
  const double dz = z - (portz);
  const double mag = std::sqrt(dx * dx + dy * dy + dz * dz);
  d[0] = x; d[1] = y; d[2] = z;
  d[3] = -dy / mag; d[4] = dx / mag; d[5] = dz / mag;
}

int check_grid(const isx_config* c) {
  if (!config_abi_ok(c)) return ISX_ERR_BAD_CONFIG;
  if (c->n_theta < 1 || c->n_phi < 1) return ISX_ERR_BAD_CONFIG;
  if ((long long)c->n_theta * c->n_phi > 36000) return ISX_ERR_BAD_CONFIG;  // LDS histogram: 4 B/bin
  if (!(c->det_diameter > 0) || !(c->det_distance > 0)) return ISX_ERR_BAD_CONFIG;
  return ISX_OK;
}

void host_tables(const isx_config* c, std::vector<double>& table, std::vector<double>& rowtab,
                 std::vector<double>& coltab) {
  const int nt = c->n_theta, np = c->n_phi;
  table.resize((size_t)nt * np * 6);
  rowtab.resize((size_t)nt * 4);
  coltab.resize((size_t)np * 2);
  for (int i = 0; i < nt; ++i) {
    const double theta = (i + 0.5) * 90.0 / nt;
    const double theta_rad = theta * M_PI / 180.0;
    double st, ct;
    ::sincos(theta_rad, &st, &ct);
    rowtab[4 * i + 0] = st;
    rowtab[4 * i + 1] = ct;
    rowtab[4 * i + 2] = c->exit_port_z - c->det_distance * ct;
    rowtab[4 * i + 3] = c->det_distance * st;
    for (int j = 0; j < np; ++j) {
      const double phi = (j + 0.5) * 360.0 / np;
      det_set_position(theta, phi, c->det_distance, c->exit_port_z, &table[6 * ((size_t)i * np + j)]);
    }
  }
  for (int j = 0; j < np; ++j) {
    const double phi = (j + 0.5) * 360.0 / np;
    const double phi_rad = phi * M_PI / 180.0;
    ::sincos(phi_rad, &coltab[2 * j + 1], &coltab[2 * j + 0]);
  }
}

bool same_grid(const isx_config& a, const isx_config& b) {
  return a.n_theta == b.n_theta && a.n_phi == b.n_phi && a.det_distance == b.det_distance &&
         a.exit_port_z == b.exit_port_z;
}

int ensure_tables(const isx_config* c) {
  if (S.have_tab && same_grid(S.tab_cfg, *c)) return ISX_OK;
  std::vector<double> table, rowtab, coltab;
  host_tables(c, table, rowtab, coltab);
  HIPCHK(hipStreamSynchronize(S.stream));
  if (table.size() > S.cap_bins) {
    if (S.d_table) HIPCHK(hipFree(S.d_table));
    HIPCHK(hipMalloc(&S.d_table, table.size() * sizeof(double)));
    S.cap_bins = table.size();
  }
  if (rowtab.size() > S.cap_rows) {
    if (S.d_rowtab) HIPCHK(hipFree(S.d_rowtab));
    HIPCHK(hipMalloc(&S.d_rowtab, rowtab.size() * sizeof(double)));
    S.cap_rows = rowtab.size();
  }
  if (coltab.size() > S.cap_cols) {
    if (S.d_coltab) HIPCHK(hipFree(S.d_coltab));
    HIPCHK(hipMalloc(&S.d_coltab, coltab.size() * sizeof(double)));
    S.cap_cols = coltab.size();
  }
  HIPCHK(hipMemcpy(S.d_table, table.data(), table.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(S.d_rowtab, rowtab.data(), rowtab.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(S.d_coltab, coltab.data(), coltab.size() * sizeof(double), hipMemcpyHostToDevice));
  S.tab_cfg = *c;
  S.have_tab = true;
  return ISX_OK;
}

int get_event(hipEvent_t* ev) {
  if (S.ev_used == S.ev_pool.size()) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    S.ev_pool.push_back(e);
  }
  *ev = S.ev_pool[S.ev_used++];
  return ISX_OK;
}

// Launch shape of a SMALL launch (round 5; measured in profiles/r05_small_call_sweep.json).  A launch cannot end before its longest
// ray has -- about ln(n) / 0.0175 bounces, one after the other (620 for 5e4 rays) -- so a small launch is bound by latency, not by
// throughput: few rays per tracer lane (the bulk of the work is then short next to the longest ray) and few waves per SIMD (a
// wave that shares its SIMD with five others takes six times as long per bounce).  5e4 rays per call is the reference's own call
// size (traceRaysParallel, fluxAtObserverOptimize.C:568: 0.82 -> 0.29 ms per call).  From ~1.4e6 rays on the grid is what is resident.
struct Shape { int block; uint64_t rays_per_lane; };
Shape small_shape(uint64_t n, int block_default) {
  Shape sh;
  sh.block = (!S.assist_block_set && n < 1000000ull && block_default > 256) ? 256 : block_default;   // 3 tracer waves + the assist wave: one wave per SIMD
  sh.rays_per_lane = S.rays_per_lane > 0 ? (uint64_t)S.rays_per_lane : (n < 150000ull ? 1ull : n < 300000ull ? 2ull : 4ull);
  return sh;
}
// workgroups of a launch of n rays: `tracer_waves` waves per workgroup trace (all of them unless the workgroup has an assist wave)
int pick_grid(uint64_t n, int block = kBlock, int blocks_per_cu = 0, int tracer_waves = 0, uint64_t rays_per_lane = 4) {
  if (S.grid_blocks > 0) return S.grid_blocks;
  const int full = S.cu_count * (blocks_per_cu > 0 ? blocks_per_cu : S.blocks_per_cu);
  const uint64_t lanes = (uint64_t)(tracer_waves > 0 ? tracer_waves : block / 64) * 64ull * (rays_per_lane ? rays_per_lane : 1ull);
  const uint64_t want = (n + lanes - 1) / lanes;
  if (want < 1) return 1;
  return want < (uint64_t)full ? (int)want : full;
}

// workgroups of `fn` (workgroup size `block`, `lds` bytes of dynamic LDS) that are resident on one CU at a time
template <class F>
int resident_per_cu(F fn, int block, size_t lds) {
  const auto key = std::make_tuple((const void*)fn, block, lds);
  const auto it = S.occ.find(key);
  if (it != S.occ.end()) return it->second;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, block, lds) != hipSuccess || nb < 1) nb = 1;
  S.occ[key] = nb;
  return nb;
}

// rays a wave takes off the launch's queue at a time: results never depend on it (a ray's history is a function of its index)
uint32_t pick_sub(uint64_t n) {
  if (S.ray_sub > 0) return (uint32_t)S.ray_sub;
  (void)n;
  return 128u;   // (rays a wave takes off the launch's queue at a time; 512 until the bounce got 20 % shorter: measured at 5e7 rays
                 //  64 / 128 / 192 / 256 / 384 / 512 / 1024: 11.26 / 11.18 / 11.19 / 11.20 / 11.23 / 11.33 / 11.40 ms)
}

// the next block of queue counters, zeroed on the stream ahead of the launch that uses it
int next_ctr(uint32_t** ctr) {
  uint32_t* c = S.d_ctr + (S.ctr_next++ % State::kCtrRing) * Q_WORDS;
  HIPCHK(hipMemsetAsync(c, 0, Q_WORDS * sizeof(uint32_t), S.stream));
  *ctr = c;
  S.pending = true;   // (every launch takes a block: from here on the stream holds work whose census has not been collected)
  return ISX_OK;
}

// pinned staging buffer of at least `bytes` bytes: [0, 64) the census words, [64, ...) a call's result
int ensure_pin(size_t bytes) {
  bytes += 64 + 4096;   // (+ the last 4 KB: upload_aux's slot for a short list)
  if (bytes > S.cap_pin) {
    HIPCHK(hipStreamSynchronize(S.stream));
    if (S.h_pin) HIPCHK(hipHostFree(S.h_pin));
    S.h_pin = nullptr; S.cap_pin = 0;
    const size_t cap = bytes < (1u << 18) ? (1u << 18) : bytes;
    HIPCHK(hipHostMalloc((void**)&S.h_pin, cap, hipHostMallocDefault));
    S.cap_pin = cap;
  }
  return ISX_OK;
}
// D2H of a call's result: enqueued into the staging buffer (no synchronisation here: collect_stats() has the one of the call);
// fetch_result() copies it out once the stream has been synchronised
int stage_result(const void* dev, size_t bytes) {
  const int rc = ensure_pin(bytes);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(S.h_pin + 64, dev, bytes, hipMemcpyDeviceToHost, S.stream));
  return ISX_OK;
}
void fetch_result(void* host, size_t bytes) { std::memcpy(host, S.h_pin + 64, bytes); }

// one launch addresses its rays by 31-bit offsets from its first ray: larger jobs are cut into launches of this many rays
constexpr uint64_t kLaunchMax = 1ull << 30;

int ensure_pipeline(size_t rays, size_t waves, int buf = 0, size_t slot_doubles = 6);
// the disc list of the current isx_disc_sweep call as isx_bin_discs_kernel wants it (upload_discs_clustered below)
struct DiscClusters { size_t n = 0, off_ordered = 0, off_clusters = 0, off_perm = 0; int n_clusters = 0; } g_disc_clusters;

// enqueue one persistent kernel accumulating into d_hist (device) and S.d_stats
struct PerPos { uint64_t map_first = 0, rays_per_group = 0; int fold = 1; const double* d_table = nullptr; double width = 0; };
struct LogSink { double* rec = nullptr; unsigned long long* count = nullptr; uint64_t cap = 0; };

int enqueue(int sink, const isx_config* c, uint64_t n, uint64_t seed, uint64_t first, unsigned long long* d_hist,
            int nbins_override, const double* d_discs, double disc_r, double disc_h, const PerPos* pp = nullptr,
            const LogSink* lg = nullptr, unsigned long long* d_stats = nullptr) {
  Geom g;
  int rc = prepare_geom(c, &g);
  if (rc) return rc;
  if (n > ISX_MAX_RAYS_PER_CALL) return ISX_ERR_TOO_LARGE;
  if (first > UINT64_MAX - n) return ISX_ERR_BAD_ARG;   // first + n (the exclusive end of the index range) must be representable
  DetGrid d;
  std::memset(&d, 0, sizeof(d));
  d.portz = c->exit_port_z;
  d.hit_line_mode = c->hit_line_mode;
  size_t lds = 0;
  if (sink == SINK_FLUX) {
    rc = check_grid(c);
    if (rc) return rc;
    rc = ensure_tables(c);
    if (rc) return rc;
    d.n_theta = c->n_theta; d.n_phi = c->n_phi; d.nbins = c->n_theta * c->n_phi; d.bin_mode = S.bin_mode;
    d.half_w2 = (c->det_diameter / 2) * (c->det_diameter / 2);
    d.rho_d = c->det_diameter / 2;
    d.R = c->det_distance;
    d.table = S.d_table; d.rowtab = S.d_rowtab; d.coltab = S.d_coltab;
    lds = (((size_t)d.nbins * 4 + 15) & ~(size_t)15) + (size_t)(4 * d.n_theta) * 8 + (size_t)(2 * d.n_phi) * sizeof(ColX) + 64 + sizeof(Geom) + sizeof(DetGrid);
  } else if (sink == SINK_PERPOS) {
    if (!pp || pp->rays_per_group < 1 || (pp->fold != 1 && pp->fold != 2)) return ISX_ERR_BAD_ARG;
    if (pp->d_table) {  // caller-supplied detector list (traceRays with one Detector)
      if (nbins_override < 1 || nbins_override > 36000 || pp->fold != 1) return ISX_ERR_BAD_ARG;
      d.n_theta = nbins_override; d.n_phi = 1; d.nbins = nbins_override;
      d.table = pp->d_table;
      d.half_w2 = (pp->width / 2) * (pp->width / 2);
    } else {
      rc = check_grid(c);
      if (rc) return rc;
      if (pp->fold == 2 && (c->n_phi % 2) != 0) return ISX_ERR_BAD_CONFIG;
      rc = ensure_tables(c);
      if (rc) return rc;
      d.n_theta = c->n_theta; d.n_phi = c->n_phi; d.nbins = c->n_theta * c->n_phi;
      d.table = S.d_table;
      d.half_w2 = (c->det_diameter / 2) * (c->det_diameter / 2);
    }
    d.map_first = pp->map_first; d.rays_per_group = pp->rays_per_group; d.fold = pp->fold;
    lds = (((size_t)d.nbins * 4 + 15) & ~(size_t)15) + 64 + sizeof(Geom) + sizeof(DetGrid);
  } else if (sink == SINK_DISCPOS) {
    if (!pp || pp->rays_per_group < 1 || nbins_override < 1 || nbins_override > 36000) return ISX_ERR_BAD_ARG;
    d.nbins = nbins_override;
    d.discs = d_discs; d.disc_r = disc_r; d.disc_h = disc_h;
    d.map_first = pp->map_first; d.rays_per_group = pp->rays_per_group; d.fold = 1;
    lds = (((size_t)d.nbins * 4 + 15) & ~(size_t)15) + 64 + sizeof(Geom) + sizeof(DetGrid);
  } else if (sink == SINK_LOG) {
    if (!lg || !lg->rec || !lg->count) return ISX_ERR_BAD_ARG;
    d.nbins = 1;
    d.log_rec = lg->rec; d.log_count = lg->count; d.log_cap = lg->cap;
    lds = 16 + 64 + sizeof(Geom) + sizeof(DetGrid);
  } else {
    if (nbins_override < 1 || nbins_override > 36000) return ISX_ERR_BAD_ARG;
    d.nbins = nbins_override;
    d.discs = d_discs; d.disc_r = disc_r; d.disc_h = disc_h;
    lds = (((size_t)d.nbins * 4 + 15) & ~(size_t)15) + 64 + sizeof(Geom) + sizeof(DetGrid);
  }
  // the per-block histogram (+ tables) must fit the workgroup's LDS: a grid too fine for that is a configuration error
  if (lds > S.lds_limit) return ISX_ERR_BAD_CONFIG;
  if (sink == SINK_FLUX) {
    // room for the per-lane exit-line records of the lean kernels (16 + 4 B per lane) and the long-row list of the column
    // walk (4 B per lane), if the grid leaves it
    const size_t stage = 16 + (size_t)kBlock * 24;
    if (lds + stage <= S.lds_limit) { lds += stage; d.rec_stage = 1; }
  }
  if (n == 0) return ISX_OK;
  Work wk;
  wk.seed = seed; wk.first = first; wk.n = n; wk.hist = d_hist; wk.stats = d_stats ? d_stats : S.d_stats;
  wk.ctr = nullptr; wk.sub = 0; wk.pad = 0;
  // the lean kernel serves the headline configuration; anything else takes the full-featured variant
  const bool lambert = c->lambertian && c->surface_model == ISX_SURFACE_ROBAST;
  const bool compat = c->hit_line_mode != ISX_HITLINE_LAST_SEGMENT;
  const bool lean_surface = lambert && !compat;
  const bool lean = lean_surface && c->source_model == ISX_SOURCE_PENCIL;
  const bool chord = lean && c->trace_mode == ISX_TRACE_CHORD;
  const bool brdf = lean_surface && c->source_model == ISX_SOURCE_BRDF && c->trace_mode != ISX_TRACE_CHORD;
  // what the assist-wave pipeline serves beyond that (round 5; `surface_pipeline` = 0: round 1's fused kernel as before): the
  // origin-compat hit line (the assist wave writes the line the binning kernel is to see) and the two other border models with
  // the pencil source -- the chord identity is a property of the Lambertian border, so trace_mode says nothing there
  const bool sp = S.surface_pipeline != 0;
  const bool p_lean = lambert && c->source_model == ISX_SOURCE_PENCIL && (!compat || sp);
  const bool p_chord = p_lean && c->trace_mode == ISX_TRACE_CHORD;
  const bool p_brdf = lambert && c->source_model == ISX_SOURCE_BRDF && c->trace_mode != ISX_TRACE_CHORD && (!compat || sp);
  const bool p_lobe = sp && c->surface_model == ISX_SURFACE_LOBE && c->source_model == ISX_SOURCE_PENCIL;
  const bool p_rough = sp && c->surface_model == ISX_SURFACE_ROBAST && !c->lambertian && c->source_model == ISX_SOURCE_PENCIL;
  // kernel variant: lean builds serve the headline surface/source configuration, the full builds everything else
  typedef void (*KernelFn)(const Geom, const DetGrid, const Work);
  const bool lean_explicit = lean && !chord;
  KernelFn fn;
  switch (sink) {
    case SINK_FLUX: fn = chord ? isx_trace_bin_chord_kernel : lean ? isx_trace_bin_kernel : brdf ? isx_trace_bin_brdf_kernel : isx_trace_bin_full_kernel; break;
    case SINK_DZ: fn = lean_explicit ? isx_trace_dz_lean_kernel : isx_trace_dz_kernel; break;
    case SINK_DISC: fn = lean_explicit ? isx_trace_disc_lean_kernel : isx_trace_disc_kernel; break;
    case SINK_PERPOS: fn = lean_explicit ? isx_trace_perpos_lean_kernel : isx_trace_perpos_kernel; break;
    case SINK_DISCPOS: fn = lean_explicit ? isx_trace_discpos_lean_kernel : isx_trace_discpos_kernel; break;
    default: fn = lean_explicit ? isx_trace_log_lean_kernel : isx_trace_log_kernel; break;
  }
  // Workgroup shape.  The kernels that keep the 64.8 KB LDS histogram run one 1024-thread workgroup per CU (4 waves per SIMD,
  // 128 VGPRs).  The lean trace-only kernels need 75-90 VGPRs and almost no LDS: as 512-thread workgroups they reach 5-6 waves
  // per SIMD (measured: 18.8 -> 17.5 ms for 5e7 rays, 299 -> 282 ms for the 8.1e8-ray per-position map).
  const bool small = lean_explicit && (sink == SINK_PERPOS || sink == SINK_DISCPOS || sink == SINK_DZ || sink == SINK_LOG);
  const int block = small ? S.trace_block : kBlock;
  auto span = [&](int kind, hipEvent_t* first_ev) -> int {   // [previous event, new event) is one kernel of `kind`
    hipEvent_t e;
    if (first_ev) { int r = get_event(first_ev); if (r) return r; HIPCHK(hipEventRecord(*first_ev, S.stream)); return ISX_OK; }
    int r = get_event(&e); if (r) return r;
    HIPCHK(hipEventRecord(e, S.stream));
    S.spans.push_back({S.ev_used - 2, S.ev_used - 1, kind});
    return ISX_OK;
  };
  hipEvent_t e0;
  // ---- two-kernel pipeline (lean flux maps: headline, chord mode, BRDF source): trace kernel -> exit lines in HBM -> binning kernel, chunk by chunk
  const bool pipe_assist = (p_lean || p_brdf || p_lobe || p_rough) && S.assist != 0;   // (the kernels without an assist wave know the lean cases only)
  if (sink == SINK_FLUX && (lean || brdf || pipe_assist) && S.pipeline && S.bin_mode != 0) {
    // trace kernel: with an assist wave per workgroup (ISX_ASSIST_BLOCK threads; isx_kernels.hpp: assist_body) or without
    const bool assist = S.assist != 0;
    const KernelFn rec_fn = assist ? (p_lobe ? isx_trace_assist_lobe_kernel : p_rough ? isx_trace_assist_rough_kernel :
                                      p_chord ? isx_trace_assist_chord_kernel : p_brdf ? isx_trace_assist_brdf_kernel : isx_trace_assist_kernel)
                                   : (chord ? isx_trace_rec_chord_kernel : brdf ? isx_trace_rec_brdf_kernel : isx_trace_rec_kernel);
    const size_t lds_trace = 16 + 64 + sizeof(Geom) + sizeof(DetGrid) +
                             (assist ? 16 + sizeof(AssistQueues) + (size_t)(kResumeCap + kPendCap) * 64 : 0);
    const size_t lds_tables = (((size_t)d.nbins * 4 + 15) & ~(size_t)15) + (size_t)(4 * d.n_theta) * 8 + (size_t)(2 * d.n_phi) * sizeof(ColX) +
                              sizeof(DetGrid) + 16;
    // the binning kernel with slot queues (1024-thread workgroups: 61 KB of queues next to the histogram) if the grid fits its
    // 32-bit slot records and the LDS; else the one without
    const bool slots = S.bin_slots && S.bin_mode == 1 && d.n_theta <= 256 && d.n_phi <= 255 &&
                       lds_tables + (size_t)(2 * d.n_phi) * sizeof(ColP) + (size_t)(kBlock / 64) * kSlotWaveWords * 4 <= S.lds_limit;
    // ... and with COLUMN slots (default for every source since round 4: grazing lines are column slots as well -- prep_band;
    // bin_cols = 0 keeps the row slots of isx_bin_slots_kernel)
    const bool cols = slots && S.bin_cols &&
                      lds_tables + (size_t)(d.n_theta + 4) * sizeof(RowX) + (size_t)(kBlock / 64) * kColWaveWords * 4 <= S.lds_limit;
    typedef void (*BinFn)(const DetGrid, const Work);
    const BinFn bin_fn = cols ? isx_bin_cols_kernel : slots ? isx_bin_slots_kernel : isx_bin_lines_kernel;
    const uint64_t chunk0 = n < S.pipe_chunk ? n : S.pipe_chunk;
    // (the lobe / rough-specular kernels hold 87-93 VGPRs -- four waves per SIMD -- and hand over as often as the lean one at a third
    //  of its pace: 512-thread workgroups, 7 tracer waves per assist wave and two workgroups per CU, measured 47.1 against 48.3 ms
    //  and 33.4 against 34.0 ms for 5e7 rays, profiles/r05_surface_shapes.json)
    const int ablock = (!S.assist_block_set && (p_lobe || p_rough)) ? 512 : S.assist_block;
    const Shape shp = assist ? small_shape(chunk0, ablock) : Shape{S.trace_block, 4};
    const int pblock = shp.block, bblock = slots ? kBlock : S.bin_block;
    const int ptracers = assist ? pblock / 64 - 1 : pblock / 64;
    const size_t lds_bin = lds_tables + (cols ? (size_t)(d.n_theta + 4) * sizeof(RowX) + (size_t)(bblock / 64) * kColWaveWords * 4
                                         : slots ? (size_t)(2 * d.n_phi) * sizeof(ColP) + (size_t)(bblock / 64) * kSlotWaveWords * 4 : (size_t)(bblock / 64) * 128 * 4);
    if (lds_bin <= S.lds_limit) {
      const uint64_t chunk = n < S.pipe_chunk ? n : S.pipe_chunk;
      if (S.attr_lds[(const void*)rec_fn] != lds_trace) {   // (once per kernel and size, not once per launch)
        HIPCHK(hipFuncSetAttribute((const void*)rec_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_trace));
        S.attr_lds[(const void*)rec_fn] = lds_trace;
      }
      if (S.attr_lds[(const void*)bin_fn] != lds_bin) {
        HIPCHK(hipFuncSetAttribute((const void*)bin_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bin));
        S.attr_lds[(const void*)bin_fn] = lds_bin;
      }
      // grids: what is resident, fewer for a small chunk (>= 16 rays per lane; a region of exit lines per binning wave)
      const int tres = S.trace_blocks_per_cu > 0 ? S.trace_blocks_per_cu : resident_per_cu(rec_fn, pblock, lds_trace);
      const int bres = S.bin_blocks_per_cu > 0 ? S.bin_blocks_per_cu : resident_per_cu(bin_fn, bblock, lds_bin);
      // one trace launch + one binning launch of `cnt` rays from `off`, through workspace `buf`, on streams st / sb
      auto launch_pair = [&](uint64_t off, uint64_t cnt, int buf, hipStream_t st, hipStream_t sb, hipEvent_t traced) -> int {
        if (cnt > kLaunchMax) return ISX_ERR_TOO_LARGE;   // (cannot happen: pipeline_chunk <= 2^26)
        Work w2 = wk;
        w2.first = first + off; w2.n = cnt; w2.sub = pick_sub(cnt);
        // (work units of the binning kernel: quarter regions of 256 exit lines; sixteenths -- 64 lines, one batch -- for a small
        //  launch, whose few thousand lines then spread over as many waves as there are batches)
        w2.pad = cnt < 1000000ull ? 2u : 0u;
        int r = next_ctr(&w2.ctr); if (r) return r;
        DetGrid dt = d;              // the trace kernel keeps no histogram
        dt.nbins = 1; dt.n_theta = 0; dt.n_phi = 0;
        dt.rec_lines = S.d_rec[buf]; dt.rec_counts = S.d_rec_counts[buf];
        const int tgrid = pick_grid(cnt, pblock, tres, ptracers, shp.rays_per_lane);
        hipLaunchKernelGGL(rec_fn, dim3(tgrid), dim3(pblock), lds_trace, st, g, dt, w2);
        HIPCHK(hipGetLastError());
        if (traced) { HIPCHK(hipEventRecord(traced, st)); HIPCHK(hipStreamWaitEvent(sb, traced, 0)); }
        else { r = span(1, nullptr); if (r) return r; }
        if (compat && S.bin_mode != 2) {   // ISX_HITLINE_ORIGIN_COMPAT: the lines the binning kernel is to see (isx_compat_lines_kernel)
          hipLaunchKernelGGL(isx_compat_lines_kernel, dim3(S.cu_count * 8), dim3(256), 0, sb, S.d_rec[buf], S.d_rec_counts[buf], w2.ctr);
          HIPCHK(hipGetLastError());
        }
        if (S.bin_mode != 2) {       // bin_mode 2: diagnostic, trace only
          DetGrid db = d;
          db.rec_lines = S.d_rec[buf]; db.rec_counts = S.d_rec_counts[buf];
          // work units of the binning kernel = quarter regions: at most cnt / 256 full ones + the open region every writing wave of
          // the trace kernel leaves behind (one per workgroup with an assist wave) -- a wave per unit, a workgroup per bblock / 64 units
          const uint64_t writers = assist ? (uint64_t)tgrid : (uint64_t)tgrid * (uint64_t)(pblock / 64);
          const uint64_t units = cnt / (kRegion >> (2u + w2.pad)) + writers;
          const uint64_t bwant = (units + (uint64_t)(bblock / 64) - 1) / (uint64_t)(bblock / 64);
          const int bfull = S.cu_count * bres;
          const int gb = S.grid_blocks > 0 ? S.grid_blocks : (bwant < (uint64_t)bfull ? (int)bwant : bfull);
          hipLaunchKernelGGL(bin_fn, dim3(gb), dim3(bblock), lds_bin, sb, db, w2);
          HIPCHK(hipGetLastError());
          if (!traced) { r = span(2, nullptr); if (r) return r; }
        }
        return ISX_OK;
      };
      const int cgrid = pick_grid(chunk, pblock, tres, ptracers, shp.rays_per_lane);
      if (S.overlap > 1 && S.bin_mode == 1 && n >= (uint64_t)S.overlap * 65536) {
        // ---- overlapped: chunk k is binned on the second stream while chunk k+1 is traced on the first; chunk k+3 reuses the
        // workspace of chunk k.  (Timing: one wall-clock span around everything; the kernels' own times overlap.)
        if (!S.stream2) HIPCHK(hipStreamCreateWithFlags(&S.stream2, hipStreamNonBlocking));
        if (!S.stream3) HIPCHK(hipStreamCreateWithFlags(&S.stream3, hipStreamNonBlocking));
        const uint64_t per = ((n + (uint64_t)S.overlap - 1) / (uint64_t)S.overlap + 63) & ~63ull;
        const uint64_t oc = per < S.pipe_chunk ? per : S.pipe_chunk;
        const int ogrid = pick_grid(oc, pblock, tres, ptracers, shp.rays_per_lane);
        for (int b = 0; b < State::kRecBufs; ++b) { rc = ensure_pipeline((size_t)oc, (size_t)ogrid * (pblock / 64), b); if (rc) return rc; }
        rc = span(0, &e0); if (rc) return rc;
        std::vector<hipEvent_t> binned;
        HIPCHK(hipStreamWaitEvent(S.stream3, e0, 0));             // (what the caller enqueued before this call comes first)
        int k = 0;
        for (uint64_t off = 0; off < n; off += oc, ++k) {
          hipEvent_t traced, done;
          rc = get_event(&traced); if (rc) return rc;
          rc = get_event(&done); if (rc) return rc;
          const hipStream_t st = (S.overlap_trace_streams == 2 && (k & 1)) ? S.stream3 : S.stream;
          if (k >= State::kRecBufs) HIPCHK(hipStreamWaitEvent(st, binned[k - State::kRecBufs], 0));
          {   // the queue counters are zeroed on the stream that launches the trace kernel
            const hipStream_t keep = S.stream;
            S.stream = st;
            rc = launch_pair(off, n - off < oc ? n - off : oc, k % State::kRecBufs, st, S.stream2, traced);
            S.stream = keep;
            if (rc) return rc;
          }
          HIPCHK(hipEventRecord(done, S.stream2));
          binned.push_back(done);
        }
        HIPCHK(hipStreamWaitEvent(S.stream, binned.back(), 0));   // the caller's stream sees the finished histogram
        hipEvent_t e1;
        rc = get_event(&e1); if (rc) return rc;
        HIPCHK(hipEventRecord(e1, S.stream));
        size_t ia = 0, ib = 0;
        for (size_t i = 0; i < S.ev_used; ++i) { if (S.ev_pool[i] == e0) ia = i; if (S.ev_pool[i] == e1) ib = i; }
        S.spans.push_back({ia, ib, 0});
        return ISX_OK;
      }
      rc = ensure_pipeline((size_t)chunk, (size_t)cgrid * (pblock / 64));
      if (rc) return rc;
      rc = span(0, &e0); if (rc) return rc;
      for (uint64_t off = 0; off < n; off += chunk) {
        rc = launch_pair(off, n - off < chunk ? n - off : chunk, 0, S.stream, S.stream, nullptr); if (rc) return rc;
      }
      return ISX_OK;
    }
  }
  // ---- the shared-ray disc sweep as a pipeline as well: assist-wave trace kernel -> exit segments in HBM (8 doubles each) ->
  // disc-binning kernel (lane = segment, the discs one after the other)
  // (a disc list whose histogram + cluster table + per-wave lists exceed the workgroup's LDS -- above ~16 000 discs on gfx950 --
  //  takes the fused SINK_DISC kernel below, which needs the histogram only)
  const size_t lds_disc_bin = (((size_t)d.nbins * 4 + 15) & ~(size_t)15) + (size_t)g_disc_clusters.n_clusters * 16 +
                              (d.nbins <= kDiscsInLds ? (size_t)d.nbins * 48 + (size_t)((d.nbins + 1) & ~1) * 4 : 0) +
                              (size_t)(kDiscBinBlock / 64) * (64 * 7 + kPairCap / 2) * sizeof(double);
  if (sink == SINK_DISC && lean_explicit && S.pipeline && S.assist && S.disc_pipeline && g_disc_clusters.n == (size_t)d.nbins &&
      d_discs == S.d_aux && lds_disc_bin <= S.lds_limit) {
    const Shape shp = small_shape(n < S.pipe_chunk ? n : S.pipe_chunk, S.assist_block);
    const int pblock = shp.block, bblock = kDiscBinBlock;
    const size_t lds_trace = 16 + 64 + sizeof(Geom) + sizeof(DetGrid) + 16 + sizeof(AssistQueues) + (size_t)(kResumeCap + kPendCap) * 64;
    const size_t lds_bin = lds_disc_bin;
    const KernelFn rec_fn = isx_trace_assist_disc_kernel;
    if (S.attr_lds[(const void*)rec_fn] != lds_trace) {
      HIPCHK(hipFuncSetAttribute((const void*)rec_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_trace));
      S.attr_lds[(const void*)rec_fn] = lds_trace;
    }
    if (S.attr_lds[(const void*)isx_bin_discs_kernel] != lds_bin) {
      HIPCHK(hipFuncSetAttribute((const void*)isx_bin_discs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bin));
      S.attr_lds[(const void*)isx_bin_discs_kernel] = lds_bin;
    }
    const int tres = resident_per_cu(rec_fn, pblock, lds_trace), bres = resident_per_cu(isx_bin_discs_kernel, bblock, lds_bin);
    const uint64_t chunk = n < S.pipe_chunk ? n : S.pipe_chunk;
    rc = ensure_pipeline((size_t)chunk, (size_t)pick_grid(chunk, pblock, tres, pblock / 64 - 1, shp.rays_per_lane) * (pblock / 64), 0, 8);
    if (rc) return rc;
    rc = span(0, &e0); if (rc) return rc;
    for (uint64_t off = 0; off < n; off += chunk) {
      const uint64_t cnt = n - off < chunk ? n - off : chunk;
      if (cnt > kLaunchMax) return ISX_ERR_TOO_LARGE;     // (cannot happen: pipeline_chunk <= 2^26)
      Work w2 = wk;
      w2.first = first + off; w2.n = cnt; w2.sub = pick_sub(cnt);
      w2.pad = cnt < 1000000ull ? 2u : 0u;   // (64-segment work units for a small launch, as for the flux maps)
      rc = next_ctr(&w2.ctr); if (rc) return rc;
      DetGrid dt = d;
      dt.rec_lines = S.d_rec[0]; dt.rec_counts = S.d_rec_counts[0];
      dt.discs = S.d_aux + g_disc_clusters.off_ordered;              // (the binning kernel walks the discs in cluster order)
      dt.clusters = reinterpret_cast<const float*>(S.d_aux + g_disc_clusters.off_clusters);
      dt.disc_perm = reinterpret_cast<const int*>(S.d_aux + g_disc_clusters.off_perm);
      dt.n_clusters = g_disc_clusters.n_clusters;
      const int tgrid = pick_grid(cnt, pblock, tres, pblock / 64 - 1, shp.rays_per_lane);
      hipLaunchKernelGGL(rec_fn, dim3(tgrid), dim3(pblock), lds_trace, S.stream, g, dt, w2);
      HIPCHK(hipGetLastError());
      rc = span(1, nullptr); if (rc) return rc;
      const uint64_t bwant = (cnt / (kRegion >> (2u + w2.pad)) + (uint64_t)tgrid + (uint64_t)(bblock / 64) - 1) / (uint64_t)(bblock / 64);   // (work units per workgroup, as above)
      const int bfull = S.cu_count * bres;
      const int gb = S.grid_blocks > 0 ? S.grid_blocks : (bwant < (uint64_t)bfull ? (int)bwant : bfull);
      hipLaunchKernelGGL(isx_bin_discs_kernel, dim3(gb), dim3(bblock), lds_bin, S.stream, dt, w2);
      HIPCHK(hipGetLastError());
      rc = span(2, nullptr); if (rc) return rc;
    }
    return ISX_OK;
  }
  // ---- the per-position sinks (one launch for all positions: the reference's 12 524 s map, the macro's own disc loop) with an
  // assist wave per workgroup as well: the one exact test per exiting ray is per-lane work the assist wave does on the spot
  // (round 5: the lobe / rough-specular borders as well -- SINK_PERPOS only, last-segment hit line: the assist wave's exact test takes
  //  the line as it is)
  const bool pp_surface = sink == SINK_PERPOS && (p_lobe || p_rough) && !compat;
  if ((sink == SINK_PERPOS || sink == SINK_DISCPOS) && (lean_explicit || pp_surface) && S.pipeline && S.assist) {
    const KernelFn afn = sink == SINK_DISCPOS ? isx_trace_assist_discpos_kernel :
                         (pp_surface ? (p_lobe ? isx_trace_assist_perpos_lobe_kernel : isx_trace_assist_perpos_rough_kernel) : isx_trace_assist_perpos_kernel);
    const Shape shp = small_shape(n < kLaunchMax ? n : kLaunchMax, (!S.assist_block_set && pp_surface) ? 512 : S.assist_block);
    const int pblock = shp.block;
    const size_t lds_trace = 16 + 64 + sizeof(Geom) + sizeof(DetGrid) + 16 + sizeof(AssistQueues) + (size_t)(kResumeCap + kPendCap) * 64;
    if (S.attr_lds[(const void*)afn] != lds_trace) {
      HIPCHK(hipFuncSetAttribute((const void*)afn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_trace));
      S.attr_lds[(const void*)afn] = lds_trace;
    }
    const int tres = S.trace_blocks_per_cu > 0 ? S.trace_blocks_per_cu : resident_per_cu(afn, pblock, lds_trace);
    rc = span(0, &e0); if (rc) return rc;
    for (uint64_t off = 0; off < n; off += kLaunchMax) {
      Work w2 = wk;
      w2.first = first + off; w2.n = n - off < kLaunchMax ? n - off : kLaunchMax; w2.sub = pick_sub(w2.n);
      rc = next_ctr(&w2.ctr); if (rc) return rc;
      hipLaunchKernelGGL(afn, dim3(pick_grid(w2.n, pblock, tres, pblock / 64 - 1, shp.rays_per_lane)), dim3(pblock), lds_trace, S.stream, g, d, w2);
      HIPCHK(hipGetLastError());
    }
    return span(0, nullptr);
  }
  if (S.attr_lds[(const void*)fn] != lds) {
    HIPCHK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    S.attr_lds[(const void*)fn] = lds;
  }
  const int bpc = small ? (S.trace_blocks_per_cu > 0 ? S.trace_blocks_per_cu : resident_per_cu(fn, block, lds)) : 0;
  rc = span(0, &e0); if (rc) return rc;
  for (uint64_t off = 0; off < n; off += kLaunchMax) {
    Work w2 = wk;
    w2.first = first + off; w2.n = n - off < kLaunchMax ? n - off : kLaunchMax; w2.sub = pick_sub(w2.n);
    rc = next_ctr(&w2.ctr); if (rc) return rc;
    hipLaunchKernelGGL(fn, dim3(pick_grid(w2.n, block, bpc)), dim3(block), lds, S.stream, g, d, w2);
    HIPCHK(hipGetLastError());
  }
  return span(0, nullptr);
}

// device copy of a caller's detector / disc list in the pooled buffer (no hipMalloc/hipFree per call)
int upload_aux(const double* host, size_t n_doubles) {
  if (n_doubles > S.cap_aux) {
    HIPCHK(hipStreamSynchronize(S.stream));
    if (S.d_aux) HIPCHK(hipFree(S.d_aux));
    S.d_aux = nullptr; S.cap_aux = 0;
    const size_t cap = n_doubles < 4096 ? 4096 : n_doubles;
    HIPCHK(hipMalloc(&S.d_aux, cap * sizeof(double)));
    S.cap_aux = cap;
  }
  if (n_doubles * sizeof(double) <= 4096 && ensure_pin(1u << 17) == ISX_OK) {
    // a short list (traceRays' one detector): through the far end of the pinned staging buffer -- the copy engine reads it when the
    // stream gets there, the caller's `host` is free at once, and nothing waits (every blocking call ends with a synchronisation,
    // so the next call cannot overwrite it too early)
    unsigned char* slot = S.h_pin + S.cap_pin - 4096;
    std::memcpy(slot, host, n_doubles * sizeof(double));
    HIPCHK(hipMemcpyAsync(S.d_aux, slot, n_doubles * sizeof(double), hipMemcpyHostToDevice, S.stream));
    return ISX_OK;
  }
  HIPCHK(hipMemcpyAsync(S.d_aux, host, n_doubles * sizeof(double), hipMemcpyHostToDevice, S.stream));
  HIPCHK(hipStreamSynchronize(S.stream));   // `host` may be a temporary of the caller
  return ISX_OK;
}

// The disc list of isx_disc_sweep for isx_bin_discs_kernel: the discs in spatial (Morton) order, eight to a cluster, with the ball
// that holds the bounding balls of a cluster's discs.  Device layout behind the caller's own list (6 n doubles) in the pooled
// aux buffer: ordered discs (6 n doubles) | clusters (4 floats each) | permutation (n ints).

int upload_discs_clustered(const double* ca, size_t n, double radius, double half_thick) {
  g_disc_clusters = DiscClusters();
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (size_t k = 0; k < n; ++k)
    for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], ca[6 * k + a]); hi[a] = std::max(hi[a], ca[6 * k + a]); }
  std::vector<std::pair<uint32_t, int>> order(n);
  for (size_t k = 0; k < n; ++k) {
    uint32_t code = 0;
    for (int a = 0; a < 3; ++a) {
      const double span = hi[a] - lo[a];
      uint32_t q = span > 0 ? (uint32_t)std::min(1023.0, std::floor((ca[6 * k + a] - lo[a]) / span * 1024.0)) : 0u;
      for (int bit = 0; bit < 10; ++bit) code |= ((q >> bit) & 1u) << (3 * bit + a);   // Morton interleave
    }
    order[k] = {code, (int)k};
  }
  std::stable_sort(order.begin(), order.end());
  const size_t ncl = (n + 7) / 8;
  std::vector<double> ordered(6 * n);
  std::vector<float> clusters(4 * ncl);
  std::vector<int> perm(n);
  const double ball = std::sqrt(radius * radius + half_thick * half_thick);
  for (size_t j = 0; j < n; ++j) {
    perm[j] = order[j].second;
    std::memcpy(&ordered[6 * j], ca + 6 * (size_t)perm[j], 6 * sizeof(double));
  }
  for (size_t c = 0; c < ncl; ++c) {
    const size_t j0 = 8 * c, j1 = std::min(n, j0 + 8);
    double ctr[3] = {0, 0, 0};
    for (size_t j = j0; j < j1; ++j)
      for (int a = 0; a < 3; ++a) ctr[a] += ordered[6 * j + a] / (double)(j1 - j0);
    float cf[3] = {(float)ctr[0], (float)ctr[1], (float)ctr[2]};
    double r = 0;
    for (size_t j = j0; j < j1; ++j) {   // against the ROUNDED centre the kernel will use
      const double dx = ordered[6 * j] - (double)cf[0], dy = ordered[6 * j + 1] - (double)cf[1], dz = ordered[6 * j + 2] - (double)cf[2];
      r = std::max(r, std::sqrt(dx * dx + dy * dy + dz * dz));
    }
    clusters[4 * c + 0] = cf[0]; clusters[4 * c + 1] = cf[1]; clusters[4 * c + 2] = cf[2];
    clusters[4 * c + 3] = std::nextafter((float)((r + ball) * (1.0 + 1e-6)), INFINITY);
  }
  const size_t off_ordered = 6 * n, off_clusters = 12 * n, off_perm = off_clusters + 2 * ncl;
  const size_t total = off_perm + (n + 1) / 2 + 1;
  std::vector<double> blob(total, 0.0);
  std::memcpy(blob.data(), ca, 6 * n * sizeof(double));
  std::memcpy(blob.data() + off_ordered, ordered.data(), 6 * n * sizeof(double));
  std::memcpy(blob.data() + off_clusters, clusters.data(), clusters.size() * sizeof(float));
  std::memcpy(blob.data() + off_perm, perm.data(), n * sizeof(int));
  const int rc = upload_aux(blob.data(), total);
  if (rc) return rc;
  g_disc_clusters.n = n; g_disc_clusters.off_ordered = off_ordered; g_disc_clusters.off_clusters = off_clusters;
  g_disc_clusters.off_perm = off_perm; g_disc_clusters.n_clusters = (int)ncl;
  return ISX_OK;
}

// workspace of the two-kernel pipeline for a chunk of `rays` rays traced by `waves` waves: every region but a wave's last is
// closed with more than kRegion - 64 lines in it, and a launch cannot have more lines than rays (isx_kernels.hpp: kRegion)
int ensure_pipeline(size_t rays, size_t waves, int buf, size_t slot_doubles) {
  const size_t regions = (rays / (kRegion - 63) + waves + 1) * slot_doubles / 6 + 1;   // (capacity is counted in 6-double slots)
  if (regions > S.cap_regions[buf]) {
    HIPCHK(hipStreamSynchronize(S.stream));
    if (S.stream2) HIPCHK(hipStreamSynchronize(S.stream2));
    if (S.d_rec[buf]) HIPCHK(hipFree(S.d_rec[buf]));
    if (S.d_rec_counts[buf]) HIPCHK(hipFree(S.d_rec_counts[buf]));
    S.d_rec[buf] = nullptr; S.d_rec_counts[buf] = nullptr; S.cap_regions[buf] = 0;
    HIPCHK(hipMalloc(&S.d_rec[buf], regions * kRegion * 6 * sizeof(double)));
    HIPCHK(hipMalloc(&S.d_rec_counts[buf], regions * sizeof(uint32_t)));
    S.cap_regions[buf] = regions;
  }
  return ISX_OK;
}

int ensure_hist(size_t nb) {
  if (nb > S.cap_hist) {
    HIPCHK(hipStreamSynchronize(S.stream));
    if (S.d_hist) HIPCHK(hipFree(S.d_hist));
    HIPCHK(hipMalloc(&S.d_hist, nb * sizeof(unsigned long long)));
    S.cap_hist = nb;
  }
  return ISX_OK;
}

int collect_stats(isx_stats* out) {
  // the census through the pinned staging buffer: copy and re-zero enqueued, ONE synchronisation (it also covers a result staged
  // by stage_result()); a call that finds nothing enqueued since the last collection has nothing to wait for
  if (!S.pending && S.spans.empty() && !out) return ISX_OK;
  int rcp = ensure_pin(0);
  if (rcp) return rcp;
  HIPCHK(hipMemcpyAsync(S.h_pin, S.d_stats, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, S.stream));
  HIPCHK(hipMemsetAsync(S.d_stats, 0, 8 * sizeof(unsigned long long), S.stream));   // (ahead of whatever this stream launches next)
  HIPCHK(hipStreamSynchronize(S.stream));
  S.pending = false;
  double ms = 0;
  S.last_ms[0] = S.last_ms[1] = S.last_ms[2] = 0;
  for (const State::Span& sp : S.spans) {
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, S.ev_pool[sp.a], S.ev_pool[sp.b]));
    ms += t;
    S.last_ms[sp.kind] += t;
  }
  S.spans.clear();
  S.ev_used = 0;
  unsigned long long h[8];
  std::memcpy(h, S.h_pin, sizeof(h));
  // stats[7]: a wave of an assist-wave trace kernel gave up a bounded wait (its results are incomplete): never seen, never silent
  if (h[7] != 0) { S.last_hip = (int)hipErrorLaunchFailure; return ISX_ERR_HIP; }
  if (out) {
    out->launched = h[0]; out->exited = h[1]; out->counted_below_z = h[2]; out->absorbed = h[3];
    out->suspended = h[4]; out->bin_increments = h[5]; out->wall_hits = h[6];
    out->t_kernel_ms = ms;
  }
  return ISX_OK;
}

}  // namespace

extern "C" {

int isx_abi_version(void) { return ISX_ABI_VERSION; }
int isx_stream_version(void) { return ISX_STREAM_VERSION; }

void isx_default_config(isx_config* c) {
  if (!c) return;
  std::memset(c, 0, sizeof(*c));
  c->struct_size = (uint32_t)sizeof(isx_config);
  // fluxAtObserverOptimize.C:33-41 and sweepSeries() :892-896
  c->r_in = 100.1; c->r_out = 101.0; c->theta_max_deg = 170.0;
  c->reflectance = 0.99; c->roughness_rad = 0.01; c->box_half = 300.0;
  c->lambertian = 1; c->max_points = 50000;
  c->src[0] = -60; c->src[1] = 0; c->src[2] = -75;
  c->dir[0] = 5; c->dir[1] = 0; c->dir[2] = 0;
  c->n_theta = 180; c->n_phi = 90;
  c->det_diameter = 40.0; c->det_distance = 100.0; c->exit_port_z = -100.0;
  c->source_model = ISX_SOURCE_PENCIL;
  c->brdf[0] = 0.3; c->brdf[1] = 0.4; c->brdf[2] = 0.6;  // nonLambertianFlux.C:211
}

const char* isx_strerror(int s) {
  switch (s) {
    case ISX_OK: return "ok";
    case ISX_ERR_NO_DEVICE: return "no HIP device available (libisx has no CPU fallback)";
    case ISX_ERR_BAD_CONFIG: return "configuration outside the supported domain";
    case ISX_ERR_BAD_ARG: return "bad argument";
    case ISX_ERR_HIP: return "HIP runtime error (see isx_last_hip_error)";
    case ISX_ERR_NOT_INIT: return "isx_init() has not been called";
    case ISX_ERR_TOO_LARGE: return "too many rays for one call";
    default: return "unknown status";
  }
}

int isx_last_hip_error(void) { return S.last_hip; }

int isx_init(int device) {
  if (S.init) {
    if (device == S.device) return ISX_OK;
    isx_shutdown();
  }
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) { S.last_hip = (int)e; return ISX_ERR_NO_DEVICE; }
  if (device < 0 || device >= count) return ISX_ERR_BAD_ARG;
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  S.cu_count = prop.multiProcessorCount;
  {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && v > 0) S.lds_limit = (size_t)v;
    if (prop.sharedMemPerBlock > S.lds_limit) S.lds_limit = prop.sharedMemPerBlock;
  }
  // from here on a failure releases what was created (isx_shutdown() works on a partially built state)
  S.device = device;
  S.init = true;
  S.have_tab = false;
  e = hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&S.d_stats, 8 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(S.d_stats, 0, 8 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMalloc(&S.d_ctr, State::kCtrRing * Q_WORDS * sizeof(uint32_t));
  if (e != hipSuccess) { isx_shutdown(); S.last_hip = (int)e; return ISX_ERR_HIP; }
  if (const char* m = std::getenv("ISX_BIN_MODE")) {
    const int v = std::atoi(m);
    if (v >= 0 && v <= 2) S.bin_mode = v;   // same domain as isx_set_option("bin_mode")
  }
  return ISX_OK;
}

void isx_shutdown(void) {
  if (!S.init) return;
  if (S.stream) (void)hipStreamSynchronize(S.stream);
  if (S.stream2) { (void)hipStreamSynchronize(S.stream2); (void)hipStreamDestroy(S.stream2); S.stream2 = nullptr; }
  if (S.stream3) { (void)hipStreamSynchronize(S.stream3); (void)hipStreamDestroy(S.stream3); S.stream3 = nullptr; }
  for (hipEvent_t e : S.ev_pool) (void)hipEventDestroy(e);
  S.ev_pool.clear();
  S.ev_used = 0;
  S.spans.clear();
  if (S.d_table) (void)hipFree(S.d_table);
  if (S.d_rowtab) (void)hipFree(S.d_rowtab);
  if (S.d_coltab) (void)hipFree(S.d_coltab);
  if (S.d_hist) (void)hipFree(S.d_hist);
  if (S.d_stats) (void)hipFree(S.d_stats);
  if (S.d_aux) (void)hipFree(S.d_aux);
  S.d_aux = nullptr; S.cap_aux = 0;
  if (S.h_pin) (void)hipHostFree(S.h_pin);
  S.h_pin = nullptr; S.cap_pin = 0; S.pending = false; S.occ.clear();
  for (int b = 0; b < State::kRecBufs; ++b) {
    if (S.d_rec[b]) (void)hipFree(S.d_rec[b]);
    if (S.d_rec_counts[b]) (void)hipFree(S.d_rec_counts[b]);
    S.d_rec[b] = nullptr; S.d_rec_counts[b] = nullptr; S.cap_regions[b] = 0;
  }
  if (S.d_ctr) (void)hipFree(S.d_ctr);
  S.d_ctr = nullptr; S.ctr_next = 0;
  S.attr_lds.clear();
  S.d_table = S.d_rowtab = S.d_coltab = nullptr;
  S.d_hist = S.d_stats = nullptr;
  S.cap_bins = S.cap_rows = S.cap_cols = S.cap_hist = 0;
  if (S.stream) (void)hipStreamDestroy(S.stream);
  S.stream = nullptr;
  S.init = false;
  S.have_tab = false;
}

int isx_device_info(char* buf, int buflen) {
  if (!S.init) return ISX_ERR_NOT_INIT;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, S.device));
  if (buf && buflen > 0) std::snprintf(buf, (size_t)buflen, "%s %s cu=%d", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return prop.multiProcessorCount;
}

int isx_set_option(const char* key, int64_t value) {
  if (!key) return ISX_ERR_BAD_ARG;
  if (!std::strcmp(key, "bin_mode")) { if (value < 0 || value > 2) return ISX_ERR_BAD_ARG; S.bin_mode = (int)value; return ISX_OK; }
  if (!std::strcmp(key, "blocks_per_cu")) { if (value < 1 || value > 8) return ISX_ERR_BAD_ARG; S.blocks_per_cu = (int)value; return ISX_OK; }
  if (!std::strcmp(key, "sched_mask")) { if (value < 0 || value > 255) return ISX_ERR_BAD_ARG; S.sched_mask = (int)value; return ISX_OK; }
  if (!std::strcmp(key, "sched_min")) { if (value < 1 || value > 65) return ISX_ERR_BAD_ARG; S.sched_min = (int)value; return ISX_OK; }
  if (!std::strcmp(key, "trace_blocks_per_cu")) { if (value < 0 || value > 32) return ISX_ERR_BAD_ARG; S.trace_blocks_per_cu = (int)value; return ISX_OK; }
  if (!std::strcmp(key, "trace_block")) { if (value != 64 && value != 128 && value != 256 && value != 512 && value != 1024) return ISX_ERR_BAD_ARG; S.trace_block = (int)value; return ISX_OK; }
  if (!std::strcmp(key, "bin_blocks_per_cu")) { if (value < 0 || value > 32) return ISX_ERR_BAD_ARG; S.bin_blocks_per_cu = (int)value; return ISX_OK; }
  if (!std::strcmp(key, "bin_block")) { if (value != 256 && value != 512 && value != 1024) return ISX_ERR_BAD_ARG; S.bin_block = (int)value; return ISX_OK; }
  if (!std::strcmp(key, "overlap_trace_streams")) { if (value < 1 || value > 2) return ISX_ERR_BAD_ARG; S.overlap_trace_streams = (int)value; return ISX_OK; }
  if (!std::strcmp(key, "overlap")) { if (value < 0 || value > 64) return ISX_ERR_BAD_ARG; S.overlap = (int)value; return ISX_OK; }
  // ("assist_block" 0: back to the default -- 768 threads, 256 for launches below 1e6 rays)
  if (!std::strcmp(key, "assist_block")) {
    if (value == 0) { S.assist_block = ISX_ASSIST_BLOCK; S.assist_block_set = false; return ISX_OK; }
    if (value < 128 || value > ISX_ASSIST_BLOCK || value % 64) return ISX_ERR_BAD_ARG;
    S.assist_block = (int)value; S.assist_block_set = true; return ISX_OK;
  }
  if (!std::strcmp(key, "rays_per_lane")) { if (value < 0 || value > 4096) return ISX_ERR_BAD_ARG; S.rays_per_lane = (int)value; return ISX_OK; }
  if (!std::strcmp(key, "disc_pipeline")) { if (value < 0 || value > 1) return ISX_ERR_BAD_ARG; S.disc_pipeline = (int)value; return ISX_OK; }
  if (!std::strcmp(key, "assist")) { if (value < 0 || value > 1) return ISX_ERR_BAD_ARG; S.assist = (int)value; return ISX_OK; }
  if (!std::strcmp(key, "bin_cols")) { if (value < 0 || value > 2) return ISX_ERR_BAD_ARG; S.bin_cols = (int)value; return ISX_OK; }
  if (!std::strcmp(key, "bin_slots")) { if (value < 0 || value > 1) return ISX_ERR_BAD_ARG; S.bin_slots = (int)value; return ISX_OK; }
  if (!std::strcmp(key, "surface_pipeline")) { if (value < 0 || value > 1) return ISX_ERR_BAD_ARG; S.surface_pipeline = (int)value; return ISX_OK; }
  if (!std::strcmp(key, "ray_sub")) { if (value < 0 || value > (1 << 20)) return ISX_ERR_BAD_ARG; S.ray_sub = (int)value; return ISX_OK; }
  if (!std::strcmp(key, "pipeline")) { if (value < 0 || value > 1) return ISX_ERR_BAD_ARG; S.pipeline = (int)value; return ISX_OK; }
  // (a launch addresses its rays by 30-bit offsets -- bits 30 and 31 of Ray::ido are flags in the queue records -- and counts
  //  them in 32 bits: the documented maximum of a chunk is 2^26 rays, 3.4 GB of exit-line workspace)
  if (!std::strcmp(key, "pipeline_chunk")) { if (value < 4096 || value > (1ll << 26)) return ISX_ERR_BAD_ARG; S.pipe_chunk = (uint64_t)value; return ISX_OK; }
  if (!std::strcmp(key, "grid_blocks")) { if (value < 0 || value > 65535) return ISX_ERR_BAD_ARG; S.grid_blocks = (int)value; return ISX_OK; }
  return ISX_ERR_BAD_ARG;
}

void* isx_stream(void) { return (void*)S.stream; }

int isx_last_kernel_ms(double* single_ms, double* trace_ms, double* bin_ms) {
  if (!S.init) return ISX_ERR_NOT_INIT;
  if (single_ms) *single_ms = S.last_ms[0];
  if (trace_ms) *trace_ms = S.last_ms[1];
  if (bin_ms) *bin_ms = S.last_ms[2];
  return ISX_OK;
}

int isx_sync(void) {
  if (!S.init) return ISX_ERR_NOT_INIT;
  HIPCHK(hipStreamSynchronize(S.stream));
  return ISX_OK;
}

int isx_take_stats(isx_stats* stats) {
  if (!S.init) return ISX_ERR_NOT_INIT;
  return collect_stats(stats);
}

int isx_fluxmap_device(const isx_config* cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray, uint64_t* d_hits) {
  if (!S.init) return ISX_ERR_NOT_INIT;
  if (!cfg || !d_hits) return ISX_ERR_BAD_ARG;
  return enqueue(SINK_FLUX, cfg, n_rays, seed, first_ray, (unsigned long long*)d_hits, 0, nullptr, 0, 0);
}

int isx_fluxmap(const isx_config* cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray, uint64_t* hits,
                isx_stats* stats) {
  if (!S.init) return ISX_ERR_NOT_INIT;
  if (!cfg || !hits) return ISX_ERR_BAD_ARG;
  int rc = check_grid(cfg);
  if (rc) return rc;
  const size_t nb = (size_t)cfg->n_theta * cfg->n_phi;
  rc = ensure_hist(nb);
  if (rc) return rc;
  // census of anything enqueued earlier must not leak into this call
  rc = collect_stats(nullptr);
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(S.d_hist, 0, nb * sizeof(unsigned long long), S.stream));
  rc = enqueue(SINK_FLUX, cfg, n_rays, seed, first_ray, S.d_hist, 0, nullptr, 0, 0);
  if (rc) return rc;
  rc = stage_result(S.d_hist, nb * sizeof(unsigned long long));
  if (rc) return rc;
  rc = collect_stats(stats);
  if (rc == ISX_OK) fetch_result(hits, nb * sizeof(unsigned long long));
  return rc;
}

int isx_trace_endstates(const isx_config* cfg, uint64_t n, uint64_t seed, uint64_t first, int32_t* status,
                        int32_t* n_points, double* last_point, double* direction) {
  if (!S.init) return ISX_ERR_NOT_INIT;
  if (!cfg || !status || !n_points || !last_point || !direction) return ISX_ERR_BAD_ARG;
  if (n == 0) return ISX_OK;
  if (n > (1ull << 28)) return ISX_ERR_TOO_LARGE;
  Geom g;
  int rc = prepare_geom(cfg, &g);
  if (rc) return rc;
  DevBuf<int32_t> b_st, b_np;
  DevBuf<double> b_lp, b_dir;
  HIPCHK(b_st.alloc(n)); HIPCHK(b_np.alloc(n)); HIPCHK(b_lp.alloc(n * 3)); HIPCHK(b_dir.alloc(n * 3));
  int32_t *d_st = b_st.p, *d_np = b_np.p;
  double *d_lp = b_lp.p, *d_dir = b_dir.p;
  const int blk = 256;
  const unsigned grid = (unsigned)((n + blk - 1) / blk);
  hipLaunchKernelGGL(isx_endstates_kernel, dim3(grid), dim3(blk), 0, S.stream, g, seed, first, n, d_st, d_np, d_lp, d_dir);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(status, d_st, n * 4, hipMemcpyDeviceToHost, S.stream));
  HIPCHK(hipMemcpyAsync(n_points, d_np, n * 4, hipMemcpyDeviceToHost, S.stream));
  HIPCHK(hipMemcpyAsync(last_point, d_lp, n * 24, hipMemcpyDeviceToHost, S.stream));
  HIPCHK(hipMemcpyAsync(direction, d_dir, n * 24, hipMemcpyDeviceToHost, S.stream));
  HIPCHK(hipStreamSynchronize(S.stream));
  return ISX_OK;
}

int isx_mathprobe(int op, const double* a, const double* b, const double* c, double* out, int32_t n) {
  if (!S.init) return ISX_ERR_NOT_INIT;
  if (!a || !out || n <= 0) return ISX_ERR_BAD_ARG;
  DevBuf<double> ba, bb, bc, bo;
  const size_t bytes = (size_t)n * 8;
  HIPCHK(ba.alloc(n)); HIPCHK(bb.alloc(n)); HIPCHK(bc.alloc(n)); HIPCHK(bo.alloc(n));
  HIPCHK(hipMemcpy(ba.p, a, bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(bb.p, b ? b : a, bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(bc.p, c ? c : a, bytes, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(isx_mathprobe_kernel, dim3((n + 255) / 256), dim3(256), 0, S.stream, op, ba.p, bb.p, bc.p, bo.p, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(S.stream));
  HIPCHK(hipMemcpy(out, bo.p, bytes, hipMemcpyDeviceToHost));
  return ISX_OK;
}

int isx_detector_table(const isx_config* cfg, double* out) {
  if (!cfg || !out) return ISX_ERR_BAD_ARG;
  if (!config_abi_ok(cfg) || cfg->n_theta < 1 || cfg->n_phi < 1) return ISX_ERR_BAD_CONFIG;
  for (int i = 0; i < cfg->n_theta; ++i) {
    const double theta = (i + 0.5) * 90.0 / cfg->n_theta;
    for (int j = 0; j < cfg->n_phi; ++j) {
      const double phi = (j + 0.5) * 360.0 / cfg->n_phi;
      det_set_position(theta, phi, cfg->det_distance, cfg->exit_port_z, out + 6 * ((size_t)i * cfg->n_phi + j));
    }
  }
  return ISX_OK;
}

int isx_disc_sweep(const isx_config* cfg, const double* centers_axes, int32_t n_disc, double radius, double half_thick,
                   uint64_t n_rays, uint64_t seed, uint64_t first_ray, uint64_t* hits, isx_stats* stats) {
  if (!S.init) return ISX_ERR_NOT_INIT;
  if (!cfg || !centers_axes || !hits || n_disc < 1) return ISX_ERR_BAD_ARG;
  if (cfg->source_model != ISX_SOURCE_PENCIL) return ISX_ERR_BAD_CONFIG;
  if (!(radius > 0) || !(half_thick > 0)) return ISX_ERR_BAD_ARG;
  int rc = ensure_hist((size_t)n_disc);
  if (rc) return rc;
  rc = collect_stats(nullptr);
  if (rc) return rc;
  g_disc_clusters = DiscClusters();
  rc = S.disc_pipeline ? upload_discs_clustered(centers_axes, (size_t)n_disc, radius, half_thick) : upload_aux(centers_axes, (size_t)n_disc * 6);
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(S.d_hist, 0, (size_t)n_disc * sizeof(unsigned long long), S.stream));
  rc = enqueue(SINK_DISC, cfg, n_rays, seed, first_ray, S.d_hist, n_disc, S.d_aux, radius, half_thick);
  if (rc == ISX_OK) {
    const hipError_t e = hipMemcpyAsync(hits, S.d_hist, (size_t)n_disc * sizeof(unsigned long long), hipMemcpyDeviceToHost, S.stream);
    if (e != hipSuccess) { S.last_hip = (int)e; rc = ISX_ERR_HIP; }
  }
  const int rc2 = collect_stats(stats);
  return rc ? rc : rc2;
}

int isx_disc_sweep_per_position(const isx_config* cfg, const double* centers_axes, int32_t n_disc, double radius,
                                double half_thick, uint64_t rays_per_position, uint64_t seed, uint64_t first_ray,
                                uint64_t* hits, isx_stats* stats) {
  if (!S.init) return ISX_ERR_NOT_INIT;
  if (!cfg || !centers_axes || !hits || n_disc < 1 || rays_per_position < 1) return ISX_ERR_BAD_ARG;
  if (cfg->source_model != ISX_SOURCE_PENCIL) return ISX_ERR_BAD_CONFIG;
  if (!(radius > 0) || !(half_thick > 0)) return ISX_ERR_BAD_ARG;
  if ((uint64_t)n_disc > ISX_MAX_RAYS_PER_CALL / rays_per_position) return ISX_ERR_TOO_LARGE;
  int rc = ensure_hist((size_t)n_disc);
  if (rc) return rc;
  rc = collect_stats(nullptr);
  if (rc) return rc;
  rc = upload_aux(centers_axes, (size_t)n_disc * 6);
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(S.d_hist, 0, (size_t)n_disc * sizeof(unsigned long long), S.stream));
  PerPos pp;
  pp.map_first = first_ray; pp.rays_per_group = rays_per_position; pp.fold = 1;
  rc = enqueue(SINK_DISCPOS, cfg, (uint64_t)n_disc * rays_per_position, seed, first_ray, S.d_hist, n_disc, S.d_aux, radius,
               half_thick, &pp);
  if (rc == ISX_OK) {
    const hipError_t e = hipMemcpyAsync(hits, S.d_hist, (size_t)n_disc * sizeof(unsigned long long), hipMemcpyDeviceToHost, S.stream);
    if (e != hipSuccess) { S.last_hip = (int)e; rc = ISX_ERR_HIP; }
  }
  const int rc2 = collect_stats(stats);
  return rc ? rc : rc2;
}

int isx_fluxmap_per_position(const isx_config* cfg, uint64_t rays_per_position, int32_t fold, uint64_t first_group,
                             uint64_t n_groups, uint64_t seed, uint64_t first_ray, uint64_t* hits, isx_stats* stats) {
  if (!S.init) return ISX_ERR_NOT_INIT;
  if (!cfg || !hits || rays_per_position < 1) return ISX_ERR_BAD_ARG;
  int rc = check_grid(cfg);
  if (rc) return rc;
  if (fold != 1 && fold != 2) return ISX_ERR_BAD_ARG;
  const size_t nb = (size_t)cfg->n_theta * cfg->n_phi;
  const uint64_t groups_total = nb / (uint64_t)fold;
  if (first_group > groups_total || n_groups > groups_total - first_group) return ISX_ERR_BAD_ARG;
  if (n_groups > ISX_MAX_RAYS_PER_CALL / rays_per_position) return ISX_ERR_TOO_LARGE;
  rc = ensure_hist(nb);
  if (rc) return rc;
  rc = collect_stats(nullptr);
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(S.d_hist, 0, nb * sizeof(unsigned long long), S.stream));
  PerPos pp;
  pp.map_first = first_ray; pp.rays_per_group = rays_per_position; pp.fold = fold;
  rc = enqueue(SINK_PERPOS, cfg, n_groups * rays_per_position, seed, first_ray + first_group * rays_per_position,
               S.d_hist, 0, nullptr, 0, 0, &pp);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(hits, S.d_hist, nb * sizeof(unsigned long long), hipMemcpyDeviceToHost, S.stream));
  return collect_stats(stats);
}

int isx_trace_rays_detector(const isx_config* cfg, const double* detector, double width, uint64_t n_rays, uint64_t seed,
                            uint64_t first_ray, uint64_t* hit_count, isx_stats* stats) {
  if (!S.init) return ISX_ERR_NOT_INIT;
  if (!cfg || !detector || !hit_count || !(width > 0)) return ISX_ERR_BAD_ARG;
  int rc = ensure_hist(1);
  if (rc) return rc;
  rc = collect_stats(nullptr);
  if (rc) return rc;
  rc = upload_aux(detector, 6);
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(S.d_hist, 0, sizeof(unsigned long long), S.stream));
  PerPos pp;
  pp.map_first = first_ray; pp.rays_per_group = n_rays > 0 ? n_rays : 1; pp.fold = 1; pp.d_table = S.d_aux; pp.width = width;
  rc = enqueue(SINK_PERPOS, cfg, n_rays, seed, first_ray, S.d_hist, 1, nullptr, 0, 0, &pp);
  unsigned long long h = 0;
  if (rc == ISX_OK) rc = stage_result(S.d_hist, sizeof(h));
  const int rc2 = collect_stats(stats);
  if (rc == ISX_OK && rc2 == ISX_OK) fetch_result(&h, sizeof(h));
  *hit_count = h;
  return rc ? rc : rc2;
}

int isx_exit_directions(const isx_config* cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray, uint64_t capacity,
                        uint64_t* ray_ids, double* directions, uint64_t* count, isx_stats* stats) {
  if (!S.init) return ISX_ERR_NOT_INIT;
  if (!cfg || !ray_ids || !directions || !count || capacity < 1) return ISX_ERR_BAD_ARG;
  // no more records than rays, and the 32-byte records of one call must stay addressable (ISX_MAX_LOG_RECORDS)
  if (capacity > n_rays && n_rays > 0) capacity = n_rays;
  if (capacity > ISX_MAX_LOG_RECORDS) return ISX_ERR_TOO_LARGE;
  int rc = ensure_hist(1);
  if (rc) return rc;
  rc = collect_stats(nullptr);
  if (rc) return rc;
  double* d_rec = nullptr;
  unsigned long long* d_cnt = nullptr;
  HIPCHK(hipMalloc(&d_rec, capacity * 32));
  hipError_t e = hipMalloc(&d_cnt, 8);
  if (e == hipSuccess) e = hipMemsetAsync(d_cnt, 0, 8, S.stream);
  if (e == hipSuccess) e = hipMemsetAsync(S.d_hist, 0, 8, S.stream);
  if (e != hipSuccess) { (void)hipFree(d_rec); if (d_cnt) (void)hipFree(d_cnt); S.last_hip = (int)e; return ISX_ERR_HIP; }
  LogSink lg;
  lg.rec = d_rec; lg.count = d_cnt; lg.cap = capacity;
  rc = enqueue(SINK_LOG, cfg, n_rays, seed, first_ray, S.d_hist, 0, nullptr, 0, 0, nullptr, &lg);
  unsigned long long total = 0;
  std::vector<double> rec;
  if (rc == ISX_OK) {
    e = hipStreamSynchronize(S.stream);
    if (e == hipSuccess) e = hipMemcpy(&total, d_cnt, 8, hipMemcpyDeviceToHost);
    const uint64_t kept = total < capacity ? total : capacity;
    rec.resize(kept * 4);
    if (e == hipSuccess && kept) e = hipMemcpy(rec.data(), d_rec, kept * 32, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { S.last_hip = (int)e; rc = ISX_ERR_HIP; }
    else {
      // slots are handed out in completion order: sort by ray index so the log is reproducible
      std::vector<size_t> order(kept);
      for (size_t k = 0; k < kept; ++k) order[k] = k;
      auto id_of = [&](size_t k) { uint64_t v; std::memcpy(&v, &rec[4 * k], 8); return v; };
      std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return id_of(a) < id_of(b); });
      for (size_t k = 0; k < kept; ++k) {
        const size_t o = order[k];
        ray_ids[k] = id_of(o);
        directions[3 * k] = rec[4 * (size_t)o + 1]; directions[3 * k + 1] = rec[4 * (size_t)o + 2]; directions[3 * k + 2] = rec[4 * (size_t)o + 3];
      }
    }
  }
  *count = total;
  const int rc2 = collect_stats(stats);
  (void)hipFree(d_rec); (void)hipFree(d_cnt);
  return rc ? rc : rc2;
}

int isx_fluxmap_series(const isx_config* cfgs, int32_t n_cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray,
                       uint64_t* hits, isx_stats* stats) {
  if (!S.init) return ISX_ERR_NOT_INIT;
  if (!cfgs || !hits || n_cfg < 1 || n_cfg > 4096) return ISX_ERR_BAD_ARG;
  int rc = check_grid(&cfgs[0]);
  if (rc) return rc;
  for (int k = 1; k < n_cfg; ++k)
    if (!same_grid(cfgs[0], cfgs[k]) || cfgs[k].det_diameter != cfgs[0].det_diameter) return ISX_ERR_BAD_CONFIG;
  const size_t nb = (size_t)cfgs[0].n_theta * cfgs[0].n_phi;
  rc = ensure_hist(nb * (size_t)n_cfg);
  if (rc) return rc;
  rc = collect_stats(nullptr);
  if (rc) return rc;
  unsigned long long* d_st = nullptr;
  HIPCHK(hipMalloc(&d_st, (size_t)n_cfg * 8 * sizeof(unsigned long long)));
  hipError_t e = hipMemsetAsync(d_st, 0, (size_t)n_cfg * 64, S.stream);
  if (e == hipSuccess) e = hipMemsetAsync(S.d_hist, 0, nb * (size_t)n_cfg * sizeof(unsigned long long), S.stream);
  if (e != hipSuccess) { (void)hipFree(d_st); S.last_hip = (int)e; return ISX_ERR_HIP; }
  // every configuration is enqueued back to back (no host round trip in between); configuration k
  // uses the ray indices [first_ray + k*n_rays, +n_rays) so the maps are statistically independent
  for (int k = 0; k < n_cfg && rc == ISX_OK; ++k)
    rc = enqueue(SINK_FLUX, &cfgs[k], n_rays, seed, first_ray + (uint64_t)k * n_rays, S.d_hist + (size_t)k * nb, 0, nullptr,
                 0, 0, nullptr, nullptr, d_st + (size_t)k * 8);
  std::vector<unsigned long long> hst((size_t)n_cfg * 8);
  if (rc == ISX_OK) {
    e = hipMemcpyAsync(hits, S.d_hist, nb * (size_t)n_cfg * sizeof(unsigned long long), hipMemcpyDeviceToHost, S.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hst.data(), d_st, hst.size() * 8, hipMemcpyDeviceToHost, S.stream);
    if (e != hipSuccess) { S.last_hip = (int)e; rc = ISX_ERR_HIP; }
  }
  isx_stats tot;
  const int rc2 = collect_stats(&tot);   // syncs; tot.t_kernel_ms = all launches
  (void)hipFree(d_st);
  if (rc == ISX_OK && rc2 == ISX_OK && stats) {
    for (int k = 0; k < n_cfg; ++k) {
      const unsigned long long* h = &hst[(size_t)k * 8];
      stats[k].launched = h[0]; stats[k].exited = h[1]; stats[k].counted_below_z = h[2]; stats[k].absorbed = h[3];
      stats[k].suspended = h[4]; stats[k].bin_increments = h[5]; stats[k].wall_hits = h[6];
      stats[k].t_kernel_ms = tot.t_kernel_ms;  // total of the series (launches are not timed separately)
    }
  }
  return rc ? rc : rc2;
}

int isx_exit_dz_hist(const isx_config* cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray, int32_t nbins,
                     uint64_t* hist, isx_stats* stats) {
  if (!S.init) return ISX_ERR_NOT_INIT;
  if (!cfg || !hist || nbins < 1) return ISX_ERR_BAD_ARG;
  int rc = ensure_hist((size_t)nbins);
  if (rc) return rc;
  rc = collect_stats(nullptr);
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(S.d_hist, 0, (size_t)nbins * sizeof(unsigned long long), S.stream));
  rc = enqueue(SINK_DZ, cfg, n_rays, seed, first_ray, S.d_hist, nbins, nullptr, 0, 0);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(hist, S.d_hist, (size_t)nbins * sizeof(unsigned long long), hipMemcpyDeviceToHost, S.stream));
  return collect_stats(stats);
}

#ifdef ISX_DIAG
// tuning builds only (not declared in isx.h): read and clear the binning diagnostics of isx_kernels.hpp
int isx_diag_read(uint64_t* out48) {
  if (!S.init || !out48) return ISX_ERR_BAD_ARG;
  HIPCHK(hipStreamSynchronize(S.stream));
  HIPCHK(hipMemcpyFromSymbol(out48, HIP_SYMBOL(isx::g_diag), 48 * sizeof(unsigned long long)));
  unsigned long long z[48] = {0};
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(isx::g_diag), z, sizeof(z)));
  return ISX_OK;
}
#endif

}  // extern "C"
