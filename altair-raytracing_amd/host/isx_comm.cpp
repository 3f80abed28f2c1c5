// isx_comm.cpp — see isx_comm.hpp.  RCCL (librccl, the ROCm build of NCCL) + a file rendezvous for the unique id.
#include "isx_comm.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <thread>
#include <vector>

#include "isx_macros.hpp"

namespace isxhost {

namespace {

int env_int(std::initializer_list<const char*> names, int dflt) {
  for (const char* n : names)
    if (const char* s = std::getenv(n)) return std::atoi(s);
  return dflt;
}

struct Rccl {
  bool ready = false, failed = false;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  unsigned long long* d_buf = nullptr;
  size_t cap = 0;
} R;

std::string rendezvous_path() {
  if (const char* s = std::getenv("ISX_RENDEZVOUS")) return s;
  std::string tag = "default";
  if (const char* s = std::getenv("MASTER_PORT")) tag = s;
  return "/tmp/isx_rccl_" + tag + "_" + std::to_string((long)getuid());
}

#define ISX_HIP_OK(call)                                                                                        \
  do {                                                                                                          \
    const hipError_t e_ = (call);                                                                               \
    if (e_ != hipSuccess) {                                                                                     \
      std::cerr << "Error: isx_comm: " #call ": " << hipGetErrorString(e_) << std::endl;                       \
      return false;                                                                                             \
    }                                                                                                           \
  } while (0)
#define ISX_NCCL_OK(call)                                                                                       \
  do {                                                                                                          \
    const ncclResult_t r_ = (call);                                                                             \
    if (r_ != ncclSuccess) {                                                                                    \
      std::cerr << "Error: isx_comm: " #call ": " << ncclGetErrorString(r_) << std::endl;                      \
      return false;                                                                                             \
    }                                                                                                           \
  } while (0)

bool rccl_init(const Comm& c) {
  if (R.ready) return true;
  if (R.failed) return false;
  R.failed = true;  // until proven otherwise
  if (!ensure_device()) return false;   // libisx has bound this process to its GPU (same HIP runtime)
  ncclUniqueId id;
  const std::string path = rendezvous_path();
  const std::time_t started = std::time(nullptr);
  if (c.rank == 0) {
    ISX_NCCL_OK(ncclGetUniqueId(&id));
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f || std::fwrite(&id, 1, sizeof(id), f) != sizeof(id)) {
      std::cerr << "Error: isx_comm: cannot write rendezvous file " << tmp << std::endl;
      if (f) std::fclose(f);
      return false;
    }
    std::fclose(f);
    if (std::rename(tmp.c_str(), path.c_str()) != 0) {
      std::cerr << "Error: isx_comm: cannot publish rendezvous file " << path << std::endl;
      return false;
    }
  } else {
    bool got = false;
    for (int tries = 0; tries < 2400 && !got; ++tries) {   // 120 s
      struct stat sb;
      // a file left behind by an earlier job is older than this process by more than the launch skew we tolerate
      if (stat(path.c_str(), &sb) == 0 && sb.st_size == (off_t)sizeof(id) && sb.st_mtime + 120 >= started) {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (f) {
          got = std::fread(&id, 1, sizeof(id), f) == sizeof(id);
          std::fclose(f);
        }
      }
      if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
    if (!got) {
      std::cerr << "Error: isx_comm: rank " << c.rank << " found no rendezvous file " << path << " within 120 s" << std::endl;
      return false;
    }
  }
  ISX_NCCL_OK(ncclCommInitRank(&R.comm, c.world, id, c.rank));   // returns once every rank has joined
  if (c.rank == 0) std::remove(path.c_str());
  ISX_HIP_OK(hipStreamCreateWithFlags(&R.stream, hipStreamNonBlocking));
  R.failed = false;
  R.ready = true;
  return true;
}

}  // namespace

Comm& comm() {
  static Comm c = [] {
    Comm k;
    k.rank = env_int({"ISX_RANK", "RANK", "OMPI_COMM_WORLD_RANK"}, 0);
    k.world = env_int({"ISX_WORLD", "WORLD_SIZE", "OMPI_COMM_WORLD_SIZE"}, 1);
    k.local_rank = env_int({"ISX_LOCAL_RANK", "LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK"}, k.rank);
    if (k.world < 1 || k.rank < 0 || k.rank >= k.world) {
      std::cerr << "Error: isx_comm: bad rank/world (" << k.rank << "/" << k.world << "), running as a single rank" << std::endl;
      k.rank = 0; k.world = 1; k.local_rank = 0;
    }
    if (const char* s = std::getenv("ISX_FORCE_COMM")) k.forced = std::atoi(s) != 0;
    if (k.world > 1 && !std::getenv("ISX_DEVICE")) options().device = k.local_rank;
    if (k.world > 1 && k.rank != 0) options().quiet = true;
    return k;
  }();
  return c;
}

void Comm::shard(uint64_t n, uint64_t& first, uint64_t& count) const {
  const uint64_t q = n / (uint64_t)world, r = n % (uint64_t)world, k = (uint64_t)rank;
  first = k * q + (k < r ? k : r);
  count = q + (k < r ? 1 : 0);
}

bool Comm::reduce(uint64_t* hits, size_t count, isx_stats* st, int n_stats) {
  if (!active()) return true;
  if (!rccl_init(*this)) return false;
  const size_t ns = st ? (size_t)n_stats : 0;
  const size_t words = count + 7 * ns, total = words + ns;   // [hits | 7 census words per stats | kernel microseconds per stats]
  if (total > R.cap) {
    if (R.d_buf) ISX_HIP_OK(hipFree(R.d_buf));
    R.d_buf = nullptr; R.cap = 0;
    ISX_HIP_OK(hipMalloc(&R.d_buf, total * sizeof(unsigned long long)));
    R.cap = total;
  }
  std::vector<unsigned long long> h(total);
  std::memcpy(h.data(), hits, count * sizeof(uint64_t));
  for (size_t k = 0; k < ns; ++k) {
    unsigned long long* c = h.data() + count + 7 * k;
    c[0] = st[k].launched; c[1] = st[k].exited; c[2] = st[k].counted_below_z; c[3] = st[k].absorbed;
    c[4] = st[k].suspended; c[5] = st[k].bin_increments; c[6] = st[k].wall_hits;
    h[words + k] = (unsigned long long)(st[k].t_kernel_ms * 1e3 + 0.5);
  }
  ISX_HIP_OK(hipMemcpyAsync(R.d_buf, h.data(), total * sizeof(unsigned long long), hipMemcpyHostToDevice, R.stream));
  ISX_NCCL_OK(ncclAllReduce(R.d_buf, R.d_buf, words, ncclUint64, ncclSum, R.comm, R.stream));
  if (ns) ISX_NCCL_OK(ncclAllReduce(R.d_buf + words, R.d_buf + words, ns, ncclUint64, ncclMax, R.comm, R.stream));
  ISX_HIP_OK(hipMemcpyAsync(h.data(), R.d_buf, total * sizeof(unsigned long long), hipMemcpyDeviceToHost, R.stream));
  ISX_HIP_OK(hipStreamSynchronize(R.stream));
  std::memcpy(hits, h.data(), count * sizeof(uint64_t));
  for (size_t k = 0; k < ns; ++k) {
    const unsigned long long* c = h.data() + count + 7 * k;
    st[k].launched = c[0]; st[k].exited = c[1]; st[k].counted_below_z = c[2]; st[k].absorbed = c[3];
    st[k].suspended = c[4]; st[k].bin_increments = c[5]; st[k].wall_hits = c[6];
    st[k].t_kernel_ms = (double)h[words + k] * 1e-3;
  }
  return true;
}

void Comm::finalize() {
  if (R.ready) {
    (void)hipStreamSynchronize(R.stream);
    (void)ncclCommDestroy(R.comm);
    (void)hipStreamDestroy(R.stream);
    if (R.d_buf) (void)hipFree(R.d_buf);
    R = Rccl();
  }
}

std::string outputPath(const std::string& base) { return comm().writer() ? getUniqueFilename(base) : std::string("/dev/null"); }

// ---------------------------------------------------------------------------------------------
int fluxmap_all(const isx_config* cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray, uint64_t* hits, isx_stats* st) {
  Comm& c = comm();
  if (!c.active()) return isx_fluxmap(cfg, n_rays, seed, first_ray, hits, st);
  uint64_t f, cnt;
  c.shard(n_rays, f, cnt);
  isx_stats local;
  const int rc = isx_fluxmap(cfg, cnt, seed, first_ray + f, hits, &local);
  if (rc != ISX_OK) return rc;
  if (!c.reduce(hits, (size_t)cfg->n_theta * cfg->n_phi, &local)) return ISX_ERR_HIP;
  if (st) *st = local;
  return ISX_OK;
}

int fluxmap_per_position_all(const isx_config* cfg, uint64_t rays_per_position, int32_t fold, uint64_t n_groups, uint64_t seed,
                             uint64_t first_ray, uint64_t* hits, isx_stats* st) {
  Comm& c = comm();
  if (!c.active()) return isx_fluxmap_per_position(cfg, rays_per_position, fold, 0, n_groups, seed, first_ray, hits, st);
  uint64_t g0, ng;
  c.shard(n_groups, g0, ng);   // whole detector groups per rank: group g keeps its rays [first_ray + g*rays_per_position, ...)
  isx_stats local;
  const int rc = isx_fluxmap_per_position(cfg, rays_per_position, fold, g0, ng, seed, first_ray, hits, &local);
  if (rc != ISX_OK) return rc;
  if (!c.reduce(hits, (size_t)cfg->n_theta * cfg->n_phi, &local)) return ISX_ERR_HIP;
  if (st) *st = local;
  return ISX_OK;
}

int fluxmap_series_all(const isx_config* cfgs, int32_t n_cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray, uint64_t* hits,
                       isx_stats* st) {
  Comm& c = comm();
  if (!c.active()) return isx_fluxmap_series(cfgs, n_cfg, n_rays, seed, first_ray, hits, st);
  if (n_cfg < 1) return ISX_ERR_BAD_ARG;
  const size_t nb = (size_t)cfgs[0].n_theta * cfgs[0].n_phi;
  uint64_t f, cnt;
  c.shard(n_rays, f, cnt);
  std::vector<isx_stats> local((size_t)n_cfg);
  for (int32_t k = 0; k < n_cfg; ++k) {   // configuration k owns the ray indices [first_ray + k*n_rays, +n_rays)
    const int rc = isx_fluxmap(&cfgs[k], cnt, seed, first_ray + (uint64_t)k * n_rays + f, hits + (size_t)k * nb, &local[(size_t)k]);
    if (rc != ISX_OK) return rc;
  }
  if (!c.reduce(hits, nb * (size_t)n_cfg, local.data(), n_cfg)) return ISX_ERR_HIP;
  if (st) for (int32_t k = 0; k < n_cfg; ++k) st[k] = local[(size_t)k];
  return ISX_OK;
}

int disc_sweep_all(const isx_config* cfg, const double* centers_axes, int32_t n_disc, double radius, double half_thick,
                   uint64_t n_rays, uint64_t seed, uint64_t first_ray, uint64_t* hits, isx_stats* st) {
  Comm& c = comm();
  if (!c.active()) return isx_disc_sweep(cfg, centers_axes, n_disc, radius, half_thick, n_rays, seed, first_ray, hits, st);
  uint64_t f, cnt;
  c.shard(n_rays, f, cnt);
  isx_stats local;
  const int rc = isx_disc_sweep(cfg, centers_axes, n_disc, radius, half_thick, cnt, seed, first_ray + f, hits, &local);
  if (rc != ISX_OK) return rc;
  if (!c.reduce(hits, (size_t)n_disc, &local)) return ISX_ERR_HIP;
  if (st) *st = local;
  return ISX_OK;
}

int exit_dz_hist_all(const isx_config* cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray, int32_t nbins, uint64_t* hist,
                     isx_stats* st) {
  Comm& c = comm();
  if (!c.active()) return isx_exit_dz_hist(cfg, n_rays, seed, first_ray, nbins, hist, st);
  uint64_t f, cnt;
  c.shard(n_rays, f, cnt);
  isx_stats local;
  const int rc = isx_exit_dz_hist(cfg, cnt, seed, first_ray + f, nbins, hist, &local);
  if (rc != ISX_OK) return rc;
  if (!c.reduce(hist, (size_t)nbins, &local)) return ISX_ERR_HIP;
  if (st) *st = local;
  return ISX_OK;
}

}  // namespace isxhost
