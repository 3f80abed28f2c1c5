// isx_comm.hpp — multi-GPU for the host driver: one PROCESS per GPU, rays (or per-position groups) sharded over the
// ranks, ONE RCCL all-reduce (ncclSum over xGMI) of the hit histogram + census at the end of every trace call
// (SURVEY.md §8e).  The reference has nothing like it (one process, <=4 threads sharing gRandom); results do not
// depend on the number of ranks because ray i always draws from Philox stream (seed, i).
//
// Launch (any launcher that exports a rank and a world size works; the names tried are, in order,
// ISX_RANK/ISX_WORLD/ISX_LOCAL_RANK, RANK/WORLD_SIZE/LOCAL_RANK (torchrun), OMPI_COMM_WORLD_RANK/_SIZE/_LOCAL_RANK):
//     for r in 0..7:  ISX_RANK=$r ISX_WORLD=8 ISX_RENDEZVOUS=/tmp/isx_job42 isx_macro fluxAtObserverFast::sweepDetectorTraceOnce ...
// Rank r binds GPU ISX_LOCAL_RANK (unless ISX_DEVICE says otherwise); rank 0 writes the files, the other ranks write
// to /dev/null.  The 128-byte ncclUniqueId travels through a private per-job directory
// <ISX_RENDEZVOUS | $XDG_RUNTIME_DIR | $TMPDIR | /tmp>/isx_rdzv_<uid>_<job>, <job> = ISX_JOB_ID or torchrun's
// TORCHELASTIC_RUN_ID + MASTER_PORT (a multi-rank launch without a per-job nonce is refused): the id file carries a
// magic word and the job tag and is verified on read; before anybody enters ncclCommInitRank every rank reports whether
// it could bind its GPU (ready.<rank> / fail.<rank>), so one broken rank stops the job instead of hanging it.
//
// Failure semantics: every *_all() call ends in ONE collective (one ncclAllReduce) that also carries the local status, so
// either every rank returns the reduced result or every rank returns the same error; Comm::agree() is the same for
// conditions checked on one rank only (the output file of the writer rank).
#pragma once
#include <cstdint>
#include <string>

#include "../../include/isx.h"

namespace isxhost {

// What the collective needs from the wire: ONE in-place sum of n uint64 words over all ranks.  RCCL in production
// (ncclAllReduce, ncclSum); tests install an in-process double (tests/native/comm_stub_test.cpp).
struct Transport {
  virtual bool exchange_sum(unsigned long long* buf, size_t n) = 0;
  virtual ~Transport() {}
};
// The collective itself (transport-independent), ONE sum per call: hits[count] and the census words of st[n_stats] are summed;
// the kernel times and the status travel as per-rank slots of the same buffer (written by one rank, zero on the others), from
// which every rank takes the MAX time and the worst status.  Returns the job-wide status; outputs are written only if it is ISX_OK.
int reduce_collective(Transport& t, int rank, int world, int local_rc, uint64_t* hits, size_t count, isx_stats* st, int n_stats);

struct Comm {
  int rank = 0, world = 1, local_rank = 0;
  bool forced = false;             // ISX_FORCE_COMM=1: go through RCCL even with one rank (rehearsal on a 1-GPU box)
  bool active() const { return world > 1 || forced; }
  bool writer() const { return rank == 0; }
  // contiguous share of [0,n): the first n % world ranks get one extra unit (same rule as sharding.py)
  void shard(uint64_t n, uint64_t& first, uint64_t& count) const;
  // called by EVERY rank with its local status: in-place SUM of hits[count] and of the census in *st over all ranks
  // (t_kernel_ms: MAX); returns the job-wide status (the worst local one)
  int reduce(int local_rc, uint64_t* hits, size_t count, isx_stats* st, int n_stats = 1);
  // the same with the histogram still on the device (RCCL only): device_hist() = zeroed device buffer of `count` words for the
  // kernels to accumulate into (nullptr: take the host path), reduce_device_hist() = the collective on it, result in hits[count]
  unsigned long long* device_hist(size_t count, int n_stats);
  int reduce_device_hist(int local_rc, uint64_t* hits, size_t count, isx_stats* st, int n_stats);
  // true on every rank iff local_ok on every rank (collective)
  bool agree(bool local_ok);
  void finalize();
  static void set_transport_for_tests(Transport* t);
};
Comm& comm();

// file the writer rank should create for `base` (never overwriting, getUniqueFilename); "/dev/null" on the other ranks
std::string outputPath(const std::string& base);

// Sharded equivalents of the ABI calls used by the sweeps: this rank's share, then Comm::reduce.  With one rank they
// are the plain calls.
int fluxmap_all(const isx_config* cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray, uint64_t* hits, isx_stats* st);
int fluxmap_per_position_all(const isx_config* cfg, uint64_t rays_per_position, int32_t fold, uint64_t n_groups, uint64_t seed,
                             uint64_t first_ray, uint64_t* hits, isx_stats* st);
// the same for the groups [first_group, first_group + n_groups) only (the other bins of `hits` come back 0): one batch of a
// sweep that writes its rows as it goes (fluxAtObserverOptimize.C:575-579)
int fluxmap_per_position_range_all(const isx_config* cfg, uint64_t rays_per_position, int32_t fold, uint64_t first_group,
                                   uint64_t n_groups, uint64_t seed, uint64_t first_ray, uint64_t* hits, isx_stats* st);
int fluxmap_series_all(const isx_config* cfgs, int32_t n_cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray, uint64_t* hits,
                       isx_stats* st);
int disc_sweep_all(const isx_config* cfg, const double* centers_axes, int32_t n_disc, double radius, double half_thick,
                   uint64_t n_rays, uint64_t seed, uint64_t first_ray, uint64_t* hits, isx_stats* st);
int disc_sweep_per_position_all(const isx_config* cfg, const double* centers_axes, int32_t n_disc, double radius,
                                double half_thick, uint64_t rays_per_position, uint64_t seed, uint64_t first_ray, uint64_t* hits,
                                isx_stats* st);
int exit_dz_hist_all(const isx_config* cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray, int32_t nbins, uint64_t* hist,
                     isx_stats* st);

}  // namespace isxhost
