// isx_comm.cpp — see isx_comm.hpp.  RCCL (librccl, the ROCm build of NCCL) + a per-job rendezvous directory.
#include "isx_comm.hpp"

#include <dirent.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
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

// ------------------------------------------------------------------------------------------------ rendezvous
// One directory per job, private to the user: <ISX_RENDEZVOUS | $XDG_RUNTIME_DIR | $TMPDIR | /tmp>/isx_rdzv_<uid>_<job>.
// <job> is the per-launch nonce: ISX_JOB_ID, else torchrun's TORCHELASTIC_RUN_ID + MASTER_PORT.  A multi-rank launch
// without any of them is refused (two jobs would share one directory).  Files: `id` = magic | job tag | ncclUniqueId,
// created with O_EXCL, mode 0600, published by rename; `launch` = rank 0's nonce of this launch; `ready.<rank>` /
// `fail.<rank>` = the pre-flight of rccl_init(), each carrying that nonce (a file of an earlier launch is ignored).
constexpr char kMagic[8] = {'I', 'S', 'X', 'R', 'D', 'Z', 'V', '2'};
constexpr size_t kTagLen = 96;
constexpr int kWaitSecondsDefault = 120;
// how long a rank waits for the others in the rendezvous directory (ISX_RENDEZVOUS_WAIT, seconds: tests shorten it)
int wait_seconds() {
  if (const char* s = std::getenv("ISX_RENDEZVOUS_WAIT")) { const int v = std::atoi(s); if (v >= 1 && v <= 3600) return v; }
  return kWaitSecondsDefault;
}

bool job_tag(std::string& tag) {
  if (const char* s = std::getenv("ISX_JOB_ID")) { tag = s; return !tag.empty(); }
  std::string t;
  if (const char* s = std::getenv("TORCHELASTIC_RUN_ID")) t += s;
  if (const char* s = std::getenv("MASTER_PORT")) t += std::string("_") + s;
  tag = t;
  return !tag.empty();
}

std::string sanitize(std::string s) {
  for (char& ch : s)
    if (!((ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z') || (ch >= '0' && ch <= '9') || ch == '-' || ch == '.')) ch = '_';
  if (s.size() > 64) s.resize(64);
  return s;
}

bool rendezvous_dir(const Comm& c, std::string& dir, std::string& tag) {
  if (!job_tag(tag)) {
    if (c.world > 1) {
      std::cerr << "Error: isx_comm: a multi-rank launch needs a per-job nonce: set ISX_JOB_ID (or launch through torchrun, "
                   "which exports TORCHELASTIC_RUN_ID / MASTER_PORT)" << std::endl;
      return false;
    }
    tag = "single_" + std::to_string((long)getpid());
  }
  std::string base;
  if (const char* s = std::getenv("ISX_RENDEZVOUS")) base = s;
  else if (const char* s = std::getenv("XDG_RUNTIME_DIR")) base = s;
  else if (const char* s = std::getenv("TMPDIR")) base = s;
  else base = "/tmp";
  dir = base + "/isx_rdzv_" + std::to_string((long)getuid()) + "_" + sanitize(tag);
  if (mkdir(dir.c_str(), 0700) != 0 && errno != EEXIST) {
    std::cerr << "Error: isx_comm: cannot create rendezvous directory " << dir << std::endl;
    return false;
  }
  struct stat sb;
  if (lstat(dir.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode) || sb.st_uid != getuid() || (sb.st_mode & 077) != 0) {
    std::cerr << "Error: isx_comm: rendezvous directory " << dir << " is not a private directory of this user" << std::endl;
    return false;
  }
  return true;
}

// publish `bytes` as dir/name atomically (O_EXCL temp file, mode 0600, rename)
bool publish(const std::string& dir, const std::string& name, const void* bytes, size_t n) {
  const std::string tmp = dir + "/." + name + ".tmp." + std::to_string((long)getpid());
  (void)unlink(tmp.c_str());
  const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL, 0600);
  if (fd < 0) return false;
  const bool ok = n == 0 || write(fd, bytes, n) == (ssize_t)n;
  close(fd);
  if (!ok || std::rename(tmp.c_str(), (dir + "/" + name).c_str()) != 0) { (void)unlink(tmp.c_str()); return false; }
  return true;
}

// Files of the pre-flight carry the nonce of THIS launch (rank 0 draws it and publishes it as `launch`); a file of an
// earlier launch with the same job tag -- a `fail.<r>` that was never removed, a `ready.<r>` of a rank that died before
// ncclCommInitRank -- has another nonce and is ignored, whatever its age.
bool read_nonce(const std::string& path, unsigned long long* nonce) {
  struct stat sb;
  if (stat(path.c_str(), &sb) != 0 || sb.st_uid != getuid() || sb.st_size != (off_t)(sizeof(kMagic) + sizeof(*nonce))) return false;
  unsigned char blob[sizeof(kMagic) + sizeof(*nonce)];
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  const bool ok = std::fread(blob, 1, sizeof(blob), f) == sizeof(blob) && std::memcmp(blob, kMagic, sizeof(kMagic)) == 0;
  std::fclose(f);
  if (ok) std::memcpy(nonce, blob + sizeof(kMagic), sizeof(*nonce));
  return ok;
}
bool publish_nonce(const std::string& dir, const std::string& name, unsigned long long nonce) {
  unsigned char blob[sizeof(kMagic) + sizeof(nonce)];
  std::memcpy(blob, kMagic, sizeof(kMagic));
  std::memcpy(blob + sizeof(kMagic), &nonce, sizeof(nonce));
  (void)unlink((dir + "/" + name).c_str());
  return publish(dir, name, blob, sizeof(blob));
}
bool has_nonce(const std::string& path, unsigned long long nonce) {
  unsigned long long got = 0;
  return read_nonce(path, &got) && got == nonce;
}

std::string g_fail_file;   // this rank's fail.<rank>, removed when the process leaves (Comm::finalize)

struct Rccl {
  bool ready = false, failed = false;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  unsigned long long* d_buf = nullptr;
  size_t cap = 0;
} R;

bool rccl_init(const Comm& c) {
  if (R.ready) return true;
  if (R.failed) return false;
  R.failed = true;  // until proven otherwise
  std::string dir, tag;
  if (!rendezvous_dir(c, dir, tag)) return false;
  // ---- pre-flight: a rank whose GPU cannot be bound says so in the directory, so that nobody enters
  // ncclCommInitRank (which has no time-out) for a job that cannot start
  // (ISX_COMM_ASSUME_DEVICE=1, tests only: rehearse the pre-flight's file protocol on a box without a GPU)
  const bool dev_ok = ensure_device() || (std::getenv("ISX_COMM_ASSUME_DEVICE") && std::atoi(std::getenv("ISX_COMM_ASSUME_DEVICE")) == 1);
  const std::string me = std::to_string(c.rank);
  (void)unlink((dir + "/ready." + me).c_str());   // nothing of this rank survives from an earlier launch with the same tag
  (void)unlink((dir + "/fail." + me).c_str());
  (void)unlink((dir + "/ack." + me).c_str());
  auto fresh = []() {
    unsigned long long v = ((unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() << 20) ^
                           (unsigned long long)getpid() ^ ((unsigned long long)std::time(nullptr) << 40);
    return v ? v : 1ull;
  };
  // A killed launch with the same job tag (a fixed ISX_JOB_ID, a torchrun restart) leaves `launch`, `ready.*` and `id` behind, all
  // with ITS nonce: a rank of the relaunch that starts before the new rank 0 must not adopt them (it would read the stale id
  // and sit in ncclCommInitRank, which has no time-out).  Two measures: rank 0 removes what an earlier launch left before it
  // publishes anything (the followers' `hello.*` included: a follower that is already waiting says hello again), and a follower
  // trusts a `launch` file only once rank 0 has ECHOED the follower's own fresh token (`hello.<rank>` -> `ack.<rank>` =
  // mix(nonce, token)): no file of an earlier launch can hold that value.
  auto mix = [](unsigned long long nonce, unsigned long long token) {
    return nonce ^ (token * 0x9E3779B97F4A7C15ull) ^ ((token << 17) | (token >> 47));
  };
  const unsigned long long token = fresh();
  unsigned long long nonce = 0;
  std::vector<unsigned long long> acked((size_t)c.world, 0ull);
  if (c.rank == 0) {
    for (const char* name : {"launch", "id"}) (void)unlink((dir + "/" + name).c_str());
    for (int r = 0; r < c.world; ++r)
      for (const char* stem : {"ready.", "fail.", "ack.", "hello."}) (void)unlink((dir + "/" + stem + std::to_string(r)).c_str());
    nonce = fresh();
    if (!publish_nonce(dir, "launch", nonce)) {
      std::cerr << "Error: isx_comm: cannot write into " << dir << std::endl;
      return false;
    }
  } else if (!publish_nonce(dir, "hello." + me, token)) {
    std::cerr << "Error: isx_comm: cannot write into " << dir << std::endl;
    return false;
  }
  auto announce = [&](unsigned long long n) {
    const std::string name = (dev_ok ? "ready." : "fail.") + me;
    if (!dev_ok) g_fail_file = dir + "/" + name;
    return publish_nonce(dir, name, n);
  };
  if (c.rank == 0 && !announce(nonce)) {
    std::cerr << "Error: isx_comm: cannot write into " << dir << std::endl;
    return false;
  }
  bool all_ready = false, someone_failed = false;
  for (int tries = 0; tries < wait_seconds() * 20 && !all_ready && !someone_failed; ++tries) {
    if (c.rank == 0) {   // echo the followers' tokens (a stale hello.<r> gets a stale ack: harmless, the live rank re-publishes)
      int n_acked = 0;
      for (int r = 1; r < c.world; ++r) {
        unsigned long long t = 0;
        if (read_nonce(dir + "/hello." + std::to_string(r), &t) && t != acked[(size_t)r] &&
            publish_nonce(dir, "ack." + std::to_string(r), mix(nonce, t)))
          acked[(size_t)r] = t;
        if (acked[(size_t)r] != 0ull) n_acked++;
      }
      // a rank 0 that cannot bind its GPU has said so (fail.0); it stays long enough to echo the followers' tokens -- they can
      // then verify the nonce of that file and stop at once instead of waiting out the rendezvous -- but no longer than 10 s
      if (!dev_ok && (n_acked == c.world - 1 || tries >= 200)) { someone_failed = true; break; }
    }
    if (c.rank != 0) {   // follow rank 0's nonce -- once rank 0 of THIS launch has echoed this rank's token
      // (rank 0 clears the directory when it starts, possibly after this rank said hello: say it again)
      if (nonce == 0 && !has_nonce(dir + "/hello." + me, token)) (void)publish_nonce(dir, "hello." + me, token);
      unsigned long long seen = 0;
      if (read_nonce(dir + "/launch", &seen) && seen != nonce && has_nonce(dir + "/ack." + me, mix(seen, token))) {
        nonce = seen;
        if (!announce(nonce)) {
          std::cerr << "Error: isx_comm: cannot write into " << dir << std::endl;
          return false;
        }
      }
      if (nonce == 0) { std::this_thread::sleep_for(std::chrono::milliseconds(50)); continue; }
      if (!dev_ok) { someone_failed = true; break; }
    }
    int n_ready = 0;
    for (int r = 0; r < c.world; ++r) {
      if (r == 0 && c.rank == 0 && !dev_ok) continue;   // (its own failure: handled above, once the followers can verify it)
      if (has_nonce(dir + "/fail." + std::to_string(r), nonce)) someone_failed = true;
      else if (has_nonce(dir + "/ready." + std::to_string(r), nonce)) n_ready++;
    }
    all_ready = n_ready == c.world;
    if (!all_ready && !someone_failed) std::this_thread::sleep_for(std::chrono::milliseconds(50));
  }
  if (!all_ready) {
    if (c.rank != 0) { (void)unlink((dir + "/hello." + me).c_str()); (void)unlink((dir + "/ack." + me).c_str()); }
    std::cerr << "Error: isx_comm: rank " << c.rank << ": " << (someone_failed ? "a rank could not bind its GPU" : "not every rank showed up")
              << " (rendezvous " << dir << "); not starting the job" << std::endl;
    return false;
  }
  // ---- the ncclUniqueId
  ncclUniqueId id;
  unsigned char blob[sizeof(kMagic) + kTagLen + sizeof(nonce) + sizeof(id)];
  if (c.rank == 0) {
    ISX_NCCL_OK(ncclGetUniqueId(&id));
    std::memset(blob, 0, sizeof(blob));
    std::memcpy(blob, kMagic, sizeof(kMagic));
    std::strncpy((char*)blob + sizeof(kMagic), tag.c_str(), kTagLen - 1);
    std::memcpy(blob + sizeof(kMagic) + kTagLen, &nonce, sizeof(nonce));
    std::memcpy(blob + sizeof(kMagic) + kTagLen + sizeof(nonce), &id, sizeof(id));
    (void)unlink((dir + "/id").c_str());   // nothing of an earlier launch survives
    if (!publish(dir, "id", blob, sizeof(blob))) {
      std::cerr << "Error: isx_comm: cannot publish " << dir << "/id" << std::endl;
      return false;
    }
  } else {
    bool got = false;
    for (int tries = 0; tries < wait_seconds() * 20 && !got; ++tries) {
      struct stat sb;
      if (stat((dir + "/id").c_str(), &sb) == 0 && sb.st_uid == getuid() && sb.st_size == (off_t)sizeof(blob)) {
        FILE* f = std::fopen((dir + "/id").c_str(), "rb");
        if (f) {
          got = std::fread(blob, 1, sizeof(blob), f) == sizeof(blob) && std::memcmp(blob, kMagic, sizeof(kMagic)) == 0 &&
                std::strncmp((const char*)blob + sizeof(kMagic), tag.c_str(), kTagLen - 1) == 0 &&
                std::memcmp(blob + sizeof(kMagic) + kTagLen, &nonce, sizeof(nonce)) == 0;   // the id of THIS launch
          std::fclose(f);
        }
      }
      if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
    if (!got) {
      std::cerr << "Error: isx_comm: rank " << c.rank << " found no id of job '" << tag << "' in " << dir << " within " << wait_seconds()
                << " s" << std::endl;
      return false;
    }
    std::memcpy(&id, blob + sizeof(kMagic) + kTagLen + sizeof(nonce), sizeof(id));
  }
  ISX_NCCL_OK(ncclCommInitRank(&R.comm, c.world, id, c.rank));   // returns once every rank has joined
  (void)unlink((dir + "/ready." + me).c_str());
  (void)unlink((dir + "/hello." + me).c_str());
  (void)unlink((dir + "/ack." + me).c_str());
  if (c.rank == 0) { (void)unlink((dir + "/id").c_str()); (void)unlink((dir + "/launch").c_str()); (void)rmdir(dir.c_str()); }   // rmdir succeeds once the last rank has cleaned up
  else (void)rmdir(dir.c_str());
  ISX_HIP_OK(hipStreamCreateWithFlags(&R.stream, hipStreamNonBlocking));
  R.failed = false;
  R.ready = true;
  return true;
}

// RCCL transport: ONE in-place ncclAllReduce(ncclSum, ncclUint64) per call on one stream, one synchronisation ("a
// single all-reduce of the histogram").  Everything that is not a sum travels as per-rank slots of the SUM buffer (see
// reduce_collective): a slot is written by one rank and zero on all others, so its sum is that rank's value.
// A rank that cannot ENTER the collective (a local HIP failure between rccl_init() and ncclAllReduce: no memory for the buffer,
// a failed copy of its contribution) must not leave the others waiting in an all-reduce that has no time-out: it aborts the
// communicator, which makes the peers' ncclAllReduce / stream synchronisation return an error -- every rank then reports a failure
// (ADVICE r04).  After an abort this process has no communicator any more (R.failed: rccl_init() refuses to build a second one).
bool abort_comm(const char* why) {
  std::cerr << "Error: isx_comm: " << why << ": aborting the communicator so that no rank waits for this one" << std::endl;
  if (R.ready) {
    (void)ncclCommAbort(R.comm);
    R.ready = false;
    R.failed = true;
  }
  return false;
}
// The wait for the collective is a POLL, not a blocking synchronisation: ncclAllReduce has no time-out, and a peer that died or
// aborted its communicator (abort_comm) leaves this rank's all-reduce kernel waiting for data that will never come.  While the
// stream is busy the communicator's asynchronous error state is checked (RCCL reports a lost peer there when its transport
// notices), and a launch-wide time-out bounds what no transport notices (ISX_COLLECTIVE_TIMEOUT seconds, default 600: the ranks of a
// launch do equal work, they reach the collective within milliseconds of each other).  Either way this rank aborts its own
// communicator -- the only way to take a stuck collective kernel off the device -- and reports the failure.
int collective_timeout_seconds() {
  if (const char* s = std::getenv("ISX_COLLECTIVE_TIMEOUT")) { const int v = std::atoi(s); if (v >= 1 && v <= 86400) return v; }
  return 600;
}
bool wait_collective() {
  const auto t0 = std::chrono::steady_clock::now();
  const auto limit = std::chrono::seconds(collective_timeout_seconds());
  for (unsigned polls = 0;; ++polls) {
    const hipError_t q = hipStreamQuery(R.stream);
    if (q == hipSuccess) return true;
    if (q != hipErrorNotReady) { std::cerr << "Error: isx_comm: the collective's stream failed: " << hipGetErrorString(q) << std::endl; return false; }
    ncclResult_t ar = ncclSuccess;
    if (ncclCommGetAsyncError(R.comm, &ar) != ncclSuccess || (ar != ncclSuccess && ar != ncclInProgress))
      return abort_comm("a peer of the collective failed (asynchronous RCCL error)");
    if (std::chrono::steady_clock::now() - t0 > limit) return abort_comm("the collective did not complete within ISX_COLLECTIVE_TIMEOUT");
    if (polls < 2000) std::this_thread::yield();                       // (the 130 KB all-reduce takes tens of microseconds)
    else std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
}
struct RcclTransport : Transport {
  bool reserve(size_t total) {
    if (total > R.cap) {
      if (R.d_buf) ISX_HIP_OK(hipFree(R.d_buf));
      R.d_buf = nullptr; R.cap = 0;
      ISX_HIP_OK(hipMalloc(&R.d_buf, total * sizeof(unsigned long long)));
      R.cap = total;
    }
    return true;
  }
  bool exchange_sum(unsigned long long* buf, size_t n) override {
    if (!reserve(n)) return abort_comm("no device buffer for the collective");
    if (hipMemcpyAsync(R.d_buf, buf, n * sizeof(unsigned long long), hipMemcpyHostToDevice, R.stream) != hipSuccess)
      return abort_comm("the copy of this rank's contribution failed");
    ISX_NCCL_OK(ncclAllReduce(R.d_buf, R.d_buf, n, ncclUint64, ncclSum, R.comm, R.stream));
    ISX_HIP_OK(hipMemcpyAsync(buf, R.d_buf, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, R.stream));
    return wait_collective();
  }
  // the first n_dev words are already in R.d_buf (the histogram, written there by the kernels: no D2H -> H2D round trip of the
  // 130 KB before the collective); the tail comes from the host.  The whole reduced buffer goes back to `buf`.
  bool exchange_sum_device_head(unsigned long long* buf, size_t n_dev, size_t n) {
    if (n > R.cap) return abort_comm("the collective's device buffer is smaller than the call");   // (device_hist() reserved it)
    if (n > n_dev &&
        hipMemcpyAsync(R.d_buf + n_dev, buf + n_dev, (n - n_dev) * sizeof(unsigned long long), hipMemcpyHostToDevice, R.stream) != hipSuccess)
      return abort_comm("the copy of this rank's census failed");
    ISX_NCCL_OK(ncclAllReduce(R.d_buf, R.d_buf, n, ncclUint64, ncclSum, R.comm, R.stream));
    ISX_HIP_OK(hipMemcpyAsync(buf, R.d_buf, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, R.stream));
    return wait_collective();
  }
} g_rccl;

Transport* g_transport_override = nullptr;

// words of the SUM buffer behind the histogram: census (7 per statistics block), kernel time (one slot per rank and block),
// status (one slot per rank)
size_t tail_words(size_t ns, int world) { return 7 * ns + ns * (size_t)world + (size_t)world; }

void pack_tail(unsigned long long* tail, int rank, int world, int local_rc, const isx_stats* st, size_t ns) {
  std::memset(tail, 0, tail_words(ns, world) * sizeof(unsigned long long));
  if (local_rc == ISX_OK) {
    for (size_t k = 0; k < ns; ++k) {
      unsigned long long* c = tail + 7 * k;
      c[0] = st[k].launched; c[1] = st[k].exited; c[2] = st[k].counted_below_z; c[3] = st[k].absorbed;
      c[4] = st[k].suspended; c[5] = st[k].bin_increments; c[6] = st[k].wall_hits;
      tail[7 * ns + k * (size_t)world + (size_t)rank] = (unsigned long long)(st[k].t_kernel_ms * 1e3 + 0.5);
    }
  }
  // status: ISX_OK = 0, errors are negative -> the slot carries -rc; the job-wide status is the worst (most negative) one
  tail[7 * ns + ns * (size_t)world + (size_t)rank] = (unsigned long long)(-(long long)local_rc);
}

int unpack_tail(const unsigned long long* tail, int world, isx_stats* st, size_t ns) {
  unsigned long long worst = 0;
  for (int r = 0; r < world; ++r) worst = std::max(worst, tail[7 * ns + ns * (size_t)world + (size_t)r]);
  if (worst != 0) return -(int)worst;
  for (size_t k = 0; k < ns; ++k) {
    const unsigned long long* c = tail + 7 * k;
    st[k].launched = c[0]; st[k].exited = c[1]; st[k].counted_below_z = c[2]; st[k].absorbed = c[3];
    st[k].suspended = c[4]; st[k].bin_increments = c[5]; st[k].wall_hits = c[6];
    unsigned long long t = 0;
    for (int r = 0; r < world; ++r) t = std::max(t, tail[7 * ns + k * (size_t)world + (size_t)r]);
    st[k].t_kernel_ms = (double)t * 1e-3;
  }
  return ISX_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ the collective
// Every rank calls this exactly once per trace call, WHATEVER its local status: the status travels with the data, so either
// every rank gets the reduced result or every rank gets the same error -- no rank is left waiting in a collective the others
// never enter.  ONE sum: buffer = [hits | census words | kernel time: one slot per rank | status: one slot per rank]; a rank
// that failed contributes zeros and its error code.
int reduce_collective(Transport& t, int rank, int world, int local_rc, uint64_t* hits, size_t count, isx_stats* st, int n_stats) {
  const size_t ns = st ? (size_t)n_stats : 0;
  if (world < 1 || rank < 0 || rank >= world) return ISX_ERR_BAD_ARG;
  std::vector<unsigned long long> h(count + tail_words(ns, world), 0ull);
  if (local_rc == ISX_OK) std::memcpy(h.data(), hits, count * sizeof(uint64_t));
  pack_tail(h.data() + count, rank, world, local_rc, st, ns);
  if (!t.exchange_sum(h.data(), h.size())) return local_rc != ISX_OK ? local_rc : ISX_ERR_HIP;
  const int rc = unpack_tail(h.data() + count, world, st, ns);
  if (rc != ISX_OK) return rc;
  std::memcpy(hits, h.data(), count * sizeof(uint64_t));
  return ISX_OK;
}

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

void Comm::set_transport_for_tests(Transport* t) { g_transport_override = t; }

int Comm::reduce(int local_rc, uint64_t* hits, size_t count, isx_stats* st, int n_stats) {
  if (!active()) return local_rc;
  if (g_transport_override) return reduce_collective(*g_transport_override, rank, world, local_rc, hits, count, st, n_stats);
  if (!rccl_init(*this)) return local_rc != ISX_OK ? local_rc : ISX_ERR_HIP;
  return reduce_collective(g_rccl, rank, world, local_rc, hits, count, st, n_stats);
}

// The histogram of a sharded flux-map call can stay on the device until the collective: device_hist() hands out (and zeroes) the
// head of the collective's own buffer, the kernels accumulate into it (isx_fluxmap_device), reduce_device_hist() appends census,
// time and status and runs the ONE all-reduce in place.  nullptr: no RCCL on this rank (a transport double is installed, or the
// communicator could not be built) -- the caller takes the host path, whose collective then reports the failure.
unsigned long long* Comm::device_hist(size_t count, int n_stats) {
  if (!active() || g_transport_override) return nullptr;
  if (!rccl_init(*this)) return nullptr;
  if (!g_rccl.reserve(count + tail_words((size_t)n_stats, world))) return nullptr;
  if (hipMemsetAsync(R.d_buf, 0, count * sizeof(unsigned long long), R.stream) != hipSuccess) return nullptr;
  if (hipStreamSynchronize(R.stream) != hipSuccess) return nullptr;   // (the kernels run on libisx's own stream)
  return R.d_buf;
}

int Comm::reduce_device_hist(int local_rc, uint64_t* hits, size_t count, isx_stats* st, int n_stats) {
  const size_t ns = st ? (size_t)n_stats : 0;
  std::vector<unsigned long long> h(count + tail_words(ns, world), 0ull);
  pack_tail(h.data() + count, rank, world, local_rc, st, ns);
  if (local_rc != ISX_OK && hipMemsetAsync(R.d_buf, 0, count * sizeof(unsigned long long), R.stream) != hipSuccess) {
    (void)abort_comm("a failed rank could not zero its contribution");   // (a failed rank contributes zeros -- or nobody waits for it)
    return local_rc;
  }
  if (!g_rccl.exchange_sum_device_head(h.data(), count, h.size())) return local_rc != ISX_OK ? local_rc : ISX_ERR_HIP;
  const int rc = unpack_tail(h.data() + count, world, st, ns);
  if (rc != ISX_OK) return rc;
  std::memcpy(hits, h.data(), count * sizeof(uint64_t));
  return ISX_OK;
}

bool Comm::agree(bool local_ok) {
  if (!active()) return local_ok;
  uint64_t none = 0;
  return reduce(local_ok ? ISX_OK : ISX_ERR_BAD_ARG, &none, 0, nullptr, 0) == ISX_OK;
}

void Comm::finalize() {
  if (!g_fail_file.empty()) { (void)unlink(g_fail_file.c_str()); g_fail_file.clear(); }   // (a relaunch with the same tag starts clean)
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
// Sharded equivalents of the ABI calls: this rank's share, then ONE collective that also carries the status.
//
// The collective's element count comes from arguments every rank shares, so it is the same on all of them -- but it must
// be a count the library accepts: a configuration the ABI refuses (n_theta * n_phi outside 1..36000, more than 36000 discs
// or histogram bins) leaves only the status word to exchange, instead of a buffer sized by unvalidated numbers.
namespace {
constexpr long long kMaxBins = 36000;   // isx.h: the LDS histogram limit of every sink
size_t grid_bins(const isx_config* cfg) {
  if (!cfg || cfg->n_theta < 1 || cfg->n_phi < 1) return 0;
  const long long nb = (long long)cfg->n_theta * (long long)cfg->n_phi;
  return nb <= kMaxBins ? (size_t)nb : 0;
}
size_t list_bins(long long n) { return n >= 1 && n <= kMaxBins ? (size_t)n : 0; }
}  // namespace
int fluxmap_all(const isx_config* cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray, uint64_t* hits, isx_stats* st) {
  Comm& c = comm();
  if (!c.active()) return isx_fluxmap(cfg, n_rays, seed, first_ray, hits, st);
  uint64_t f, cnt;
  c.shard(n_rays, f, cnt);
  isx_stats local{};
  int rc;
  if (unsigned long long* d_hist = grid_bins(cfg) ? c.device_hist(grid_bins(cfg), 1) : nullptr) {
    // the histogram stays on the device: kernels -> the collective's buffer -> ONE all-reduce -> host
    rc = isx_take_stats(nullptr);                                     // (census of anything enqueued earlier must not leak into this call)
    if (rc == ISX_OK) rc = isx_fluxmap_device(cfg, cnt, seed, first_ray + f, (uint64_t*)d_hist);
    const int rc2 = isx_take_stats(&local);                           // synchronises libisx's stream
    if (rc == ISX_OK) rc = rc2;
    rc = c.reduce_device_hist(rc, hits, grid_bins(cfg), &local, 1);
  } else {
    rc = isx_fluxmap(cfg, cnt, seed, first_ray + f, hits, &local);
    rc = c.reduce(rc, hits, grid_bins(cfg), &local);
  }
  if (rc == ISX_OK && st) *st = local;
  return rc;
}

int fluxmap_per_position_range_all(const isx_config* cfg, uint64_t rays_per_position, int32_t fold, uint64_t first_group,
                                   uint64_t n_groups, uint64_t seed, uint64_t first_ray, uint64_t* hits, isx_stats* st) {
  Comm& c = comm();
  if (!c.active()) return isx_fluxmap_per_position(cfg, rays_per_position, fold, first_group, n_groups, seed, first_ray, hits, st);
  uint64_t g0, ng;
  c.shard(n_groups, g0, ng);   // whole detector groups per rank: group g keeps its rays [first_ray + g*rays_per_position, ...)
  isx_stats local{};
  int rc = isx_fluxmap_per_position(cfg, rays_per_position, fold, first_group + g0, ng, seed, first_ray, hits, &local);
  rc = c.reduce(rc, hits, grid_bins(cfg), &local);
  if (rc == ISX_OK && st) *st = local;
  return rc;
}

int fluxmap_per_position_all(const isx_config* cfg, uint64_t rays_per_position, int32_t fold, uint64_t n_groups, uint64_t seed,
                             uint64_t first_ray, uint64_t* hits, isx_stats* st) {
  return fluxmap_per_position_range_all(cfg, rays_per_position, fold, 0, n_groups, seed, first_ray, hits, st);
}

int fluxmap_series_all(const isx_config* cfgs, int32_t n_cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray, uint64_t* hits,
                       isx_stats* st) {
  Comm& c = comm();
  if (!c.active()) return isx_fluxmap_series(cfgs, n_cfg, n_rays, seed, first_ray, hits, st);
  if (n_cfg < 1) return ISX_ERR_BAD_ARG;   // the same on every rank: no collective needed
  const size_t nb = (size_t)cfgs[0].n_theta * cfgs[0].n_phi;
  uint64_t f, cnt;
  c.shard(n_rays, f, cnt);
  std::vector<isx_stats> local((size_t)n_cfg);
  int rc = ISX_OK;
  for (int32_t k = 0; k < n_cfg && rc == ISX_OK; ++k)   // configuration k owns the ray indices [first_ray + k*n_rays, +n_rays)
    rc = isx_fluxmap(&cfgs[k], cnt, seed, first_ray + (uint64_t)k * n_rays + f, hits + (size_t)k * nb, &local[(size_t)k]);
  rc = c.reduce(rc, hits, nb * (size_t)n_cfg, local.data(), n_cfg);
  if (rc == ISX_OK && st) for (int32_t k = 0; k < n_cfg; ++k) st[k] = local[(size_t)k];
  return rc;
}

int disc_sweep_all(const isx_config* cfg, const double* centers_axes, int32_t n_disc, double radius, double half_thick,
                   uint64_t n_rays, uint64_t seed, uint64_t first_ray, uint64_t* hits, isx_stats* st) {
  Comm& c = comm();
  if (!c.active()) return isx_disc_sweep(cfg, centers_axes, n_disc, radius, half_thick, n_rays, seed, first_ray, hits, st);
  uint64_t f, cnt;
  c.shard(n_rays, f, cnt);
  isx_stats local{};
  int rc = isx_disc_sweep(cfg, centers_axes, n_disc, radius, half_thick, cnt, seed, first_ray + f, hits, &local);
  rc = c.reduce(rc, hits, list_bins(n_disc), &local);
  if (rc == ISX_OK && st) *st = local;
  return rc;
}

int disc_sweep_per_position_all(const isx_config* cfg, const double* centers_axes, int32_t n_disc, double radius,
                                double half_thick, uint64_t rays_per_position, uint64_t seed, uint64_t first_ray, uint64_t* hits,
                                isx_stats* st) {
  Comm& c = comm();
  if (!c.active())
    return isx_disc_sweep_per_position(cfg, centers_axes, n_disc, radius, half_thick, rays_per_position, seed, first_ray, hits, st);
  // whole disc positions per rank; position k keeps its rays [first_ray + k*rays_per_position, ...) whatever the rank count
  uint64_t k0, nk;
  c.shard(n_disc > 0 ? (uint64_t)n_disc : 0, k0, nk);
  isx_stats local{};
  int rc = ISX_OK;
  if (list_bins(n_disc)) std::memset(hits, 0, list_bins(n_disc) * sizeof(uint64_t));
  if (nk > 0)
    rc = isx_disc_sweep_per_position(cfg, centers_axes + 6 * k0, (int32_t)nk, radius, half_thick, rays_per_position, seed,
                                     first_ray + k0 * rays_per_position, hits + k0, &local);
  rc = c.reduce(rc, hits, list_bins(n_disc), &local);
  if (rc == ISX_OK && st) *st = local;
  return rc;
}

int exit_dz_hist_all(const isx_config* cfg, uint64_t n_rays, uint64_t seed, uint64_t first_ray, int32_t nbins, uint64_t* hist,
                     isx_stats* st) {
  Comm& c = comm();
  if (!c.active()) return isx_exit_dz_hist(cfg, n_rays, seed, first_ray, nbins, hist, st);
  uint64_t f, cnt;
  c.shard(n_rays, f, cnt);
  isx_stats local{};
  int rc = isx_exit_dz_hist(cfg, cnt, seed, first_ray + f, nbins, hist, &local);
  rc = c.reduce(rc, hist, list_bins(nbins), &local);
  if (rc == ISX_OK && st) *st = local;
  return rc;
}

}  // namespace isxhost
