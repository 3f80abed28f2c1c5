// isx_macro — command-line front end: runs one of the reference's macro entry points.
//   isx_macro <file>::<function> [key=value ...]
// e.g. isx_macro fluxAtObserverFast::sweepDetectorTraceOnce folder=out srcZ=-75 dirY=0 thetaMax=170
// (what `root -l -b -q 'fluxAtObserverFast.C+' -e 'sweepDetectorTraceOnce(...)'` did in the reference).
// Environment: ISX_DEVICE, ISX_SEED, ISX_RAYS (override the macro's hard-coded ray count), ISX_QUIET, ISX_FLUSH_ROWS (theta rows
// per launch of the per-position sweep, written and flushed as they are produced).
#include <sys/stat.h>

#include <cctype>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <string>

#include "isx_comm.hpp"
#include "isx_macros.hpp"

using namespace isxhost;

static double num(const std::map<std::string, std::string>& kv, const char* k, double dflt) {
  auto it = kv.find(k);
  return it == kv.end() ? dflt : std::atof(it->second.c_str());
}

// writer self-test: no GPU needed; a synthetic hit map goes through the same header/row/footer code
static int selftest_writer(const std::string& path) {
  FluxMapMeta mm;
  mm.title = "Flux Map Data"; mm.n_label = "Number of rays per position"; mm.n = 50000;
  std::vector<uint64_t> hits(180 * 90);
  for (size_t k = 0; k < hits.size(); ++k) hits[k] = (k * 7919u) % 1000u;
  FILE* f = std::fopen(path.c_str(), "w");
  if (!f) return 2;
  const std::string h = fluxmap_header(mm, "2025-04-01 01:42:14"), r = fluxmap_rows(hits.data(), mm.n, 180, 90);
  std::fwrite(h.data(), 1, h.size(), f);
  std::fwrite(r.data(), 1, r.size(), f);
  std::fclose(f);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    std::cerr << "usage: isx_macro <file>::<function> [key=value ...]\n"
                 "  fluxAtObserver::sweepDetector | fluxAtObserverOptimize::sweepDetector | fluxAtObserverOptimize::sweepSeries |\n"
                 "  fluxAtObserverFast::sweepDetectorTwofold | fluxAtObserverFast::sweepDetectorTraceOnce | fluxAtObserverFast::sweepSeries |\n"
                 "  nonLambertianFlux::sweepDetector | nonLambertianFluxCopy::sweepDetector | nonLambertianFluxCopy::visualizeDetectorText [theta= phi=] |\n"
                 "  makeIntegratingSphereNRays |\n"
                 "  integratingSphereDetectorSweep |\n"
                 "  distributionSphereDetectorSweep | --selftest-writer <file> | --unique <path> | --shard <n> | --analyze <csv>... | --analyze <folder> [average]\n";
    return 2;
  }
  const std::string entry = argv[1];
  if (entry == "--selftest-writer" && argc > 2) return selftest_writer(argv[2]);
  if (entry == "--shard" && argc > 2) {  // this rank's share of n units (rank/world from the environment, isx_comm.hpp)
    uint64_t f, c;
    comm().shard(std::strtoull(argv[2], nullptr, 10), f, c);
    std::cout << comm().rank << " " << comm().world << " " << f << " " << c << std::endl;
    return 0;
  }
  if (entry == "--unique" && argc > 2) { std::cout << getUniqueFilename(argv[2]) << std::endl; return 0; }
  if (entry == "--analyze" && argc > 2) {  // python flux_analysis.py <csv_file_or_folder> [average] (numbers; no GPU)
    struct stat sb;
    if (stat(argv[2], &sb) == 0 && S_ISDIR(sb.st_mode)) {
      std::string mode = argc > 3 ? argv[3] : "";
      for (char& ch : mode) ch = char(std::tolower(ch));
      std::vector<ThetaAnalysis> all;
      return analyzeFluxMapFolder(argv[2], mode == "average", all) ? 0 : 1;
    }
    int bad = 0;
    for (int i = 2; i < argc; ++i) { ThetaAnalysis ta; if (!analyzeFluxMap(argv[i], ta)) bad++; }
    return bad ? 1 : 0;
  }
  std::map<std::string, std::string> kv;
  for (int i = 2; i < argc; ++i) {
    const char* eq = std::strchr(argv[i], '=');
    if (!eq) { std::cerr << "bad argument " << argv[i] << " (want key=value)\n"; return 2; }
    kv[std::string(argv[i], eq - argv[i])] = eq + 1;
  }
  const std::string folder = kv.count("folder") ? kv["folder"] : "results";
  const bool notify = num(kv, "notify", 0) != 0;
  const double sx = num(kv, "srcX", -60), sy = num(kv, "srcY", 0), sz = num(kv, "srcZ", -80);
  const double dx = num(kv, "dirX", 5), dy = num(kv, "dirY", 2), dz = num(kv, "dirZ", 0), tm = num(kv, "thetaMax", 170);
  if (entry == "fluxAtObserver::sweepDetector") fluxAtObserver::sweepDetector();
  else if (entry == "fluxAtObserverOptimize::sweepDetector") fluxAtObserverOptimize::sweepDetector(notify, folder.c_str(), -1, sx, sy, sz, dx, dy, dz, tm);
  else if (entry == "fluxAtObserverOptimize::sweepSeries") fluxAtObserverOptimize::sweepSeries();
  else if (entry == "fluxAtObserverFast::sweepDetectorTwofold") fluxAtObserverFast::sweepDetectorTwofold(notify, folder.c_str(), -1, sx, sy, sz, dx, dy, dz, tm);
  else if (entry == "fluxAtObserverFast::sweepDetectorTraceOnce") fluxAtObserverFast::sweepDetectorTraceOnce(notify, folder.c_str(), -1, sx, sy, sz, dx, dy, dz, tm);
  else if (entry == "fluxAtObserverFast::sweepSeries") fluxAtObserverFast::sweepSeries();
  else if (entry == "nonLambertianFlux::sweepDetector") nonLambertianFlux::sweepDetector();
  else if (entry == "nonLambertianFluxCopy::sweepDetector") nonLambertianFluxCopy::sweepDetector();
  else if (entry == "nonLambertianFluxCopy::visualizeDetectorText") nonLambertianFluxCopy::visualizeDetectorText(num(kv, "theta", 45.0), num(kv, "phi", 0.0));
  else if (entry == "makeIntegratingSphereNRays") rootMacros::makeIntegratingSphereNRays();
  else if (entry == "integratingSphereDetectorSweep") rootMacros::integratingSphereDetectorSweep();
  else if (entry == "distributionSphereDetectorSweep") rootMacros::distributionSphereDetectorSweep();
  else { std::cerr << "unknown entry point " << entry << "\n"; return 2; }
  const bool ok = ensure_device();  // false: the entry point printed its error and returned early
  comm().finalize();
  isx_shutdown();
  return !ok ? 3 : (anyError() ? 4 : 0);
}
