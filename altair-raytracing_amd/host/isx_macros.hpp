// isx_macros.hpp — host driver that keeps the reference's ROOT-macro entry signatures
//  and routes them to libisx (include/isx.h).  No ROOT, no ROBAST.
//
// One namespace per reference macro file, because the files reuse names
// (sweepDetector, traceRays, setupOpticsManager) with different constants:
//
//   isxhost::fluxAtObserver          flux_at_observer/fluxAtObserver.C
//   isxhost::fluxAtObserverOptimize  flux_at_observer/fluxAtObserverOptimize.C
//   isxhost::fluxAtObserverFast      flux_at_observer/fluxAtObserverFast.C
//   isxhost::nonLambertianFlux       flux_at_observer/nonLambertianFlux.C
//   isxhost::nonLambertianFluxCopy   flux_at_observer/"nonLambertianFlux copy.C"
//   isxhost::rootMacros              makeIntegratingSphereNRays.C, integratingSphereDetectorSweep.C,
//                                    distributionSphereDetectorSweep.C
//
// `AOpticsManager*` becomes the opaque `OpticsManager*` (it holds an isx_config); the
// visualisation arguments (drawRays) are accepted and ignored; `threads` is accepted and
// ignored exactly as the reference ignores it (fluxAtObserverOptimize.C:433).
// Errors: message on std::cerr and early return, never an exception (reference behaviour,
// fluxAtObserverOptimize.C:485-488); existing files are never overwritten (getUniqueFilename).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../include/isx.h"

namespace isxhost {

constexpr double cm = 1.0;  // AOpticsManager::cm()
constexpr double nm = 1e-7; // AOpticsManager::nm() (unused by the physics: reflectance is a scalar)

// Detector (fluxAtObserver.C:31-145) — data + setPosition; the hit test runs on the GPU.
struct Detector {
  double x = 0, y = 0, z = 0;
  double nx = 0, ny = 0, nz = 0;
  double width, height;
  int hitCount = 0;
  explicit Detector(double w = 10 * cm, double h = 10 * cm) : width(w), height(h) {}
  void setPosition(double theta, double phi, double radius);  // fluxAtObserver.C:49-68
};

struct OpticsManager {  // stands in for AOpticsManager: geometry + surface parameters
  isx_config cfg;
  OpticsManager();
};

// --- run-wide settings (the reference has none: it uses the global, unseeded gRandom)
struct RunOptions {
  uint64_t seed = 0x5EED0001ull;   // env ISX_SEED
  uint64_t next_ray = 0;           // advances so successive sweeps use fresh Philox streams
  int device = 0;                  // env ISX_DEVICE
  long rays_override = -1;         // env ISX_RAYS: replaces the macros' hard-coded n (tests)
  bool quiet = false;              // env ISX_QUIET
  int flush_rows = 0;              // env ISX_FLUSH_ROWS: theta rows per launch of the per-position sweep, written and flushed
                                   // before the next launch starts (0 = as many as hold ~4e9 rays: one launch for the reference's n)
};
RunOptions& options();
// Binds libisx to options().device on first use; false (and a message on cerr) if no GPU.
bool ensure_device();
// true once any entry point has reported an error (they never throw and return void, like the reference's)
bool anyError();

// --- CSV plumbing shared by the sweeps (exposed for tests)
std::string getUniqueFilename(const std::string& basePath);  // fluxAtObserverOptimize.C:336-387
std::string currentTimeString();                              // "%Y-%m-%d %H:%M:%S"
struct FluxMapMeta {
  std::string title;               // "Flux Map Data" | "Flux Map Data (Trace-Once Method)" | ...
  std::string n_label;             // "Number of rays per position" | "Number of rays"
  std::string method_line;         // optional "# Method: ..." line
  long n = 0;
  double det_w = 40, det_h = 40, r_in = 100.1, r_out = 101, thetaMax = 170;
  int nTheta = 180, nPhi = 90;
  double reflectance = 0.99, roughness = 0.01;
  double src[3] = {-60, 0, -75}, dir[3] = {5, 0, 0};
  int maxReflections = 50000;
};
// header exactly as fluxAtObserverOptimize.C:504-518 / fluxAtObserverFast.C:1117-1132
std::string fluxmap_header(const FluxMapMeta& m, const std::string& generated);
// rows "theta,phi,fraction" fixed/6, theta-major (fold 2: twofold row order j, j+nPhi/2)
std::string fluxmap_rows(const uint64_t* hits, long n, int nTheta, int nPhi, int fold = 1, int row0 = 0, int nRows = -1);

// --- offline analysis of a flux-map CSV, the numeric part of flux_at_observer/flux_analysis.py
// (process_file :11-57, per-theta mean + standard error :182-199, fit a*cos(b*theta)+c :60-62,200-209,
// R^2 :229-233).  Writes "<stem>_theta_analysis.txt" next to the CSV and prints the script's lines.
struct FluxMapTable {                           // process_file(): metadata dict + the three columns
  std::map<std::string, std::string> metadata;
  std::vector<double> theta, phi, fraction;
};
bool readFluxMap(const std::string& csvPath, FluxMapTable& out);
// Analytic integrating-sphere model of finitePort/: port area fraction f = (1-cos(180deg-thetaMax))/2,
// Phi_eff = 1/(1-rho(1-f)) (test.py:11-14, subtendedFlux.py:18-20), exit probability f*Phi_eff.
struct PortModel { double f = 0, phi_eff = 0, p_exit = 0; };
PortModel portModel(double thetaMaxDeg, double reflectance);
// finite-port projection factor, projectionFactor.py:19-46 (theta in radians, grid of num_points^2)
double projectionFactor(double theta, double R, double r_p, int num_points = 100);
struct ThetaAnalysis {
  std::vector<double> theta, mean, stderr_;   // one entry per theta row
  std::vector<double> lambertian, finite_port; // analytic overlays in the map's units (fraction of launched rays)
  PortModel port;
  double a = 0, b = 0, c = 0, r_squared = 0;
  bool fit_ok = false;
  std::string metadata_port_angle;            // "Exit port angle" header value, if present
};
bool analyzeFluxMap(const std::string& csvPath, ThetaAnalysis& out, bool writeReport = true);
// `flux_analysis.py <dir> [average]`: every *.csv of the folder, plus (average) the per-(theta,phi) mean over
// files with standard error across files (:128-160) as a last entry.  Writes "<dir>[_averaged]_theta_comparison.txt".
bool analyzeFluxMapFolder(const std::string& dir, bool average, std::vector<ThetaAnalysis>& out, bool writeReport = true);

namespace fluxAtObserver {
void setupOpticsManager(OpticsManager* manager);                                  // :147-160
bool isRayPassingThroughExitPort(const double lastPoint[3], double exitPortZ);    // :162-166
int traceRays(OpticsManager* manager, int n, double exitPortZ, Detector& detector, bool drawRays = false);  // :169-228
void sweepDetector();                                                             // :231-406
}  // namespace fluxAtObserver

namespace fluxAtObserverOptimize {
constexpr double THETA_MAX = 170.;
constexpr int MAX_REFLECTIONS = 50000;
constexpr double INNER_RADIUS = 100.1 * cm, OUTER_RADIUS = 101 * cm, REFLECTANCE = 0.99, ROUGHNESS = 0.01;
void setupOpticsManager(OpticsManager* manager, int maxReflections = MAX_REFLECTIONS, double roughness = ROUGHNESS,
                        double reflectance = REFLECTANCE, double thetaMax = THETA_MAX, bool drawRays = false);  // :192-230
int traceRays(OpticsManager* manager, int n, double exitPortZ, Detector& detector, bool drawRays = false,
              int maxPoints = MAX_REFLECTIONS);                                                                   // :239-278
int traceRaysParallel(OpticsManager* manager, int n, double exitPortZ, Detector& detector, bool drawRays = false,
                      double x = -60 * cm, double y = 0 * cm, double z = -80 * cm, double dirX = 5, double dirY = 2,
                      double dirZ = 0);                                                                            // :281-333
void sweepDetector(bool notify = true, const char* saveFolder = "results", int threads = -1, double srcX = -60 * cm,
                   double srcY = 0 * cm, double srcZ = -80 * cm, double dirX = 5, double dirY = 2, double dirZ = 0,
                   double thetaMax = THETA_MAX);                                                                   // :433-702
void sweepSeries();                                                                                                // :892-921
}  // namespace fluxAtObserverOptimize

namespace fluxAtObserverFast {
int traceRaysParallelTwofold(OpticsManager* manager, int n, double exitPortZ, Detector& detector1, Detector& detector2,
                             bool drawRays = false, double x = -60 * cm, double y = 0 * cm, double z = -80 * cm,
                             double dirX = 5, double dirY = 2, double dirZ = 0);                                   // :336-408
void sweepDetectorTwofold(bool notify = true, const char* saveFolder = "results", int threads = -1,
                          double srcX = -60 * cm, double srcY = 0 * cm, double srcZ = -80 * cm, double dirX = 5,
                          double dirY = 2, double dirZ = 0, double thetaMax = 170.);                               // :518-865
void sweepDetectorTraceOnce(bool notify = true, const char* saveFolder = "results", int threads = -1,
                            double srcX = -60 * cm, double srcY = 0 * cm, double srcZ = -80 * cm, double dirX = 5,
                            double dirY = 2, double dirZ = 0, double thetaMax = 170.);                             // :1068-1397
void sweepSeries();                                                                                                // :1641-1673
}  // namespace fluxAtObserverFast

namespace nonLambertianFlux {
void setupOpticsManager(OpticsManager* manager);                                                                   // :213-226
int traceRays(OpticsManager* manager, int n, double exitPortZ, Detector& detector, bool drawRays = false);        // :235-304
void sweepDetector();                                                                                              // :307-387
}  // namespace nonLambertianFlux

// flux_at_observer/"nonLambertianFlux copy.C" (the de-facto CustomMirror: NonLambertianSurface, cos^2 lobe by rejection, :31-70,188-221)
namespace nonLambertianFluxCopy {
void setupOpticsManager(OpticsManager* manager);                                  // :213-255
int traceRays(OpticsManager* manager, int n, double exitPortZ, Detector& detector, bool drawRays = false);  // :263-303
void sweepDetector();                                                             // :306-386
void visualizeDetectorText(double theta = 45.0, double phi = 0.0);                // :604-667 (the text report: one 20 cm detector, 10 000 rays)
}  // namespace nonLambertianFluxCopy

namespace rootMacros {
void makeIntegratingSphereNRays();                                         // makeIntegratingSphereNRays.C:22-100
// inner sweep of integratingSphereDetectorSweep.C:31-105 (AOpticalComponent* world is folded into the manager)
void sweepDetector(OpticsManager* manager, double diskRadius, int nRays, double dtheta, double thetaMax);
void integratingSphereDetectorSweep();                                     // integratingSphereDetectorSweep.C:107-131
// disc placement of addDetectorDisk (:145-172): centre + tube axis after RotateZ(rotPhi), RotateY(rotTheta)
void detectorDiskPlacement(double theta, double phi, double out6[6]);
void distributionSphereDetectorSweep();                                    // distributionSphereDetectorSweep.C:26-130
}  // namespace rootMacros

}  // namespace isxhost
