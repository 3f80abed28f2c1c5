// dumpBounces.C -- REFERENCE-SIDE recipe (ROOT 6 + ROBAST; never built or run in this repository, which has neither).
//
// Purpose: the one thing that can pin this build's CPU oracle at the reference boundary.  The arithmetic of the hot path
// is AOpticsManager::TraceNonSequential (ROBAST), configured as in flux_at_observer/fluxAtObserverOptimize.C:192-230 and
// called at :295; the reference holds no seeded vectors for it.  This macro produces them: it traces a few rays
// SINGLE-THREADED with a seeded, logging random generator and writes, per ray, every track point, the final direction, the
// end status, every uniform the generator handed out while the ray was traced, and Detector::checkIntersection's answer for
// a few fixed detector positions.  tests/test_robast_dump.py (skipped while tests/golden/robast_bounces.txt is absent)
// replays that file through the oracle: boundary search point by point, polar emission law draw by draw, detector test
// bit for bit.  How to run it: INTEGRATION.md, "Pinning the oracle".
//
//   root -l -b -q 'dumpBounces.C+(1000, 12345, "robast_bounces.txt")'      (with ROBAST's libROBAST.so loaded, as for the
//                                                                           reference's own macros)
//
// Written for this repository; it shares no text with the reference's macros.  Geometry parameters are the constants of
// fluxAtObserverOptimize.C:33-41 / sweepSeries() :892-896 (rho = 0.99, sigma = 0.01, port 170 deg, src (-60,0,-75), dir (5,0,0)).
#include <cstdio>
#include <vector>

#include "TGeoBBox.h"
#include "TGeoSphere.h"
#include "TRandom3.h"

#include "ABorderSurfaceCondition.h"
#include "AMirror.h"
#include "AOpticalComponent.h"
#include "AOpticsManager.h"
#include "ARay.h"

namespace {

// TRandom3 that remembers what it handed out.  Rndm() is the primitive everything else in TRandom goes through
// (Uniform, Gaus, Exp ... call the virtual Rndm()), so its log is the complete list of raw uniforms; Gaus() is bracketed
// so that the consumer can tell which uniforms fed a Gaussian draw (ABorderSurfaceCondition::SetGaussianRoughness).
class LoggingRandom : public TRandom3 {
 public:
  explicit LoggingRandom(UInt_t seed) : TRandom3(seed) {}
  Double_t Rndm() override {
    const Double_t u = TRandom3::Rndm();
    log.push_back(u);
    inGaus.push_back(gausDepth > 0 ? 1 : 0);
    return u;
  }
  Double_t Gaus(Double_t mean = 0, Double_t sigma = 1) override {
    ++gausDepth;
    const Double_t g = TRandom3::Gaus(mean, sigma);
    --gausDepth;
    gaus.push_back(g);
    return g;
  }
  std::vector<double> log, gaus;
  std::vector<char> inGaus;
  int gausDepth = 0;
};

// Detector::setPosition + checkIntersection, operation for operation as in flux_at_observer/fluxAtObserver.C:49-107
// (the macro cannot include the reference's file -- it defines entry points of the same names -- so the two short
// functions are restated; the maintainer may equally call the reference's own struct here).
struct Det {
  double x, y, z, nx, ny, nz, width;
  void set(double thetaDeg, double phiDeg, double radius, double portZ) {
    const double th = thetaDeg * M_PI / 180.0, ph = phiDeg * M_PI / 180.0;
    x = radius * sin(th) * cos(ph);
    y = radius * sin(th) * sin(ph);
    z = portZ - radius * cos(th);
    const double dx = x - 0, dy = y - 0, dz = z - portZ;
    const double mag = sqrt(dx * dx + dy * dy + dz * dz);
    nx = -dy / mag; ny = dx / mag; nz = dz / mag;
  }
  bool hit(const double* lp, const double* dir) const {
    const double dot = dir[0] * nx + dir[1] * ny + dir[2] * nz;
    if (fabs(dot) < 1e-10) return false;
    const double t = -((lp[0] - x) * nx + (lp[1] - y) * ny + (lp[2] - z) * nz) / dot;
    const double ix = lp[0] + dir[0] * t, iy = lp[1] + dir[1] * t, iz = lp[2] + dir[2] * t;
    const double rx = ix - x, ry = iy - y, rz = iz - z;
    const double ux = ny * rz - nz * ry, uy = nz * rx - nx * rz, uz = nx * ry - ny * rx;
    return ux * ux + uy * uy + uz * uz <= (width / 2) * (width / 2);
  }
};

}  // namespace

void dumpBounces(int nRays = 1000, unsigned seed = 12345, const char* outName = "robast_bounces.txt",
                 double reflectance = 0.99, double roughness = 0.01, double thetaMax = 170.0) {
  const double cm = AOpticsManager::cm(), nm = AOpticsManager::nm();
  const double rIn = 100.1 * cm, rOut = 101.0 * cm, boxHalf = 300.0 * cm, portZ = -100.0 * cm;
  const int limit = 50000;
  const double src[3] = {-60.0 * cm, 0.0, -75.0 * cm}, dir[3] = {5.0, 0.0, 0.0};

  LoggingRandom* rng = new LoggingRandom(seed);
  delete gRandom;
  gRandom = rng;

  AOpticsManager* manager = new AOpticsManager("manager", "bounce dump");
  manager->SetLimit(limit);
  AOpticalComponent* world = new AOpticalComponent("world", new TGeoBBox("box", boxHalf, boxHalf, boxHalf));
  manager->SetTopVolume(world);
  AMirror* mirror = new AMirror("mirror", new TGeoSphere("shell", rIn, rOut, 0.0, thetaMax));
  mirror->SetReflectance(reflectance);
  ABorderSurfaceCondition* border = new ABorderSurfaceCondition(world, mirror);
  border->EnableLambertian(true);
  border->SetGaussianRoughness(roughness);
  world->AddNode(mirror, 1);
  manager->SetNsegments(20);
  manager->CloseGeometry();
  world->Voxelize("");
  // (no SetMaxThreads: one thread, so the generator's log is the ray's own draws in order)

  const double probes[5][2] = {{0.25, 2.0}, {20.25, 46.0}, {45.25, 182.0}, {70.25, 270.0}, {89.75, 358.0}};   // (theta, phi) deg
  Det det[5];
  for (int k = 0; k < 5; ++k) { det[k].width = 40.0 * cm; det[k].set(probes[k][0], probes[k][1], 100.0 * cm, portZ); }

  FILE* f = fopen(outName, "w");
  if (!f) { fprintf(stderr, "cannot open %s\n", outName); return; }
  fprintf(f, "# isx-robast-bounce-dump 1\n");
  fprintf(f, "# seed %u rays %d reflectance %.17g roughness %.17g theta_max %.17g r_in %.17g r_out %.17g box_half %.17g "
             "limit %d port_z %.17g src %.17g %.17g %.17g dir %.17g %.17g %.17g cm %.17g\n",
          seed, nRays, reflectance, roughness, thetaMax, rIn, rOut, boxHalf, limit, portZ, src[0], src[1], src[2], dir[0], dir[1],
          dir[2], cm);
  for (int k = 0; k < 5; ++k)
    fprintf(f, "# detector %d theta %.17g phi %.17g : %.17g %.17g %.17g %.17g %.17g %.17g width %.17g\n", k, probes[k][0],
            probes[k][1], det[k].x, det[k].y, det[k].z, det[k].nx, det[k].ny, det[k].nz, det[k].width);

  for (int i = 0; i < nRays; ++i) {
    const size_t mark = rng->log.size(), gmark = rng->gaus.size();
    ARay ray(i, 660 * nm, src[0], src[1], src[2], 0, dir[0], dir[1], dir[2]);
    manager->TraceNonSequential(ray);
    const char status = ray.IsExited() ? 'E' : ray.IsStopped() ? 'S' : ray.IsAbsorbed() ? 'A' : ray.IsSuspended() ? 'U' : 'R';
    const int np = ray.GetNpoints();
    fprintf(f, "R %d %d %c %zu %zu\n", i, np, status, rng->log.size() - mark, rng->gaus.size() - gmark);
    for (int k = 0; k < np; ++k) {
      Double_t x, y, z, t;
      ray.GetPoint(k, x, y, z, t);   // TGeoTrack::GetPoint(Int_t, Double_t&, Double_t&, Double_t&, Double_t&): the INDEXED overload
      fprintf(f, "P %d %.17g %.17g %.17g\n", k, x, y, z);
    }
    fprintf(f, "U");
    for (size_t k = mark; k < rng->log.size(); ++k) fprintf(f, " %s%.17g", rng->inGaus[k] ? "g" : "", rng->log[k]);
    fprintf(f, "\nG");
    for (size_t k = gmark; k < rng->gaus.size(); ++k) fprintf(f, " %.17g", rng->gaus[k]);
    Double_t lp[3], d[3];
    ray.GetLastPoint(lp);
    ray.GetDirection(d);
    fprintf(f, "\nD %.17g %.17g %.17g", d[0], d[1], d[2]);
    for (int k = 0; k < 5; ++k) fprintf(f, " %d", (lp[2] < portZ && det[k].hit(lp, d)) ? 1 : 0);
    fprintf(f, "\n");
  }
  fclose(f);
  printf("wrote %s: %d rays, %zu uniforms\n", outName, nRays, rng->log.size());
}
