// isx_macros.cpp — the reference's macro entry points over libisx (see isx_macros.hpp).
#include "isx_macros.hpp"
#include "isx_comm.hpp"

#include <dirent.h>
#include <sys/stat.h>
#include <sys/types.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <algorithm>
#include <map>
#include <sstream>

namespace isxhost {

// Errors go to std::cerr and the entry point returns early (reference behaviour); the driver also remembers that one
// happened so that a command-line front end can exit non-zero.
static bool g_failed = false;
bool anyError() { return g_failed; }
static std::ostream& err() { g_failed = true; return std::cerr; }

// ---------------------------------------------------------------------------------------------
// run options / device
// ---------------------------------------------------------------------------------------------
RunOptions& options() {
  static RunOptions o = [] {
    RunOptions r;
    if (const char* s = std::getenv("ISX_SEED")) r.seed = std::strtoull(s, nullptr, 0);
    if (const char* s = std::getenv("ISX_DEVICE")) r.device = std::atoi(s);
    if (const char* s = std::getenv("ISX_RAYS")) r.rays_override = std::atol(s);
    if (const char* s = std::getenv("ISX_QUIET")) r.quiet = std::atoi(s) != 0;
    if (const char* s = std::getenv("ISX_FLUSH_ROWS")) r.flush_rows = std::atoi(s);
    return r;
  }();
  return o;
}

bool ensure_device() {
  static int state = 0;  // 0 unknown, 1 ok, -1 failed
  if (state == 0) {
    (void)comm();        // a multi-rank launch picks this rank's GPU before the first bind
    const int rc = isx_init(options().device);
    if (rc != ISX_OK) {
      err() << "Error: libisx cannot bind GPU " << options().device << ": " << isx_strerror(rc) << std::endl;
      state = -1;
    } else {
      state = 1;
    }
  }
  return state == 1;
}

// Every rank of a multi-rank launch reaches this point: true everywhere iff the device is bound on every rank and the
// output file (opened on the writer rank only) is usable -- so no rank walks away from a collective the others enter.
static bool ready_everywhere(bool file_ok = true) {
  const bool dev_ok = ensure_device();
  const bool ok = comm().agree(file_ok && dev_ok);
  if (!ok && file_ok && dev_ok)   // the cause was reported by another rank, or by isx_comm on this one
    err() << "Error: isx_comm: the job cannot run on every rank (device, output file or RCCL start-up); nothing was traced" << std::endl;
  return ok;
}

static uint64_t take_rays(uint64_t n) {
  const uint64_t first = options().next_ray;
  options().next_ray += n;
  return first;
}

static long pick_n(long reference_n) { return options().rays_override > 0 ? options().rays_override : reference_n; }

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------------------
// Detector / OpticsManager
// ---------------------------------------------------------------------------------------------
void Detector::setPosition(double theta, double phi, double radius) {
  // host mirror of fluxAtObserver.C:49-68 for ONE arbitrary position (bulk grids: isx_detector_table)
  const double theta_rad = theta * M_PI / 180.0;
  const double phi_rad = phi * M_PI / 180.0;
  double st, ct, sp, cp;
  ::sincos(theta_rad, &st, &ct);
  ::sincos(phi_rad, &sp, &cp);
  x = radius * st * cp;
  y = radius * st * sp;
  z = -100 * cm - radius * ct;
  const double dx = x - 0;
  const double dy = y - 0;
  const double dz = z - (-100 * cm);
  const double mag = std::sqrt(dx * dx + dy * dy + dz * dz);
  nx = -dy / mag;
  ny = dx / mag;
  nz = dz / mag;
}

OpticsManager::OpticsManager() { isx_default_config(&cfg); }

// ---------------------------------------------------------------------------------------------
// files
// ---------------------------------------------------------------------------------------------
static bool file_exists(const std::string& p) {
  FILE* f = std::fopen(p.c_str(), "r");
  if (!f) return false;
  std::fclose(f);
  return true;
}

// never overwrite (fluxAtObserverOptimize.C:336-387: the name the macros export; the behaviour is the contract): the first of
// <path>, <stem>_1<ext>, <stem>_2<ext>, ... that does not exist, the extension being whatever follows the last '.' of the file part
std::string getUniqueFilename(const std::string& basePath) {
  if (!file_exists(basePath)) return basePath;
  const size_t cut = basePath.find_last_of("/\\");
  const size_t name_at = cut == std::string::npos ? 0 : cut + 1;     // where the file part starts
  const size_t dot = basePath.find_last_of('.');
  const size_t ext_at = (dot != std::string::npos && dot >= name_at) ? dot : basePath.size();
  const std::string head = basePath.substr(0, ext_at), tail = basePath.substr(ext_at);
  for (int k = 1;; ++k) {
    const std::string candidate = head + "_" + std::to_string(k) + tail;
    if (!file_exists(candidate)) return candidate;
  }
}

static bool make_dirs(const std::string& path) {  // mkdir -p
  if (path.empty()) return true;
  std::string cur;
  for (size_t i = 0; i <= path.size(); ++i) {
    if (i == path.size() || path[i] == '/') {
      if (!cur.empty() && cur != "." && ::mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST) return false;
    }
    if (i < path.size()) cur += path[i];
  }
  struct stat st;
  return ::stat(path.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

std::string currentTimeString() {
  const time_t now = time(nullptr);
  char buf[80];
  strftime(buf, sizeof(buf), "%Y-%m-%d %H:%M:%S", localtime(&now));
  return buf;
}

std::string fluxmap_header(const FluxMapMeta& m, const std::string& generated) {
  std::ostringstream o;  // default stream formatting, as the reference before its first std::fixed
  o << "# " << m.title << " - Generated: " << generated << std::endl;
  o << "# " << m.n_label << ": " << m.n << std::endl;
  o << "# Detector dimensions: " << m.det_w << "cm x " << m.det_h << "cm" << std::endl;
  o << "# Sphere inner radius: " << m.r_in << "cm" << std::endl;
  o << "# Sphere outer radius: " << m.r_out << "cm" << std::endl;
  o << "# Exit port angle: " << m.thetaMax << " degrees" << std::endl;
  o << "# Theta bins: " << m.nTheta << std::endl;
  o << "# Phi bins: " << m.nPhi << std::endl;
  o << "# Mirror reflectance: " << m.reflectance << std::endl;
  o << "# Gaussian roughness: " << m.roughness << std::endl;
  o << "# Lambertian scattering: enabled" << std::endl;
  o << "# Source position (x,y,z): " << m.src[0] << "cm, " << m.src[1] << "cm, " << m.src[2] << "cm" << std::endl;
  o << "# Source direction (x,y,z): " << m.dir[0] << ", " << m.dir[1] << ", " << m.dir[2] << std::endl;
  o << "# Max reflections: " << m.maxReflections << std::endl;
  if (!m.method_line.empty()) o << "# Method: " << m.method_line << std::endl;
  o << "theta,phi,fraction" << std::endl;
  return o.str();
}

std::string fluxmap_rows(const uint64_t* hits, long n, int nTheta, int nPhi, int fold, int row0, int nRows) {
  std::ostringstream o;
  o << std::fixed << std::setprecision(6);
  const int rowEnd = nRows < 0 ? nTheta : std::min(nTheta, row0 + nRows);
  for (int i = row0; i < rowEnd; i++) {
    const double theta = (i + 0.5) * 90.0 / nTheta;
    if (fold == 2) {  // fluxAtObserverFast.C:693-720: (theta,phi1) then (theta,phi1+180)
      for (int j = 0; j < nPhi / 2; j++) {
        const double phi1 = (j + 0.5) * 360.0 / nPhi;
        double phi2 = phi1 + 180.0;
        if (phi2 >= 360.0) phi2 -= 360.0;
        const int j2 = j + nPhi / 2;
        o << theta << "," << phi1 << "," << double(hits[(size_t)i * nPhi + j]) / double(n) << std::endl;
        o << theta << "," << phi2 << "," << double(hits[(size_t)i * nPhi + j2]) / double(n) << std::endl;
      }
    } else {
      for (int j = 0; j < nPhi; j++) {
        const double phi = (j + 0.5) * 360.0 / nPhi;
        o << theta << "," << phi << "," << double(hits[(size_t)i * nPhi + j]) / double(n) << std::endl;
      }
    }
  }
  return o.str();
}

// ---------------------------------------------------------------------------------------------
// flux_analysis.py (numeric part)
// ---------------------------------------------------------------------------------------------
namespace {
// Levenberg-Marquardt for y = a*cos(b*x*pi/180) + c  (scipy.optimize.curve_fit's default for an unbounded problem)
bool fit_cosine(const std::vector<double>& x, const std::vector<double>& y, double p[3]) {
  const size_t n = x.size();
  if (n < 3) return false;
  auto model = [&](const double q[3], size_t k) { return q[0] * std::cos(q[1] * x[k] * M_PI / 180.0) + q[2]; };
  auto sse = [&](const double q[3]) { double s = 0; for (size_t k = 0; k < n; ++k) { const double r = y[k] - model(q, k); s += r * r; } return s; };
  double lambda = 1e-3, cur = sse(p);
  for (int it = 0; it < 200; ++it) {
    double JtJ[3][3] = {{0}}, Jtr[3] = {0};
    for (size_t k = 0; k < n; ++k) {
      const double arg = p[1] * x[k] * M_PI / 180.0;
      const double J[3] = {std::cos(arg), -p[0] * std::sin(arg) * x[k] * M_PI / 180.0, 1.0};
      const double r = y[k] - model(p, k);
      for (int i = 0; i < 3; ++i) { Jtr[i] += J[i] * r; for (int j = 0; j < 3; ++j) JtJ[i][j] += J[i] * J[j]; }
    }
    bool improved = false;
    for (int tries = 0; tries < 30 && !improved; ++tries) {
      double A[3][4];
      for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) A[i][j] = JtJ[i][j] + (i == j ? lambda * JtJ[i][i] : 0.0); A[i][3] = Jtr[i]; }
      bool singular = false;
      for (int c = 0; c < 3 && !singular; ++c) {  // Gauss-Jordan with partial pivoting
        int piv = c;
        for (int r = c + 1; r < 3; ++r) if (std::fabs(A[r][c]) > std::fabs(A[piv][c])) piv = r;
        if (std::fabs(A[piv][c]) < 1e-300) { singular = true; break; }
        if (piv != c) for (int j = 0; j < 4; ++j) std::swap(A[piv][j], A[c][j]);
        for (int r = 0; r < 3; ++r) if (r != c) { const double f = A[r][c] / A[c][c]; for (int j = c; j < 4; ++j) A[r][j] -= f * A[c][j]; }
      }
      if (singular) { lambda *= 10; continue; }
      double q[3];
      for (int i = 0; i < 3; ++i) q[i] = p[i] + A[i][3] / A[i][i];
      const double s = sse(q);
      if (s < cur) {
        const double rel = (cur - s) / (cur > 0 ? cur : 1.0);
        for (int i = 0; i < 3; ++i) p[i] = q[i];
        cur = s; lambda = std::max(lambda * 0.3, 1e-12); improved = true;
        if (rel < 1e-14) return true;
      } else lambda *= 10;
    }
    if (!improved) return true;  // converged: no step decreases the residual
  }
  return true;
}
}  // namespace

bool readFluxMap(const std::string& csvPath, FluxMapTable& out) {
  std::ifstream in(csvPath);
  if (!in.is_open()) {
    std::cerr << "File not found: " << csvPath << std::endl;
    return false;
  }
  // process_file(): '#' lines are "key: value" metadata (split on the first ':'), the first other line is the
  // header, then theta,phi,fraction rows
  out = FluxMapTable();
  std::string line;
  bool header = false;
  auto trim = [](std::string& t) {
    const size_t a = t.find_first_not_of(" \t\r"), b = t.find_last_not_of(" \t\r");
    t = a == std::string::npos ? "" : t.substr(a, b - a + 1);
  };
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    if (line[0] == '#') {
      const size_t colon = line.find(':');
      if (colon != std::string::npos) {
        std::string key = line.substr(1, colon - 1), val = line.substr(colon + 1);
        trim(key); trim(val);
        out.metadata[key] = val;
      }
      continue;
    }
    if (!header) { header = true; continue; }
    double t, p, f;
    if (std::sscanf(line.c_str(), "%lf,%lf,%lf", &t, &p, &f) == 3) { out.theta.push_back(t); out.phi.push_back(p); out.fraction.push_back(f); }
  }
  if (out.theta.empty()) {
    err() << "Error reading CSV data from " << csvPath << std::endl;
    return false;
  }
  return true;
}

PortModel portModel(double thetaMaxDeg, double reflectance) {
  PortModel m;
  m.f = 0.5 * (1.0 - std::cos((180.0 - thetaMaxDeg) * M_PI / 180.0));
  m.phi_eff = 1.0 / (1.0 - reflectance * (1.0 - m.f));
  m.p_exit = m.f * m.phi_eff;
  return m;
}

double projectionFactor(double theta, double R, double r_p, int num_points) {
  // the grid sum of safe_projection_factor: r and phi on inclusive linspaces, dA = r (r_p/n)(2 pi/n)
  const double tt = std::tan(theta);
  double flux = 0;
  for (int ip = 0; ip < num_points; ++ip) {
    const double ph = 2 * M_PI * double(ip) / double(num_points - 1), sp = std::sin(ph);
    for (int ir = 0; ir < num_points; ++ir) {
      const double r = r_p * double(ir) / double(num_points - 1);
      const double den = std::sqrt(std::max(R * R + r * r - 2 * R * r * sp * tt, 1e-10));
      const double c = std::min(1.0, std::max(-1.0, (R - r * sp * tt) / den));
      flux += c * r * (r_p / num_points) * (2 * M_PI / num_points);
    }
  }
  return flux;
}

namespace {
double leading_number(const std::map<std::string, std::string>& md, const char* key, double dflt) {
  auto it = md.find(key);
  if (it == md.end()) return dflt;
  char* e = nullptr;
  const double v = std::strtod(it->second.c_str(), &e);
  return e == it->second.c_str() ? dflt : v;
}

// per-theta statistics of one table: groupby('theta') mean, std (ddof=1; one row -> 0.001), count
void theta_groups(const std::vector<double>& th, const std::vector<double>& fr, const std::vector<double>* se_in, ThetaAnalysis& out) {
  std::vector<size_t> order(th.size());
  for (size_t k = 0; k < order.size(); ++k) order[k] = k;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return th[a] < th[b]; });
  out.theta.clear(); out.mean.clear(); out.stderr_.clear();
  for (size_t k = 0; k < order.size();) {
    size_t e = k;
    double s = 0, se = 0;
    while (e < order.size() && th[order[e]] == th[order[k]]) { s += fr[order[e]]; if (se_in) se += (*se_in)[order[e]]; ++e; }
    const double cnt = double(e - k), m = s / cnt;
    double ss = 0;
    for (size_t q = k; q < e; ++q) ss += (fr[order[q]] - m) * (fr[order[q]] - m);
    const double sd = cnt > 1 ? std::sqrt(ss / (cnt - 1)) : 0.001;
    out.theta.push_back(th[order[k]]); out.mean.push_back(m);
    out.stderr_.push_back(se_in ? se / cnt : sd / std::sqrt(cnt));   // AVERAGE: mean of the per-bin standard errors (:190)
    k = e;
  }
}

void fit_and_report(const std::string& label, ThetaAnalysis& out) {
  const double mx = *std::max_element(out.mean.begin(), out.mean.end()), mn = *std::min_element(out.mean.begin(), out.mean.end());
  double avg = 0;
  for (double v : out.mean) avg += v;
  avg /= double(out.mean.size());
  double p[3] = {(mx - mn) / 2, 1.0, avg};  // p0 of flux_analysis.py:203
  out.fit_ok = fit_cosine(out.theta, out.mean, p);
  if (!out.fit_ok) { p[0] = avg / 2; p[1] = 1.0; p[2] = avg / 2; }
  out.a = p[0]; out.b = p[1]; out.c = p[2];
  double ss_res = 0, ss_tot = 0;
  for (size_t k = 0; k < out.theta.size(); ++k) {
    const double r = out.mean[k] - (p[0] * std::cos(p[1] * out.theta[k] * M_PI / 180.0) + p[2]);
    ss_res += r * r; ss_tot += (out.mean[k] - avg) * (out.mean[k] - avg);
  }
  out.r_squared = ss_tot > 0 ? 1 - ss_res / ss_tot : 0;
  char buf[256];
  std::cout << "File: " << label << std::endl;
  std::snprintf(buf, sizeof(buf), "  Fit parameters: a=%.5f, b=%.5f, c=%.5f", out.a, out.b, out.c);
  std::cout << buf << std::endl;
  std::snprintf(buf, sizeof(buf), "  R-squared value: %.5f", out.r_squared);
  std::cout << buf << std::endl;
}

// analytic overlays (finitePort/): ideal Lambertian port P_exit*(rho_d/R_det)^2*cos(theta) and the same
// scaled by the finite-port projection factor normalised to its maximum
void overlays(const std::map<std::string, std::string>& md, ThetaAnalysis& out) {
  const double port = leading_number(md, "Exit port angle", 170.0), refl = leading_number(md, "Mirror reflectance", 0.99);
  const double r_in = leading_number(md, "Sphere inner radius", 100.1), det = leading_number(md, "Detector dimensions", 40.0);
  const double r_det = 100.0;  // fluxAtObserver.C:358
  out.port = portModel(port, refl);
  const double scale = out.port.p_exit * (0.5 * det / r_det) * (0.5 * det / r_det);
  const double r_p = r_in * std::sin((180.0 - port) * M_PI / 180.0);
  std::vector<double> pf(out.theta.size());
  double pfmax = 0;
  for (size_t k = 0; k < pf.size(); ++k) { pf[k] = projectionFactor(out.theta[k] * M_PI / 180.0, r_in, r_p); pfmax = std::max(pfmax, pf[k]); }
  out.lambertian.resize(pf.size()); out.finite_port.resize(pf.size());
  for (size_t k = 0; k < pf.size(); ++k) {
    out.lambertian[k] = scale * std::cos(out.theta[k] * M_PI / 180.0);
    out.finite_port[k] = pfmax > 0 ? scale * pf[k] / pfmax : 0.0;
  }
}

void write_report(const std::string& path, const std::string& label, const ThetaAnalysis& out) {
  std::ofstream o(path);
  o << "# theta analysis of " << label << " (flux_analysis.py equivalent)" << std::endl;
  if (!out.metadata_port_angle.empty()) o << "# Exit port angle: " << out.metadata_port_angle << std::endl;
  o << std::setprecision(10);
  o << "# fit: fraction = a*cos(b*theta) + c  a=" << out.a << " b=" << out.b << " c=" << out.c << " R2=" << out.r_squared << std::endl;
  o << "# analytic: port fraction f=" << out.port.f << " Phi_eff=" << out.port.phi_eff << " P_exit=f*Phi_eff=" << out.port.p_exit
    << " (finitePort/test.py:11-14); lambertian=P_exit*(r_det/R)^2*cos(theta); finite_port=projectionFactor.py:19-46 normalised" << std::endl;
  o << "theta,mean_fraction,stderr,fit,lambertian,finite_port" << std::endl;
  for (size_t k = 0; k < out.theta.size(); ++k)
    o << out.theta[k] << "," << out.mean[k] << "," << out.stderr_[k] << ","
      << (out.a * std::cos(out.b * out.theta[k] * M_PI / 180.0) + out.c) << "," << out.lambertian[k] << "," << out.finite_port[k] << std::endl;
}

std::string base_name(const std::string& path) {
  const size_t slash = path.find_last_of("/\\");
  return slash == std::string::npos ? path : path.substr(slash + 1);
}
}  // namespace

bool analyzeFluxMap(const std::string& csvPath, ThetaAnalysis& out, bool writeReport) {
  FluxMapTable tab;
  if (!readFluxMap(csvPath, tab)) return false;
  if (tab.metadata.count("Exit port angle")) out.metadata_port_angle = tab.metadata["Exit port angle"];
  theta_groups(tab.theta, tab.fraction, nullptr, out);
  fit_and_report(base_name(csvPath), out);
  overlays(tab.metadata, out);
  if (writeReport) {
    const size_t dot = csvPath.find_last_of('.');
    write_report(getUniqueFilename((dot == std::string::npos ? csvPath : csvPath.substr(0, dot)) + "_theta_analysis.txt"), base_name(csvPath), out);
  }
  return true;
}

bool analyzeFluxMapFolder(const std::string& dir, bool average, std::vector<ThetaAnalysis>& out, bool writeReport) {
  DIR* d = opendir(dir.c_str());
  if (!d) {
    std::cerr << "File not found: " << dir << std::endl;
    return false;
  }
  std::vector<std::string> files;
  while (dirent* e = readdir(d)) {
    const std::string n = e->d_name;
    if (n.size() > 4 && n.compare(n.size() - 4, 4, ".csv") == 0) files.push_back(n);
  }
  closedir(d);
  std::sort(files.begin(), files.end());   // os.listdir order is arbitrary; sorted here for reproducible reports
  if (files.empty()) {
    std::cerr << "No CSV files found in directory: " << dir << std::endl;
    return false;
  }
  out.clear();
  std::vector<FluxMapTable> tabs;
  std::vector<std::string> names;
  for (const std::string& n : files) {
    FluxMapTable t;
    if (!readFluxMap(dir + "/" + n, t)) continue;   // process_file returning None: skipped (:108-109)
    tabs.push_back(std::move(t)); names.push_back(n);
  }
  if (tabs.empty()) return false;
  for (size_t i = 0; i < tabs.size(); ++i) {
    ThetaAnalysis ta;
    if (tabs[i].metadata.count("Exit port angle")) ta.metadata_port_angle = tabs[i].metadata["Exit port angle"];
    theta_groups(tabs[i].theta, tabs[i].fraction, nullptr, ta);
    fit_and_report(names[i], ta);
    overlays(tabs[i].metadata, ta);
    out.push_back(ta);
  }
  std::string norm = dir;
  while (norm.size() > 1 && (norm.back() == '/' || norm.back() == '\\')) norm.pop_back();
  std::string base = base_name(norm);
  if (average && tabs.size() > 1) {
    // groupby(['theta','phi']) over all files: mean, std (ddof=1), count, stderr = std/sqrt(count) (:136-147)
    std::cout << "Averaging data across all files..." << std::endl;
    std::map<std::pair<double, double>, std::vector<double>> cells;
    for (const FluxMapTable& t : tabs)
      for (size_t k = 0; k < t.theta.size(); ++k) cells[{t.theta[k], t.phi[k]}].push_back(t.fraction[k]);
    std::vector<double> th, fr, se;
    for (const auto& kv : cells) {
      const std::vector<double>& v = kv.second;
      double m = 0;
      for (double x : v) m += x;
      m /= double(v.size());
      double ss = 0;
      for (double x : v) ss += (x - m) * (x - m);
      // pandas: std of a single value is NaN; NaN stderr rows then vanish from the per-theta mean (skipna)
      th.push_back(kv.first.first); fr.push_back(m);
      se.push_back(v.size() > 1 ? std::sqrt(ss / double(v.size() - 1)) / std::sqrt(double(v.size())) : std::nan(""));
    }
    ThetaAnalysis ta;
    // per-theta mean of the cell means, error bar = per-theta mean of the cell standard errors, NaNs skipped
    theta_groups(th, fr, nullptr, ta);
    {
      std::map<double, std::pair<double, int>> acc;
      for (size_t k = 0; k < th.size(); ++k) if (!std::isnan(se[k])) { acc[th[k]].first += se[k]; acc[th[k]].second++; }
      for (size_t k = 0; k < ta.theta.size(); ++k) {
        auto it = acc.find(ta.theta[k]);
        ta.stderr_[k] = it == acc.end() ? std::nan("") : it->second.first / it->second.second;
      }
    }
    fit_and_report("AVERAGE", ta);
    overlays(tabs[0].metadata, ta);
    out.push_back(ta);
    names.push_back("AVERAGE");
    base += "_averaged";
  }
  if (writeReport) {
    // one text file where the script saved <base>_theta_comparison.png
    const std::string rep = getUniqueFilename(base + "_theta_comparison.txt");
    std::ofstream o(rep);
    o << "# theta comparison of " << norm << " (flux_analysis.py equivalent, numbers instead of the PNG)" << std::endl;
    o << std::setprecision(10);
    o << "file,a,b,c,r_squared" << std::endl;
    for (size_t i = 0; i < out.size(); ++i) o << names[i] << "," << out[i].a << "," << out[i].b << "," << out[i].c << "," << out[i].r_squared << std::endl;
    o << "# per-theta means, one block per file" << std::endl;
    for (size_t i = 0; i < out.size(); ++i) {
      o << "# " << names[i] << std::endl << "theta,mean_fraction,stderr,fit,lambertian,finite_port" << std::endl;
      for (size_t k = 0; k < out[i].theta.size(); ++k)
        o << out[i].theta[k] << "," << out[i].mean[k] << "," << out[i].stderr_[k] << ","
          << (out[i].a * std::cos(out[i].b * out[i].theta[k] * M_PI / 180.0) + out[i].c) << "," << out[i].lambertian[k] << ","
          << out[i].finite_port[k] << std::endl;
    }
    std::cout << "Analysis saved as " << rep << std::endl;
  }
  return true;
}

static void say(const std::string& s) {
  if (!options().quiet) std::cout << s << std::endl;
}

// ---------------------------------------------------------------------------------------------
// fluxAtObserver.C
// ---------------------------------------------------------------------------------------------
namespace fluxAtObserver {

void setupOpticsManager(OpticsManager* m) {
  isx_config& c = m->cfg;
  isx_default_config(&c);
  c.max_points = 10000;          // manager->SetLimit(10000)
  c.box_half = 200 * cm;         // TGeoBBox 200
  c.r_in = 100.1 * cm; c.r_out = 101 * cm; c.theta_max_deg = 170.;
  c.reflectance = 1.0;           // AMirror default, no SetReflectance
  c.lambertian = 1;              // EnableLambertian(true)
  c.roughness_rad = 0.5;         // SetGaussianRoughness(0.5)
}

bool isRayPassingThroughExitPort(const double lastPoint[3], double exitPortZ) { return lastPoint[2] < exitPortZ; }

static int trace_one_detector(OpticsManager* m, int n, double exitPortZ, Detector& det, const double src[3],
                              const double dir[3], int source_model) {
  if (!ensure_device() || n <= 0) return 0;
  isx_config c = m->cfg;
  for (int k = 0; k < 3; ++k) { c.src[k] = src[k]; c.dir[k] = dir[k]; }
  c.exit_port_z = exitPortZ;
  c.source_model = source_model;
  const double d6[6] = {det.x, det.y, det.z, det.nx, det.ny, det.nz};
  uint64_t hits = 0;
  isx_stats st;
  const int rc = isx_trace_rays_detector(&c, d6, det.width, (uint64_t)n, options().seed, take_rays((uint64_t)n), &hits, &st);
  if (rc != ISX_OK) {
    err() << "Error: isx_trace_rays_detector: " << isx_strerror(rc) << std::endl;
    return 0;
  }
  det.hitCount += (int)hits;
  return (int)hits;
}

int traceRays(OpticsManager* m, int n, double exitPortZ, Detector& det, bool) {
  const double src[3] = {-60 * cm, 0 * cm, -80 * cm}, dir[3] = {5, 2, 0};  // :194-200
  return trace_one_detector(m, n, exitPortZ, det, src, dir, ISX_SOURCE_PENCIL);
}

void sweepDetector() {
  OpticsManager manager;
  setupOpticsManager(&manager);
  const long n = pick_n(50000);
  const double exitPortZ = -100 * cm;
  const int nThetaBins = 180, nPhiBins = 90;
  Detector detector;  // 10 cm x 10 cm
  const char* saveFolder = "results";
  if (!make_dirs(saveFolder)) std::cerr << "Warning: Could not create directory: " << saveFolder << std::endl;
  else say(std::string("Using directory: ") + saveFolder);
  std::string fullPath = std::string(saveFolder) + "/fluxmap_data_" + std::to_string(n) + "rays_" +
                         std::to_string(nThetaBins * nPhiBins) + "points.csv";
  fullPath = outputPath(fullPath);
  std::ofstream csvFile(fullPath);
  const bool file_ok = csvFile.is_open();   // decided with the other ranks below (ready_everywhere): no rank may leave before the collective
  if (!file_ok) err() << "Error: Could not open file " << fullPath << " for writing." << std::endl;
  const std::string timeBuffer = currentTimeString();
  csvFile << "# Flux Map Data - Generated: " << timeBuffer << std::endl;
  csvFile << "# Number of rays per position: " << n << std::endl;
  csvFile << "# Detector dimensions: 10cm x 10cm" << std::endl;
  csvFile << "# Sphere inner radius: 100.1cm" << std::endl;
  csvFile << "# Sphere outer radius: 101cm" << std::endl;
  csvFile << "# Exit port angle: " << 170. << " degrees" << std::endl;
  csvFile << "# Theta bins: " << nThetaBins << std::endl;
  csvFile << "# Phi bins: " << nPhiBins << std::endl;
  csvFile << "# y direction: 2" << std::endl;
  csvFile << "theta,phi,fraction" << std::endl;
  if (!ready_everywhere(file_ok)) return;
  isx_config c = manager.cfg;
  c.src[0] = -60; c.src[1] = 0; c.src[2] = -80;
  c.dir[0] = 5; c.dir[1] = 2; c.dir[2] = 0;
  c.n_theta = nThetaBins; c.n_phi = nPhiBins; c.det_diameter = detector.width; c.det_distance = 100 * cm;
  c.exit_port_z = exitPortZ;
  std::vector<uint64_t> hits((size_t)nThetaBins * nPhiBins);
  isx_stats st;
  const uint64_t total = (uint64_t)n * hits.size();
  const int rc = fluxmap_per_position_all(&c, (uint64_t)n, 1, hits.size(), options().seed, take_rays(total), hits.data(), &st);
  if (rc != ISX_OK) {
    err() << "Error: isx_fluxmap_per_position: " << isx_strerror(rc) << std::endl;
    return;
  }
  csvFile << fluxmap_rows(hits.data(), n, nThetaBins, nPhiBins);
  csvFile << "# Sweep completed at: " << timeBuffer << std::endl;  // (sic) the reference re-uses the start time, :384
  csvFile.close();
  say("\nFlux map data saved to '" + fullPath + "'");
}

}  // namespace fluxAtObserver

// ---------------------------------------------------------------------------------------------
// fluxAtObserverOptimize.C
// ---------------------------------------------------------------------------------------------
namespace fluxAtObserverOptimize {

void setupOpticsManager(OpticsManager* m, int maxReflections, double roughness, double reflectance, double thetaMax, bool) {
  isx_config& c = m->cfg;
  isx_default_config(&c);
  c.max_points = maxReflections;
  c.box_half = 300 * cm;
  c.r_in = INNER_RADIUS; c.r_out = OUTER_RADIUS; c.theta_max_deg = thetaMax;
  c.reflectance = reflectance;
  c.lambertian = 1;
  c.roughness_rad = roughness;
}

int traceRays(OpticsManager* m, int n, double exitPortZ, Detector& det, bool, int maxPoints) {
  const double src[3] = {-60 * cm, 0 * cm, -80 * cm}, dir[3] = {5, 0, 0};  // :251-252
  OpticsManager tmp = *m;
  tmp.cfg.max_points = maxPoints;
  return fluxAtObserver::trace_one_detector(&tmp, n, exitPortZ, det, src, dir, ISX_SOURCE_PENCIL);
}

int traceRaysParallel(OpticsManager* m, int n, double exitPortZ, Detector& det, bool, double x, double y, double z,
                      double dirX, double dirY, double dirZ) {
  const double src[3] = {x, y, z}, dir[3] = {dirX, dirY, dirZ};
  return fluxAtObserver::trace_one_detector(m, n, exitPortZ, det, src, dir, ISX_SOURCE_PENCIL);
}

static FluxMapMeta meta_for(const char* title, const char* nlabel, long n, double thetaMax, const double src[3],
                            const double dir[3]) {
  FluxMapMeta mm;
  mm.title = title; mm.n_label = nlabel; mm.n = n;
  mm.det_w = 40; mm.det_h = 40; mm.r_in = INNER_RADIUS / cm; mm.r_out = OUTER_RADIUS / cm; mm.thetaMax = thetaMax;
  mm.reflectance = REFLECTANCE; mm.roughness = ROUGHNESS; mm.maxReflections = MAX_REFLECTIONS;
  for (int k = 0; k < 3; ++k) { mm.src[k] = src[k] / cm; mm.dir[k] = dir[k]; }
  return mm;
}

void sweepDetector(bool notify, const char* saveFolder, int /*threads: ignored, as in the reference*/, double srcX,
                   double srcY, double srcZ, double dirX, double dirY, double dirZ, double thetaMax) {
  OpticsManager manager;
  setupOpticsManager(&manager, MAX_REFLECTIONS, ROUGHNESS, REFLECTANCE, thetaMax, false);
  const long n = pick_n(50000);
  const double exitPortZ = -100 * cm;
  const int nThetaBins = 180, nPhiBins = 90;
  if (!make_dirs(saveFolder)) std::cerr << "Warning: Could not create directory: " << saveFolder << std::endl;
  else say(std::string("Using directory: ") + saveFolder);
  std::string fullPath = std::string(saveFolder) + "/fluxmap_" + std::to_string(n) + "rays_" + std::to_string(nThetaBins) +
                         "x" + std::to_string(nPhiBins) + "_src" + std::to_string(int(srcX / cm)) + "_" +
                         std::to_string(int(srcY / cm)) + "_" + std::to_string(int(srcZ / cm)) + ".csv";
  fullPath = outputPath(fullPath);
  std::ofstream csvFile(fullPath);
  const bool file_ok = csvFile.is_open();   // decided with the other ranks below (ready_everywhere): no rank may leave before the collective
  if (!file_ok) err() << "Error: Could not open file " << fullPath << " for writing." << std::endl;
  Detector detector(40 * cm, 40 * cm);
  const double src[3] = {srcX, srcY, srcZ}, dir[3] = {dirX, dirY, dirZ};
  csvFile << fluxmap_header(meta_for("Flux Map Data", "Number of rays per position", n, thetaMax, src, dir), currentTimeString());
  const int totalPositions = nThetaBins * nPhiBins;
  say("\nStarting detector sweep with " + std::to_string(n) + " rays per position (" + std::to_string(totalPositions) +
      " positions total)...");
  if (!ready_everywhere(file_ok)) return;
  const double t0 = now_s();
  isx_config c = manager.cfg;
  for (int k = 0; k < 3; ++k) { c.src[k] = src[k]; c.dir[k] = dir[k]; }
  c.n_theta = nThetaBins; c.n_phi = nPhiBins; c.det_diameter = detector.width; c.det_distance = 100 * cm;
  c.exit_port_z = exitPortZ;
  std::vector<uint64_t> hits((size_t)totalPositions);
  isx_stats st{};
  const uint64_t total = (uint64_t)n * (uint64_t)totalPositions;
  const uint64_t first = take_rays(total);
  // The reference writes and flushes every row as it is produced (:575-579), so a run of hours leaves its rows behind if it
  // dies.  A map takes 0.3 s here; one made long (ISX_RAYS) is cut into batches of whole theta rows -- ISX_FLUSH_ROWS, default
  // as many rows as hold about 4e9 rays -- each ONE launch, written and flushed before the next starts.  Position g keeps its
  // rays [first + g n, +n) whatever the batching, so the rows are the same rows.
  int rowsPerBatch = options().flush_rows;
  if (rowsPerBatch < 1) rowsPerBatch = (int)std::max<uint64_t>(1, 4000000000ull / ((uint64_t)n * (uint64_t)nPhiBins));
  rowsPerBatch = std::min(rowsPerBatch, nThetaBins);
  for (int row0 = 0; row0 < nThetaBins; row0 += rowsPerBatch) {
    const int nRows = std::min(rowsPerBatch, nThetaBins - row0);
    isx_stats part{};
    const int rc = fluxmap_per_position_range_all(&c, (uint64_t)n, 1, (uint64_t)row0 * nPhiBins, (uint64_t)nRows * nPhiBins,
                                                  options().seed, first, hits.data(), &part);
    if (rc != ISX_OK) {   // (the status is the job's: every rank leaves here together)
      err() << "Error: isx_fluxmap_per_position: " << isx_strerror(rc) << " (theta rows " << row0 << "... not written)" << std::endl;
      return;
    }
    csvFile << fluxmap_rows(hits.data(), n, nThetaBins, nPhiBins, 1, row0, nRows);
    csvFile.flush();
    st.launched += part.launched; st.exited += part.exited; st.counted_below_z += part.counted_below_z;
    st.absorbed += part.absorbed; st.suspended += part.suspended; st.bin_increments += part.bin_increments;
    st.wall_hits += part.wall_hits; st.t_kernel_ms += part.t_kernel_ms;
  }
  const double realTime = now_s() - t0;
  // the stream is still in fixed/6 mode in the reference when the footer is written (:575,:668)
  csvFile << std::fixed << std::setprecision(6);
  csvFile << "# Sweep completed at: " << currentTimeString() << std::endl;
  csvFile << "# Total execution time: " << realTime << " seconds" << std::endl;
  csvFile << "# Total ray hits: " << st.bin_increments << " out of " << total << std::endl;
  csvFile.close();
  say("\nFlux map data saved to '" + fullPath + "'");
  if (!options().quiet) {
    std::cout << "Sweep completed in " << realTime << " seconds (wall clock), GPU kernels " << st.t_kernel_ms * 1e-3
              << " s, " << total / (st.t_kernel_ms * 1e-3) / 1e6 << " Mrays/s" << std::endl;
    if (notify) std::cout << "\n***** SWEEP COMPLETE *****\n" << std::endl << '\a' << std::endl;
  }
}

void sweepSeries() {
  const double srcX = -60 * cm, srcY = 0 * cm, srcZ = -75 * cm, dirXBase = 5;
  const std::string baseFolder = "results_overnight_04_1" + std::to_string(int(srcX / cm)) + "_" +
                                 std::to_string(int(srcY / cm)) + "_" + std::to_string(int(srcZ / cm)) + "_" +
                                 std::to_string(int(dirXBase));
  for (double port : {163., 166., 169., 172., 175., 178.})
    sweepDetector(false, baseFolder.c_str(), 1, srcX, srcY, srcZ, dirXBase, 0, 0, port);
}

}  // namespace fluxAtObserverOptimize

// ---------------------------------------------------------------------------------------------
// fluxAtObserverFast.C
// ---------------------------------------------------------------------------------------------
namespace fluxAtObserverFast {
using fluxAtObserverOptimize::INNER_RADIUS;
using fluxAtObserverOptimize::MAX_REFLECTIONS;
using fluxAtObserverOptimize::OUTER_RADIUS;
using fluxAtObserverOptimize::REFLECTANCE;
using fluxAtObserverOptimize::ROUGHNESS;

int traceRaysParallelTwofold(OpticsManager* m, int n, double exitPortZ, Detector& d1, Detector& d2, bool, double x,
                             double y, double z, double dirX, double dirY, double dirZ) {
  // the SAME n rays against two detectors: identical ray indices for both calls
  if (!ensure_device() || n <= 0) return 0;
  const uint64_t first = take_rays((uint64_t)n);
  isx_config c = m->cfg;
  c.src[0] = x; c.src[1] = y; c.src[2] = z; c.dir[0] = dirX; c.dir[1] = dirY; c.dir[2] = dirZ;
  c.exit_port_z = exitPortZ;
  int total = 0;
  for (Detector* d : {&d1, &d2}) {
    const double d6[6] = {d->x, d->y, d->z, d->nx, d->ny, d->nz};
    uint64_t h = 0;
    isx_stats st;
    const int rc = isx_trace_rays_detector(&c, d6, d->width, (uint64_t)n, options().seed, first, &h, &st);
    if (rc != ISX_OK) {
      err() << "Error: isx_trace_rays_detector: " << isx_strerror(rc) << std::endl;
      return total;
    }
    d->hitCount += (int)h;
    total += (int)h;
  }
  return total;
}

// pre != nullptr: the trace-once map was already computed (batched series); only the file is written
struct Precomputed { const uint64_t* hits; isx_stats st; };
static void sweep_common(bool traceOnce, bool notify, const char* saveFolder, double srcX, double srcY, double srcZ,
                         double dirX, double dirY, double dirZ, double thetaMax, const Precomputed* pre = nullptr) {
  const double tSetup = now_s();
  OpticsManager manager;
  fluxAtObserverOptimize::setupOpticsManager(&manager, MAX_REFLECTIONS, ROUGHNESS, REFLECTANCE, thetaMax, false);
  const long n = pick_n(traceOnce ? 100000 : 50000);
  const double exitPortZ = -100 * cm;
  const int nThetaBins = 180, nPhiBins = 90;
  const int totalPositions = nThetaBins * nPhiBins;
  if (!make_dirs(saveFolder)) std::cerr << "Warning: Could not create directory: " << saveFolder << std::endl;
  std::string fullPath = std::string(saveFolder) + (traceOnce ? "/fluxmap_traceonce_" : "/fluxmap_twofold_") +
                         std::to_string(n) + "rays_" + std::to_string(nThetaBins) + "x" + std::to_string(nPhiBins) +
                         "_src" + std::to_string(int(srcX / cm)) + "_" + std::to_string(int(srcY / cm)) + "_" +
                         std::to_string(int(srcZ / cm)) + ".csv";
  fullPath = outputPath(fullPath);
  const double src[3] = {srcX, srcY, srcZ}, dir[3] = {dirX, dirY, dirZ};
  FluxMapMeta mm;
  mm.title = traceOnce ? "Flux Map Data (Trace-Once Method)" : "Flux Map Data (Twofold Method)";
  mm.n_label = traceOnce ? "Number of rays" : "Number of rays per position";
  mm.method_line = traceOnce ? "Trace-Once (single trace, multiple detector positions)" : "Twofold (two detectors 180° apart)";
  mm.n = n; mm.r_in = INNER_RADIUS / cm; mm.r_out = OUTER_RADIUS / cm; mm.thetaMax = thetaMax;
  mm.reflectance = REFLECTANCE; mm.roughness = ROUGHNESS; mm.maxReflections = MAX_REFLECTIONS;
  for (int k = 0; k < 3; ++k) { mm.src[k] = src[k] / cm; mm.dir[k] = dir[k]; }
  std::ofstream csvFile(fullPath, std::ios::trunc);
  const bool file_ok = csvFile.is_open();   // decided with the other ranks below (ready_everywhere): no rank may leave before the collective
  if (!file_ok) err() << "Error: Could not open file " << fullPath << " for writing." << std::endl;
  csvFile << fluxmap_header(mm, currentTimeString());
  if (!ready_everywhere(file_ok)) return;
  isx_config c = manager.cfg;
  for (int k = 0; k < 3; ++k) { c.src[k] = src[k]; c.dir[k] = dir[k]; }
  c.n_theta = nThetaBins; c.n_phi = nPhiBins; c.det_diameter = 40 * cm; c.det_distance = 100 * cm;
  c.exit_port_z = exitPortZ;
  std::vector<uint64_t> hits((size_t)totalPositions);
  isx_stats st;
  int rc;
  const double t0 = now_s();
  if (pre) {
    std::copy(pre->hits, pre->hits + hits.size(), hits.begin());
    st = pre->st;
    rc = ISX_OK;
  } else if (traceOnce) {
    // one trace, every exiting line tested against all 16200 positions (fluxAtObserverFast.C:1143-1315), with
    // the per-position hit semantics (last point + final direction); the reference's GetPoint(nPoints-2)
    // defect (:1181,:1225) is deliberately not reproduced.
    rc = fluxmap_all(&c, (uint64_t)n, options().seed, take_rays((uint64_t)n), hits.data(), &st);
  } else {
    const uint64_t groups = (uint64_t)totalPositions / 2;
    rc = fluxmap_per_position_all(&c, (uint64_t)n, 2, groups, options().seed, take_rays((uint64_t)n * groups), hits.data(), &st);
  }
  if (rc != ISX_OK) {
    err() << "Error: libisx: " << isx_strerror(rc) << std::endl;
    return;
  }
  const double rayTime = st.t_kernel_ms * 1e-3;
  const double t1 = now_s();
  csvFile << fluxmap_rows(hits.data(), n, nThetaBins, nPhiBins, traceOnce ? 1 : 2);
  csvFile.close();
  const double sweepTime = now_s() - t1;
  const double totalTime = now_s() - tSetup;
  std::ofstream app(fullPath, std::ios::app);  // fresh stream => default number format, as in the reference (:1374-1382)
  if (app.is_open()) {
    app << "# Sweep completed at: " << currentTimeString() << std::endl;
    if (traceOnce) {
      app << "# Total execution time: " << totalTime << " seconds" << std::endl;
      app << "# Ray tracing time: " << rayTime << " seconds" << std::endl;
      app << "# Detector sweep time: " << sweepTime << " seconds" << std::endl;
      app << "# Total rays exiting port: " << st.counted_below_z << " out of " << n << std::endl;
    } else {
      app << std::fixed << std::setprecision(6);
      app << "# Total execution time: " << (now_s() - t0) << " seconds" << std::endl;
      app << "# Total ray hits: " << st.bin_increments << " out of " << (uint64_t)n * (uint64_t)totalPositions << std::endl;
    }
  }
  say("\nFlux map data saved to '" + fullPath + "'");
  if (!options().quiet) {
    std::cout << "Ray tracing completed in " << rayTime << " seconds" << std::endl;
    if (traceOnce) std::cout << "Total rays exiting port: " << st.counted_below_z << " out of " << n << std::endl;
    if (notify) std::cout << (traceOnce ? "\n***** TRACE-ONCE SWEEP COMPLETE *****\n" : "\n***** SWEEP COMPLETE *****\n") << std::endl << '\a' << std::endl;
  }
}

void sweepDetectorTwofold(bool notify, const char* saveFolder, int, double srcX, double srcY, double srcZ, double dirX,
                          double dirY, double dirZ, double thetaMax) {
  sweep_common(false, notify, saveFolder, srcX, srcY, srcZ, dirX, dirY, dirZ, thetaMax);
}

void sweepDetectorTraceOnce(bool notify, const char* saveFolder, int, double srcX, double srcY, double srcZ, double dirX,
                            double dirY, double dirZ, double thetaMax) {
  sweep_common(true, notify, saveFolder, srcX, srcY, srcZ, dirX, dirY, dirZ, thetaMax);
}

void sweepSeries() {
  const double srcX = -60 * cm, srcY = 0 * cm, srcZ = -75 * cm, dirXBase = 5, portAngle = 164.0;
  const std::string baseFolder = "portAngleSweep_04_03_" + std::to_string(int(srcX / cm)) + "_" +
                                 std::to_string(int(srcY / cm)) + "_" + std::to_string(int(srcZ / cm)) + "_" +
                                 std::to_string(int(portAngle));
  // the five repeats are traced back to back on the device with one synchronisation (isx_fluxmap_series),
  // each on its own ray-index range, then written as five files exactly as five calls would
  const int reps = 5;
  const long n = pick_n(100000);
  OpticsManager manager;
  fluxAtObserverOptimize::setupOpticsManager(&manager, MAX_REFLECTIONS, ROUGHNESS, REFLECTANCE, portAngle, false);
  isx_config c = manager.cfg;
  c.src[0] = srcX; c.src[1] = srcY; c.src[2] = srcZ; c.dir[0] = dirXBase; c.dir[1] = 0; c.dir[2] = 0;
  c.n_theta = 180; c.n_phi = 90; c.det_diameter = 40 * cm; c.det_distance = 100 * cm; c.exit_port_z = -100 * cm;
  if (!ready_everywhere()) return;
  std::vector<isx_config> cfgs(reps, c);
  std::vector<uint64_t> hits((size_t)reps * 16200);
  std::vector<isx_stats> st(reps);
  const int rc = fluxmap_series_all(cfgs.data(), reps, (uint64_t)n, options().seed, take_rays((uint64_t)n * reps), hits.data(), st.data());
  if (rc != ISX_OK) {
    err() << "Error: isx_fluxmap_series: " << isx_strerror(rc) << std::endl;
    return;
  }
  for (int i = 0; i < reps; i++) {
    Precomputed pre{hits.data() + (size_t)i * 16200, st[i]};
    pre.st.t_kernel_ms /= reps;
    sweep_common(true, false, baseFolder.c_str(), srcX, srcY, srcZ, dirXBase, 0, 0, portAngle, &pre);
  }
  if (!options().quiet) std::cout << "\n***** ALL SWEEP SERIES COMPLETE *****\n" << std::endl << '\a' << std::endl;
}

}  // namespace fluxAtObserverFast

// ---------------------------------------------------------------------------------------------
// nonLambertianFlux.C
// ---------------------------------------------------------------------------------------------
namespace nonLambertianFlux {

void setupOpticsManager(OpticsManager* m) {
  fluxAtObserver::setupOpticsManager(m);  // identical constants (:213-226)
  m->cfg.brdf[0] = 0.3; m->cfg.brdf[1] = 0.4; m->cfg.brdf[2] = 0.6;  // gBRDF(0.3, 0.4, 0.6) :211
}

int traceRays(OpticsManager* m, int n, double exitPortZ, Detector& det, bool) {
  const double src[3] = {-60 * cm, 0 * cm, -80 * cm}, dir[3] = {5, 0, 0};  // :243-244
  return fluxAtObserver::trace_one_detector(m, n, exitPortZ, det, src, dir, ISX_SOURCE_BRDF);
}

void sweepDetector() {
  OpticsManager manager;
  setupOpticsManager(&manager);
  const long n = pick_n(100000);
  const double exitPortZ = -100 * cm;
  const int nThetaBins = 45, nPhiBins = 20;
  Detector detector;
  if (!ready_everywhere()) return;
  isx_config c = manager.cfg;
  c.src[0] = -60; c.src[1] = 0; c.src[2] = -80; c.dir[0] = 5; c.dir[1] = 0; c.dir[2] = 0;
  c.n_theta = nThetaBins; c.n_phi = nPhiBins; c.det_diameter = detector.width; c.det_distance = 100 * cm;
  c.exit_port_z = exitPortZ; c.source_model = ISX_SOURCE_BRDF;
  std::vector<uint64_t> hits((size_t)nThetaBins * nPhiBins);
  isx_stats st;
  const uint64_t total = (uint64_t)n * hits.size();
  const int rc = fluxmap_per_position_all(&c, (uint64_t)n, 1, hits.size(), options().seed, take_rays(total), hits.data(), &st);
  if (rc != ISX_OK) {
    err() << "Error: isx_fluxmap_per_position: " << isx_strerror(rc) << std::endl;
    return;
  }
  const std::string path = outputPath("fluxmap_data.csv");  // the reference overwrites; this driver never does
  std::ofstream csvFile(path);
  if (!csvFile.is_open()) {   // after the last collective of this entry point: a local return is safe
    err() << "Error: Could not open file " << path << " for writing." << std::endl;
    return;
  }
  csvFile << "theta,phi,fraction\n";
  std::string rows = fluxmap_rows(hits.data(), n, nThetaBins, nPhiBins);
  csvFile << rows;
  csvFile.close();
  say("\nFlux map data saved to '" + path + "'");
}

}  // namespace nonLambertianFlux

// ---------------------------------------------------------------------------------------------
// flux_at_observer/"nonLambertianFlux copy.C": the same sweep with the NonLambertianSurface border (cos^2 lobe within 60 deg of
// the normal, :31-70,188-221) instead of ROBAST's own conditions -- ISX_SURFACE_LOBE; no BRDF re-scatter in this file
namespace nonLambertianFluxCopy {

void setupOpticsManager(OpticsManager* m) {
  fluxAtObserver::setupOpticsManager(m);     // limit 10000, box 200, shell 100.1-101, port 170, AMirror's default reflectance (:213-255)
  m->cfg.lambertian = 0;                     // condition->EnableLambertian(false) (:238); roughness 0.5 is inert: Reflection() is overridden
  m->cfg.surface_model = ISX_SURFACE_LOBE;   // NonLambertianSurface::Reflection (:188-221)
}

int traceRays(OpticsManager* m, int n, double exitPortZ, Detector& det, bool) {
  const double src[3] = {-60 * cm, 0 * cm, -80 * cm}, dir[3] = {5, 0, 0};  // ARay(0, 400 nm, -60, 0, -80, 0, 5, 0, 0) (:268-269)
  return fluxAtObserver::trace_one_detector(m, n, exitPortZ, det, src, dir, ISX_SOURCE_PENCIL);
}

void sweepDetector() {
  OpticsManager manager;
  setupOpticsManager(&manager);
  const long n = pick_n(100000);
  const double exitPortZ = -100 * cm;
  const int nThetaBins = 45, nPhiBins = 20;
  Detector detector;   // 10 cm x 10 cm (:78)
  if (!ready_everywhere()) return;
  isx_config c = manager.cfg;
  c.src[0] = -60; c.src[1] = 0; c.src[2] = -80; c.dir[0] = 5; c.dir[1] = 0; c.dir[2] = 0;
  c.n_theta = nThetaBins; c.n_phi = nPhiBins; c.det_diameter = detector.width; c.det_distance = 100 * cm;
  c.exit_port_z = exitPortZ; c.source_model = ISX_SOURCE_PENCIL;
  std::vector<uint64_t> hits((size_t)nThetaBins * nPhiBins);
  isx_stats st;
  const uint64_t total = (uint64_t)n * hits.size();
  // the macro's loop (:325-345) -- 900 positions x n fresh rays, one detector each -- in one launch
  const int rc = fluxmap_per_position_all(&c, (uint64_t)n, 1, hits.size(), options().seed, take_rays(total), hits.data(), &st);
  if (rc != ISX_OK) {
    err() << "Error: isx_fluxmap_per_position: " << isx_strerror(rc) << std::endl;
    return;
  }
  const std::string path = outputPath("fluxmap_data.csv");  // (:371; the reference overwrites, this driver never does)
  std::ofstream csvFile(path);
  if (!csvFile.is_open()) {
    err() << "Error: Could not open file " << path << " for writing." << std::endl;
    return;
  }
  csvFile << "theta,phi,fraction\n";
  csvFile << fluxmap_rows(hits.data(), n, nThetaBins, nPhiBins);
  csvFile.close();
  say("\nFlux map data saved to '" + path + "'");
}

// :604-667: the text report of one detector position -- position, normal, rays detected -- and the macro's ASCII sketch
void visualizeDetectorText(double theta, double phi) {
  OpticsManager manager;
  setupOpticsManager(&manager);
  Detector detector(20 * cm, 20 * cm);
  detector.setPosition(theta, phi, 100 * cm);
  const int n = (int)pick_n(10000);
  const double exitPortZ = -100 * cm;
  if (!ready_everywhere()) return;
  const int hitCount = traceRays(&manager, n, exitPortZ, detector, false);
  std::ostream& o = std::cout;
  o << "\n===================================" << std::endl;
  o << "DETECTOR INFORMATION" << std::endl;
  o << "===================================" << std::endl;
  o << "Angular position: theta = " << theta << "°, phi = " << phi << "°" << std::endl;
  o << "Position (x,y,z): (" << detector.x / cm << ", " << detector.y / cm << ", " << detector.z / cm << ") cm" << std::endl;
  o << "Normal vector: (" << detector.nx << ", " << detector.ny << ", " << detector.nz << ")" << std::endl;
  o << "Rays traced: " << n << std::endl;
  o << "Rays detected: " << hitCount << " (" << 100.0 * hitCount / n << "%)" << std::endl;
  o << "===================================" << std::endl;
  o << "\nTop View (X-Z plane, Y=0):" << std::endl;
  o << "    ^Z" << std::endl << "    |" << std::endl << "    |   ,-------," << std::endl << "    |  /         \\" << std::endl
    << "    | |     *     | Mirror" << std::endl << "    |  \\         /" << std::endl << "    |   '-------'" << std::endl << "    |" << std::endl;
  const int detX = (int)(detector.x / cm / 20) + 10;
  const int detZ = -(int)((detector.z + 100 * cm) / cm / 10) + 15;
  for (int z = 15; z >= 0; z--) {
    o << "    |";
    for (int x = 0; x < 20; x++) o << ((z == detZ && x == detX) ? "D" : " ");
    if (z == 15) o << "  Exit Port at Z=-100cm";
    if (z == detZ) o << "  <- Detector";
    o << std::endl;
  }
  o << "    +---------------------> X" << std::endl;
}

}  // namespace nonLambertianFluxCopy

// ---------------------------------------------------------------------------------------------
// root-level macros
// ---------------------------------------------------------------------------------------------
namespace rootMacros {

static void root_geometry(isx_config& c, double r_out) {
  isx_default_config(&c);
  c.max_points = 10000; c.box_half = 200 * cm; c.r_in = 100.1 * cm; c.r_out = r_out; c.theta_max_deg = 170.;
  c.reflectance = 1.0; c.lambertian = 1; c.roughness_rad = 0.0;
  c.src[0] = -60 * cm; c.src[1] = 0; c.src[2] = -80 * cm; c.dir[0] = 5; c.dir[1] = 0; c.dir[2] = 0;
}

void makeIntegratingSphereNRays() {
  if (!ready_everywhere()) return;
  isx_config c;
  root_geometry(c, 101 * cm);
  const long n = pick_n(1000);
  c.n_theta = 1; c.n_phi = 1;
  std::vector<uint64_t> hist(1);
  isx_stats st;
  const int rc = exit_dz_hist_all(&c, (uint64_t)n, options().seed, take_rays((uint64_t)n), 1, hist.data(), &st);
  if (rc != ISX_OK) {
    err() << "Error: libisx: " << isx_strerror(rc) << std::endl;
    return;
  }
  std::cout << "Flux of rays through the exit port: " << st.counted_below_z << std::endl;  // :93
}

void detectorDiskPlacement(double theta, double phi, double out[6]) {
  const double r = 200 * cm;  // :151-154
  const double x = r * std::sin(theta * M_PI / 180.0) * std::cos(phi * M_PI / 180.0);
  const double y = r * std::sin(theta * M_PI / 180.0) * std::sin(phi * M_PI / 180.0);
  const double z = -r * std::cos(theta * M_PI / 180.0);
  const double dx = 0 - x, dy = 0 - y, dz = -100 * cm - z;
  const double rotTheta = -std::atan2(std::sqrt(dx * dx + dy * dy), dz);  // radians here; degrees in the macro
  // rot->RotateZ(rotPhi); rot->RotateY(rotTheta): TGeoRotation left-multiplies, so the tube axis (local z) becomes
  // RY(rotTheta)*RZ(rotPhi)*ez = (sin rotTheta, 0, cos rotTheta) whatever rotPhi is (checked against detector_sweep.txt).
  out[0] = x; out[1] = y; out[2] = z;
  out[3] = std::sin(rotTheta); out[4] = 0.0; out[5] = std::cos(rotTheta);
}

void sweepDetector(OpticsManager* manager, double diskRadius, int nRays, double dtheta, double thetaMax) {
  const double dphi = 180;
  std::ofstream outFile(outputPath("detector_sweep3.txt"));
  const bool file_ok = outFile.is_open();
  if (!file_ok) err() << "Error: Could not open file detector_sweep3.txt for writing." << std::endl;
  outFile << "Theta(deg)\tPhi(deg)\tHitFraction\n";
  if (!ready_everywhere(file_ok)) return;
  // the reference's loop (integratingSphereDetectorSweep.C:54-77) places one disc, traces nRays fresh rays, moves on;
  // here every position keeps its own nRays fresh rays and all positions go through ONE launch
  std::vector<double> ca, angles;
  for (double theta = -thetaMax; theta <= thetaMax; theta += dtheta) {
    for (double phi = 0; phi < 360; phi += dphi) {
      double one[6];
      detectorDiskPlacement(theta, phi, one);
      ca.insert(ca.end(), one, one + 6);
      angles.push_back(theta); angles.push_back(phi);
    }
  }
  const int32_t nd = (int32_t)(ca.size() / 6);
  std::vector<uint64_t> hits((size_t)nd);
  isx_stats st;
  const int rc = disc_sweep_per_position_all(&manager->cfg, ca.data(), nd, diskRadius, 0.1 * cm, (uint64_t)nRays, options().seed,
                                             take_rays((uint64_t)nRays * (uint64_t)nd), hits.data(), &st);
  if (rc != ISX_OK) {
    err() << "Error: isx_disc_sweep_per_position: " << isx_strerror(rc) << std::endl;
    return;
  }
  for (int32_t k = 0; k < nd; ++k) {
    const double theta = angles[2 * (size_t)k], phi = angles[2 * (size_t)k + 1];
    const double hitFraction = static_cast<double>(hits[(size_t)k]) / nRays;
    if (!options().quiet)
      std::cout << "Theta: " << theta << "° Phi: " << phi << "° Hit fraction: " << hitFraction << std::endl;
    outFile << theta << "\t" << phi << "\t" << hitFraction << "\n";
  }
  outFile.close();
}

void integratingSphereDetectorSweep() {
  OpticsManager manager;
  root_geometry(manager.cfg, 105 * cm);  // TGeoSphere(100.1, 105, 0, 170) :118
  const int nRays = (int)pick_n(100000);
  sweepDetector(&manager, 5 * cm, nRays, 0.5, 45);
}

void distributionSphereDetectorSweep() {
  if (!ready_everywhere()) return;
  isx_config c;
  root_geometry(c, 101 * cm);
  const long n = pick_n(10000);
  std::vector<uint64_t> hist(100);
  isx_stats st;
  const int rc = exit_dz_hist_all(&c, (uint64_t)n, options().seed, take_rays((uint64_t)n), 100, hist.data(), &st);
  if (rc != ISX_OK) {
    err() << "Error: libisx: " << isx_strerror(rc) << std::endl;
    return;
  }
  std::cout << "Flux of rays through the exit port: " << st.counted_below_z << std::endl;
  // hDirectionZ = TH1D(100,-1,1) (:54,:91); written in the format of the committed angular_dist.txt
  std::ofstream f(outputPath("angular_dist.txt"));
  f << "# bin_center content\n";
  for (int b = 0; b < 100; ++b) f << (-1.0 + (b + 0.5) * 0.02) << " " << hist[b] << "\n";
  // the un-binned log of the same rays, in the format of the committed 3dRayLog.txt
  std::vector<uint64_t> ids((size_t)n);
  std::vector<double> dirs((size_t)n * 3);
  uint64_t count = 0;
  const uint64_t first = options().next_ray - (uint64_t)n;  // the very rays the histogram was made from
  const int rc2 = isx_exit_directions(&c, (uint64_t)n, options().seed, first, (uint64_t)n, ids.data(), dirs.data(), &count, &st);
  if (rc2 != ISX_OK) {
    err() << "Error: isx_exit_directions: " << isx_strerror(rc2) << std::endl;
    return;
  }
  std::ofstream lg(outputPath("3dRayLog.txt"));
  lg << "# dx dy dz\n";
  for (uint64_t k = 0; k < count; ++k) lg << dirs[3 * k] << " " << dirs[3 * k + 1] << " " << dirs[3 * k + 2] << "\n";
}

}  // namespace rootMacros

}  // namespace isxhost
