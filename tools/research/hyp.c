/*
 * hyp.c — RESEARCH TOOL (not product, not the oracle): a fast CPU tracer of the integrating sphere with
 * switches for every ROBAST behaviour the reference does not spell out, used to scan hypotheses against the
 * reference's seven 8.1e8-ray maps (tests/golden/reference_maps.npz; VERDICT r01 "next" #1).
 *
 * Statistics only: xoshiro256** streams (seeded per ray), libm, no bit-exactness contract.  The detector side
 * (setPosition / checkIntersection) is the reference's own arithmetic (fluxAtObserver.C:49-107).
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC -o libhyp.so hyp.c -lm      (tools/research/scan.py does it)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

typedef struct {
  double r_in, r_out, theta_max_deg, rho, sigma, box_half;
  double src[3], dir[3];
  int n_theta, n_phi;
  double det_diameter, det_distance, port_z;
  int max_points;
  /* hypothesis switches */
  int law;          /* 0 cosine; 1 uniform in cos(theta) (theta=acos(u)); 2 sin(theta)=u (theta=asin(u));
                       3 n + uniform point in unit BALL; 4 cos^law_pow lobe (pdf ~ cos^p per solid angle) */
  double law_pow;
  int rough_lambert; /* 1: tilt the normal by N(0,sigma) (uniform azimuth) before the Lambertian emission */
  int rim;          /* 0 same surface as the sphere; 1 absorbing; 2 absent (rays pass; shell has no thickness at the port);
                       3 specular */
  double rho_rim;   /* reflectance of the rim if >= 0 (else rho) */
  int outer;        /* 0 outer sphere mirror like the rest; 1 absorbing */
  int absorb_after; /* 1: the absorbed ray's last point still moves on?  (no effect on counts; placeholder) */
  double step_back; /* ROBAST stops fgkEpsilon before the mirror: hit point pulled back along the ray by this much */
  int count_absorbed; /* 1: absorbed rays whose last point is below port_z are counted too */
  int hit_line;     /* 0 last point + direction; 1 origin-compat (line from (0,0,0) through the last point) */
  int first_specular; /* 1: the first interaction is specular instead of Lambertian (testing) */
  double rho_angle_k; /* reflectance falls with incidence: rho_eff = rho * (1 - k * (1 - cos_i)) */
  int retry_into_wall; /* placeholder */
  int det_normal_mode; /* 0 reference quirk (-dy,dx,dz); 1 towards the port (dx,dy,dz)/mag */
  int port_test;       /* 0 last point z < port_z on the box; 1 every ray that leaves the shell downward */
  int two_sided;       /* 1: class exit rays by emission angle (invert.py) */
  double replay_q;     /* model of a racy shared RNG: with this probability a bounce re-uses the PREVIOUS bounce's random numbers */
  int replay_what;     /* 1 polar number only, 2 azimuth only, 3 both */
} hyp_cfg;

typedef struct {
  uint64_t launched, exited, counted, absorbed, suspended, wall_hits, rim_hits, outer_hits, increments;
} hyp_stats;

typedef struct { double x, y, z; } v3;
static inline double dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 add(v3 a, v3 b) { v3 r = { a.x + b.x, a.y + b.y, a.z + b.z }; return r; }
static inline v3 mul(v3 a, double s) { v3 r = { a.x * s, a.y * s, a.z * s }; return r; }
static inline v3 unit(v3 a) { return mul(a, 1.0 / sqrt(dot(a, a))); }
static inline v3 cross(v3 a, v3 b) { v3 r = { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; return r; }

typedef struct { uint64_t s[4]; } rng;
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t splitmix(uint64_t* x) {
  uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline void rng_seed(rng* r, uint64_t seed, uint64_t ray) {
  uint64_t x = seed * 0xD1342543DE82EF95ull + ray * 0x9E3779B97F4A7C15ull + 0x1234567;
  for (int i = 0; i < 4; i++) r->s[i] = splitmix(&x);
}
static inline uint64_t rng_next(rng* r) {
  uint64_t* s = r->s;
  uint64_t res = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
  s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
  return res;
}
static inline double rng_u(rng* r) { return ((double)(rng_next(r) >> 11) + 0.5) * 0x1.0p-53; }
static inline double rng_gaus(rng* r) { return sqrt(-2.0 * log(rng_u(r))) * cos(2.0 * M_PI * rng_u(r)); }

enum { K_NONE = 0, K_INNER, K_OUTER, K_CONE, K_BOX };

typedef struct {
  double rin2, rout2, zin, zout, k2, H;
} geom;

static int next_hit(const geom* g, const hyp_cfg* c, v3 p, v3 v, int on, v3* q, double* tt) {
  double best = INFINITY; int bk = K_BOX; v3 bq = p;
  double b = dot(p, v), pp = dot(p, p);
#define CONSIDER(T, KIND, COND) do { double t_ = (T); if (t_ > 1e-9 && t_ < best) { v3 q_ = add(p, mul(v, t_)); if (COND) { best = t_; bq = q_; bk = KIND; } } } while (0)
  double di = b * b - (pp - g->rin2);
  if (di >= 0) {
    double s = sqrt(di);
    if (!(on == K_INNER && b < 0)) CONSIDER(-b - s, K_INNER, q_.z >= g->zin);
    if (!(on == K_INNER && !(b < 0))) CONSIDER(s - b, K_INNER, q_.z >= g->zin);
    if (on == K_INNER && b < 0) CONSIDER(-2.0 * b, K_INNER, q_.z >= g->zin);
  }
  double dO = b * b - (pp - g->rout2);
  if (dO >= 0) {
    double s = sqrt(dO);
    if (!(on == K_OUTER && b < 0)) CONSIDER(-b - s, K_OUTER, q_.z >= g->zout);
    if (!(on == K_OUTER && !(b < 0))) CONSIDER(s - b, K_OUTER, q_.z >= g->zout);
  }
  if (c->rim != 2) {
    double A = v.x * v.x + v.y * v.y - g->k2 * v.z * v.z;
    double B = p.x * v.x + p.y * v.y - g->k2 * p.z * v.z;
    double C = p.x * p.x + p.y * p.y - g->k2 * p.z * p.z;
    if (on == K_CONE) { if (A != 0) CONSIDER(-2.0 * B / A, K_CONE, q_.z < 0 && dot(q_, q_) >= g->rin2 && dot(q_, q_) <= g->rout2); }
    else if (A == 0) { if (B != 0) CONSIDER(-C / (2 * B), K_CONE, q_.z < 0 && dot(q_, q_) >= g->rin2 && dot(q_, q_) <= g->rout2); }
    else {
      double D = B * B - A * C;
      if (D >= 0) {
        double sD = sqrt(D);
        CONSIDER((-B - sD) / A, K_CONE, q_.z < 0 && dot(q_, q_) >= g->rin2 && dot(q_, q_) <= g->rout2);
        CONSIDER((-B + sD) / A, K_CONE, q_.z < 0 && dot(q_, q_) >= g->rin2 && dot(q_, q_) <= g->rout2);
      }
    }
  }
  if (bk != K_BOX) { *q = bq; *tt = best; return bk; }
  double tx = v.x > 0 ? (g->H - p.x) / v.x : (v.x < 0 ? (-g->H - p.x) / v.x : INFINITY);
  double ty = v.y > 0 ? (g->H - p.y) / v.y : (v.y < 0 ? (-g->H - p.y) / v.y : INFINITY);
  double tz = v.z > 0 ? (g->H - p.z) / v.z : (v.z < 0 ? (-g->H - p.z) / v.z : INFINITY);
  double t = fmin(tx, fmin(ty, tz));
  *q = add(p, mul(v, t)); *tt = t;
  return K_BOX;
}

static inline void basis(v3 n, v3* a, v3* b) {
  v3 h = fabs(n.x) < 0.6 ? (v3){ 1, 0, 0 } : (v3){ 0, 1, 0 };
  *a = unit(cross(n, h));
  *b = cross(n, *a);
}

/* free-side unit normal */
static inline v3 normal_of(const geom* g, const hyp_cfg* c, int kind, v3 q) {
  if (kind == K_INNER) return mul(q, -1.0 / sqrt(dot(q, q)));
  if (kind == K_OUTER) return mul(q, 1.0 / sqrt(dot(q, q)));
  v3 n = { -q.x, -q.y, g->k2 * q.z };
  return unit(n);
}

static __thread double last_u1 = 0.5, last_u2 = 0.5;
static v3 emit(const hyp_cfg* c, v3 n, rng* r) {
  v3 a, b;
  double ct, st;
  if (c->law == 3) {
    /* n + uniform point in the unit ball */
    for (;;) {
      v3 u = { 2 * rng_u(r) - 1, 2 * rng_u(r) - 1, 2 * rng_u(r) - 1 };
      if (dot(u, u) <= 1.0) { v3 w = add(n, u); double m = dot(w, w); if (m > 1e-20) return mul(w, 1 / sqrt(m)); }
    }
  }
  double u1 = rng_u(r), u2 = rng_u(r);
  if (c->replay_q > 0 && rng_u(r) < c->replay_q) { if (c->replay_what & 1) u1 = last_u1; if (c->replay_what & 2) u2 = last_u2; }
  last_u1 = u1; last_u2 = u2;
  switch (c->law) {
    case 1: ct = u1; st = sqrt(1 - ct * ct); break;
    case 2: st = u1; ct = sqrt(1 - st * st); break;
    case 4: ct = pow(u1, 1.0 / (c->law_pow + 1.0)); st = sqrt(1 - ct * ct); break;
    case 8: default: st = sqrt(u1); ct = sqrt(1 - u1); break;
  }
  basis(n, &a, &b);
  double ph = 2 * M_PI * u2, cp = cos(ph), sp = sin(ph);
  v3 w = add(add(mul(a, st * cp), mul(b, st * sp)), mul(n, ct));
  return w;
}

typedef struct { int status; v3 p, v, last; int last_kind; uint64_t hits, rim, outer; } endstate;
enum { ST_EXIT = 1, ST_ABS = 2, ST_SUSP = 3 };

static void trace(const geom* g, const hyp_cfg* c, rng* r, endstate* es) {
  v3 p = { c->src[0], c->src[1], c->src[2] };
  v3 v = unit((v3){ c->dir[0], c->dir[1], c->dir[2] });
  int on = K_NONE, npts = 1;
  uint64_t nh = 0, nr = 0, no = 0;
  for (;;) {
    v3 q; double t;
    int kind = next_hit(g, c, p, v, on, &q, &t);
    npts++;
    if (kind == K_BOX) { es->status = ST_EXIT; es->p = q; es->v = v; es->last = p; es->last_kind = on; break; }
    if (c->step_back > 0) q = add(q, mul(v, -c->step_back));
    p = q; on = kind; nh++;
    if (kind == K_CONE) nr++;
    if (kind == K_OUTER) no++;
    v3 n = normal_of(g, c, kind, q);
    double rho = c->rho;
    if (kind == K_CONE) {
      if (c->rim == 1) rho = 0;
      else if (c->rho_rim >= 0) rho = c->rho_rim;
    }
    if (kind == K_OUTER && c->outer == 1) rho = 0;
    if (c->rho_angle_k != 0) { double ci = -dot(v, n); rho *= 1.0 - c->rho_angle_k * (1.0 - ci); }
    if (!(rng_u(r) < rho)) { es->status = ST_ABS; es->p = q; es->v = v; break; }
    v3 m = n;
    if (c->rough_lambert && c->sigma != 0) {
      v3 a, b; basis(n, &a, &b);
      double ph = 2 * M_PI * rng_u(r), d = c->sigma * rng_gaus(r);
      v3 e = add(mul(a, cos(ph)), mul(b, sin(ph)));
      m = add(mul(n, cos(d)), mul(e, sin(d)));
    }
    v3 w;
    if ((kind == K_CONE && c->rim == 3) || (c->first_specular && nh == 1)) {
      w = add(v, mul(m, -2.0 * dot(v, m)));
    } else if (c->law >= 5 && c->law <= 7) {
      /* "Lambertian" as a specular reflection about a RANDOM facet normal: 5 = facet normal cosine-distributed about n,
       * 6 = uniform over the hemisphere, 7 = cos^law_pow; retry_into_wall: 0 mirror the result back, 1 draw again */
      hyp_cfg cc = *c; cc.law = c->law == 5 ? 0 : (c->law == 6 ? 1 : 4);
      for (int tries = 0; tries < 1000; tries++) {
        v3 f = unit(emit(&cc, m, r));
        w = add(v, mul(f, -2.0 * dot(v, f)));
        if (!c->retry_into_wall || dot(w, n) > 0) break;
      }
    } else if (c->law == 8) {
      /* mixture: specular with probability law_pow, else cosine */
      if (rng_u(r) < c->law_pow) w = add(v, mul(m, -2.0 * dot(v, m))); else w = emit(c, m, r);
    } else {
      w = emit(c, m, r);
    }
    double dn = dot(w, n);
    if (dn <= 0) w = add(w, mul(n, -2.0 * dn));
    v = unit(w);
    if (npts > c->max_points) { es->status = ST_SUSP; es->p = q; es->v = v; break; }
  }
  es->hits = nh; es->rim = nr; es->outer = no;
}

static void det_set_position(const hyp_cfg* c, double theta, double phi, double* d) {
  double tr = theta * M_PI / 180.0, pr = phi * M_PI / 180.0, R = c->det_distance;
  double x = R * sin(tr) * cos(pr), y = R * sin(tr) * sin(pr), z = c->port_z - R * cos(tr);
  double dx = x, dy = y, dz = z - c->port_z, mag = sqrt(dx * dx + dy * dy + dz * dz);
  d[0] = x; d[1] = y; d[2] = z;
  if (c->det_normal_mode == 1) { d[3] = dx / mag; d[4] = dy / mag; d[5] = dz / mag; }
  else { d[3] = -dy / mag; d[4] = dx / mag; d[5] = dz / mag; }
}

static inline int check(const double* dt, double w, const double* lp, const double* dr) {
  double x = dt[0], y = dt[1], z = dt[2], nx = dt[3], ny = dt[4], nz = dt[5];
  double dotp = dr[0] * nx + dr[1] * ny + dr[2] * nz;
  if (fabs(dotp) < 1e-10) return 0;
  double dx = lp[0] - x, dy = lp[1] - y, dz = lp[2] - z;
  double t = -(dx * nx + dy * ny + dz * nz) / dotp;
  double ix = lp[0] + dr[0] * t, iy = lp[1] + dr[1] * t, iz = lp[2] + dr[2] * t;
  double rx = ix - x, ry = iy - y, rz = iz - z;
  double ux = ny * rz - nz * ry, uy = nz * rx - nx * rz, uz = nx * ry - ny * rx;
  return ux * ux + uy * uy + uz * uz <= (w / 2) * (w / 2);
}

/* hits[n_theta*n_phi]; dzhist[100] (exit direction z, TH1D(100,-1,1)); exitpos[64] radial histogram of where the
 * counted rays cross z = zin plane (0..32 cm, 0.5 cm bins) */
/* by_alpha[18][n_theta]: row sums of the map split by the exit direction's polar angle from -z (5 deg classes);
 * by_rad[40][n_theta]: split by the polar angle (from +z, 5 deg classes) of the LAST reflection point; 38 = rim, 39 = other */
int hyp_run(const hyp_cfg* c, uint64_t n, uint64_t seed, uint64_t* hits, hyp_stats* st_out, uint64_t* dzhist,
            uint64_t* radhist, uint64_t* by_alpha, uint64_t* by_rad) {
  geom g;
  double th = c->theta_max_deg * M_PI / 180.0;
  g.rin2 = c->r_in * c->r_in; g.rout2 = c->r_out * c->r_out;
  g.zin = c->r_in * cos(th); g.zout = c->r_out * cos(th);
  g.k2 = tan(th) * tan(th); g.H = c->box_half;
  size_t nb = (size_t)c->n_theta * c->n_phi;
  double* tab = malloc(nb * 6 * sizeof(double));
  float* cx = malloc(nb * sizeof(float)), *cy = malloc(nb * sizeof(float)), *cz = malloc(nb * sizeof(float));
  for (int i = 0; i < c->n_theta; i++)
    for (int j = 0; j < c->n_phi; j++) {
      size_t k = (size_t)i * c->n_phi + j;
      det_set_position(c, (i + 0.5) * 90.0 / c->n_theta, (j + 0.5) * 360.0 / c->n_phi, tab + 6 * k);
      cx[k] = (float)tab[6 * k]; cy[k] = (float)tab[6 * k + 1]; cz[k] = (float)tab[6 * k + 2];
    }
  memset(hits, 0, nb * sizeof(uint64_t));
  if (dzhist) memset(dzhist, 0, 100 * sizeof(uint64_t));
  if (radhist) memset(radhist, 0, 64 * sizeof(uint64_t));
  if (by_alpha) memset(by_alpha, 0, (size_t)18 * c->n_theta * sizeof(uint64_t));
  if (by_rad) memset(by_rad, 0, (size_t)40 * c->n_theta * sizeof(uint64_t));
  hyp_stats tot; memset(&tot, 0, sizeof(tot));
  const float lim = (float)((c->det_diameter / 2) * (c->det_diameter / 2) * 1.001 + 1.0);
#pragma omp parallel
  {
    uint32_t* h = calloc(nb, sizeof(uint32_t));
    uint64_t dzh[100] = { 0 }, rh[64] = { 0 };
    hyp_stats st; memset(&st, 0, sizeof(st));
    unsigned char* cand = malloc(nb);
    uint64_t* ba = calloc((size_t)18 * c->n_theta, sizeof(uint64_t));
    uint64_t* br = calloc((size_t)40 * c->n_theta, sizeof(uint64_t));
    const double rport = c->r_in * sin(th);
#pragma omp for schedule(dynamic, 512)
    for (int64_t i = 0; i < (int64_t)n; i++) {
      rng r; rng_seed(&r, seed, (uint64_t)i);
      endstate es; trace(&g, c, &r, &es);
      st.launched++; st.wall_hits += es.hits; st.rim_hits += es.rim; st.outer_hits += es.outer;
      int counted = 0;
      if (es.status == ST_EXIT) { st.exited++; if (c->port_test == 1 ? es.v.z < 0 : es.p.z < c->port_z) counted = 1; }
      else if (es.status == ST_ABS) { st.absorbed++; if (c->count_absorbed && es.p.z < c->port_z) counted = 1; }
      else st.suspended++;
      if (!counted) continue;
      st.counted++;
      double lp[3] = { es.p.x, es.p.y, es.p.z }, d[3] = { es.v.x, es.v.y, es.v.z };
      if (c->hit_line == 1) {
        double m = sqrt(dot(es.p, es.p));
        lp[0] = lp[1] = lp[2] = 0; d[0] = es.p.x / m; d[1] = es.p.y / m; d[2] = es.p.z / m;
      }
      if (dzhist) { int b = (int)floor((d[2] + 1.0) * 50.0); if (b >= 0 && b < 100) dzh[b]++; }
      if (radhist && d[2] < 0) {
        double t = (g.zin - lp[2]) / d[2];
        double x = lp[0] + t * d[0], y = lp[1] + t * d[1];
        int b = (int)(sqrt(x * x + y * y) * 2.0);
        if (b >= 0 && b < 64) rh[b]++;
      }
      /* cull: hit => distance(centre, line) <= w/2 */
      float px = (float)lp[0], py = (float)lp[1], pz = (float)lp[2], vx = (float)d[0], vy = (float)d[1], vz = (float)d[2];
      for (size_t k = 0; k < nb; k++) {
        float wx = cx[k] - px, wy = cy[k] - py, wz = cz[k] - pz;
        float wv = wx * vx + wy * vy + wz * vz;
        float d2 = wx * wx + wy * wy + wz * wz - wv * wv;
        cand[k] = d2 <= lim;
      }
      int ca = (int)(acos(fmin(1.0, fmax(-1.0, -es.v.z))) * (180.0 / M_PI) / 5.0);
      if (ca > 17) ca = 17;
      int cr = 39;   /* class of the last reflection point: polar angle from +z in 5 deg steps; 38 = rim; 39 = other */
      if (es.last_kind == K_INNER) {
        if (c->two_sided == 1) { /* class by the emission angle psi at the last reflection point */
          double cpsi = -dot(es.v, es.last) / sqrt(dot(es.last, es.last));
          cr = (int)(acos(fmin(1.0, fmax(-1.0, cpsi))) * (180.0 / M_PI) / 2.5); if (cr > 35) cr = 35;
        } else { cr = (int)(acos(es.last.z / sqrt(dot(es.last, es.last))) * (180.0 / M_PI) / 5.0); if (cr > 37) cr = 37; }
      }
      else if (es.last_kind == K_CONE) cr = 38;
      (void)rport;
      for (size_t k = 0; k < nb; k++)
        if (cand[k] && check(tab + 6 * k, c->det_diameter, lp, d)) {
          h[k]++; st.increments++;
          ba[(size_t)ca * c->n_theta + k / c->n_phi]++; br[(size_t)cr * c->n_theta + k / c->n_phi]++;
        }
    }
#pragma omp critical
    {
      for (size_t k = 0; k < nb; k++) hits[k] += h[k];
      if (dzhist) for (int k = 0; k < 100; k++) dzhist[k] += dzh[k];
      if (radhist) for (int k = 0; k < 64; k++) radhist[k] += rh[k];
      tot.launched += st.launched; tot.exited += st.exited; tot.counted += st.counted; tot.absorbed += st.absorbed;
      tot.suspended += st.suspended; tot.wall_hits += st.wall_hits; tot.rim_hits += st.rim_hits;
      tot.outer_hits += st.outer_hits; tot.increments += st.increments;
      if (by_alpha) for (size_t k = 0; k < (size_t)18 * c->n_theta; k++) by_alpha[k] += ba[k];
      if (by_rad) for (size_t k = 0; k < (size_t)40 * c->n_theta; k++) by_rad[k] += br[k];
    }
    free(h); free(cand); free(ba); free(br);
  }
  *st_out = tot;
  free(tab); free(cx); free(cy); free(cz);
  return 0;
}
