// dfx_design.h -- design -> geometry on the host, fused (round 6): the lattice map of the reference's QuadGeometry / KagomeGeometry
// (difflexmm/geometry.py:607-952: node = static base vector + one row of the design), the polygon pass (area, centroid, polar moment about
// the centroid: geometry.py:71-127), compute_inertia (geometry.py:144-160) and the undeformed void angles (energy.py:204-219 with
// geometry.py:181-253 at zero displacement) in ONE loop over the blocks, and the cotangent of all of it in one loop back.  Plain C++,
// no device code: the design changes once per objective evaluation, not per time step, but at 16 x 128 x 128 the NumPy version of this
// (four fancy-index gathers and a dozen temporaries per design) was 20 ms of host time in front of a 5 ms solve.
// Compiled into libdfx (engine_abi.hip) and into the CPU port (oracle/cpu/dfx_cpu.cpp); checked against difflexmm_amd/geometry.py (NumPy,
// itself checked against the torch oracle and complex-step Jacobians) to 1e-13 in tests/test_host_helpers.py.
#pragma once
#include <math.h>
#include <stdint.h>

#include <vector>

#include "../../include/dfx.h"
#include "dfx_hostpar.h"

namespace dfx_design {

constexpr int kMaxNpb = 4;

// (the designs of a batch are independent: dfx_hostpar.h)
template <class F>
inline void for_each_design(int batch, size_t work_per_design, F&& body) { dfx_hostpar::for_each(batch, work_per_design, body); }

struct Poly {            // one block's polygon pass
  double area, cx, cy, ip, sgn_s, sgn_m;
};

inline Poly polygon(const double* v, int n) {
  double S = 0.0, nx = 0.0, ny = 0.0;
  for (int k = 0; k < n; ++k) {
    const int p = (k + n - 1) % n;
    const double cr = v[2 * p] * v[2 * k + 1] - v[2 * p + 1] * v[2 * k];
    S += cr; nx += (v[2 * p] + v[2 * k]) * cr; ny += (v[2 * p + 1] + v[2 * k + 1]) * cr;
  }
  Poly P;
  P.sgn_s = S < 0.0 ? -1.0 : 1.0;
  P.area = 0.5 * P.sgn_s * S;
  P.cx = nx / (6.0 * P.area); P.cy = ny / (6.0 * P.area);
  double M = 0.0;
  for (int k = 0; k < n; ++k) {
    const int p = (k + n - 1) % n;
    const double ax = v[2 * p] - P.cx, ay = v[2 * p + 1] - P.cy, bx = v[2 * k] - P.cx, by = v[2 * k + 1] - P.cy;
    M += (ax * by - ay * bx) * (ax * ax + ax * bx + bx * bx + ay * ay + ay * by + by * by);
  }
  M /= 12.0;
  P.sgn_m = M < 0.0 ? -1.0 : 1.0;
  P.ip = P.sgn_m * M;
  return P;
}

// cotangent of the vertices for cotangents (area_bar, (gx, gy) of the centroid, ip_bar): geometry.py polygon_props_vjp, same closed forms
inline void polygon_vjp(const double* v, int n, const Poly& P, double area_bar, double gx, double gy, double ip_bar, double* out) {
  double cr[kMaxNpb], tax[kMaxNpb], tay[kMaxNpb], tbx[kMaxNpb], tby[kMaxNpb];
  for (int k = 0; k < n; ++k) {
    const int p = (k + n - 1) % n;
    cr[k] = v[2 * p] * v[2 * k + 1] - v[2 * p + 1] * v[2 * k];
  }
  const double sm = P.sgn_m * ip_bar / 12.0;
  double ex = 0.0, ey = 0.0;
  for (int k = 0; k < n; ++k) {
    const int p = (k + n - 1) % n;
    const double ax = v[2 * p] - P.cx, ay = v[2 * p + 1] - P.cy, bx = v[2 * k] - P.cx, by = v[2 * k + 1] - P.cy;
    const double q = ax * ax + ax * bx + bx * bx + ay * ay + ay * by + by * by, w = ax * by - ay * bx;
    tbx[k] = -ay * q + w * (ax + 2.0 * bx); tby[k] = ax * q + w * (ay + 2.0 * by);
    tax[k] = by * q + w * (2.0 * ax + bx); tay[k] = -bx * q + w * (2.0 * ay + by);
    ex -= sm * (tax[k] + tbx[k]); ey -= sm * (tay[k] + tby[k]);
  }
  gx += ex; gy += ey;
  const double i6a = 1.0 / (6.0 * P.area), ia = 1.0 / P.area;
  for (int k = 0; k < n; ++k) {
    const int p = (k + n - 1) % n, q = (k + 1) % n;
    const double x = v[2 * k], y = v[2 * k + 1], x1 = v[2 * p], y1 = v[2 * p + 1], xn = v[2 * q], yn = v[2 * q + 1];
    const double dAx = 0.5 * P.sgn_s * (yn - y1), dAy = 0.5 * P.sgn_s * (x1 - xn);
    const double crn = cr[q];
    const double dNx_dx = cr[k] - (x1 + x) * y1 + crn + (x + xn) * yn, dNx_dy = (x1 + x) * x1 - (x + xn) * xn;
    const double dNy_dx = -(y1 + y) * y1 + (y + yn) * yn, dNy_dy = cr[k] + (y1 + y) * x1 + crn - (y + yn) * xn;
    out[2 * k] = area_bar * dAx + sm * (tbx[k] + tax[q]) + gx * (dNx_dx * i6a - P.cx * dAx * ia) + gy * (dNy_dx * i6a - P.cy * dAx * ia);
    out[2 * k + 1] = area_bar * dAy + sm * (tby[k] + tay[q]) + gx * (dNx_dy * i6a - P.cx * dAy * ia) + gy * (dNy_dy * i6a - P.cy * dAy * ia);
  }
}

inline double angle(double ux, double uy, double wx, double wy) { return atan2(ux * wy - uy * wx, ux * wx + uy * wy); }

inline int check(const dfx_design_map* m) {
  return (m && m->n_blocks > 0 && (m->n_npb == 3 || m->n_npb == 4) && m->n_design > 0 && m->base && m->gather && m->ref_points) ? 0 : 1;
}

inline int forward(const dfx_design_map* m, const double* design, int32_t batch, double density, double* centroids, double* cnv, double* inertia,
                   double* void_angle0) {
  if (check(m) || !design || !cnv) return 1;
  const int nb = m->n_blocks, n = m->n_npb;
  for_each_design(batch, (size_t)nb * n, [&](int mm) {
    const double* d = design + (size_t)mm * m->n_design * 2;
    double* cv = cnv + (size_t)mm * nb * n * 2;
    for (int b = 0; b < nb; ++b) {
      double v[2 * kMaxNpb];
      for (int k = 0; k < n; ++k) {
        const int g = m->gather[b * n + k];
        v[2 * k] = m->base[(b * n + k) * 2] + d[2 * g];
        v[2 * k + 1] = m->base[(b * n + k) * 2 + 1] + d[2 * g + 1];
      }
      const Poly P = polygon(v, n);
      for (int k = 0; k < n; ++k) { cv[(b * n + k) * 2] = v[2 * k] - P.cx; cv[(b * n + k) * 2 + 1] = v[2 * k + 1] - P.cy; }
      if (centroids) {
        centroids[((size_t)mm * nb + b) * 2] = m->ref_points[2 * b] + P.cx;
        centroids[((size_t)mm * nb + b) * 2 + 1] = m->ref_points[2 * b + 1] + P.cy;
      }
      if (inertia) {
        double* q = inertia + ((size_t)mm * nb + b) * 3;
        q[0] = q[1] = density * P.area; q[2] = density * P.ip;
      }
    }
    if (void_angle0 && m->bonds)
      for (int e = 0; e < m->n_bonds; ++e) {
        const int a = m->bonds[2 * e], c = m->bonds[2 * e + 1];
        const int b1 = a / n, l1 = a % n, b2 = c / n, l2 = c % n;
        const double* p1 = cv + (size_t)(b1 * n) * 2;
        const double* p2 = cv + (size_t)(b2 * n) * 2;
        const int n1 = (l1 + 1) % n, q1 = (l1 + n - 1) % n, n2 = (l2 + 1) % n, q2 = (l2 + n - 1) % n;
        const double e1px = p1[2 * n1] - p1[2 * l1], e1py = p1[2 * n1 + 1] - p1[2 * l1 + 1], e1mx = p1[2 * q1] - p1[2 * l1], e1my = p1[2 * q1 + 1] - p1[2 * l1 + 1];
        const double e2px = p2[2 * n2] - p2[2 * l2], e2py = p2[2 * n2 + 1] - p2[2 * l2 + 1], e2mx = p2[2 * q2] - p2[2 * l2], e2my = p2[2 * q2 + 1] - p2[2 * l2 + 1];
        double* o = void_angle0 + ((size_t)mm * m->n_bonds + e) * 2;
        o[0] = angle(e2mx, e2my, e1px, e1py);       // geometry.py:248-249
        o[1] = angle(e1mx, e1my, e2px, e2py);
      }
  });
  return 0;
}

inline int vjp(const dfx_design_map* m, const double* design, int32_t batch, double density, const double* cnv_bar, const double* centroid_bar,
               const double* inertia_bar, const double* void_bar, double* design_bar) {
  if (check(m) || !design || !cnv_bar || !design_bar) return 1;
  const int nb = m->n_blocks, n = m->n_npb;
  for_each_design(batch, (size_t)nb * n, [&](int mm) {
    const double* d = design + (size_t)mm * m->n_design * 2;
    double* db = design_bar + (size_t)mm * m->n_design * 2;
    for (int i = 0; i < m->n_design * 2; ++i) db[i] = 0.0;
    std::vector<double> ref((size_t)nb * n * 2), cb(cnv_bar + (size_t)mm * nb * n * 2, cnv_bar + (size_t)(mm + 1) * nb * n * 2);
    std::vector<Poly> props(nb);
    for (int b = 0; b < nb; ++b) {
      double* v = ref.data() + (size_t)b * n * 2;
      for (int k = 0; k < n; ++k) {
        const int g = m->gather[b * n + k];
        v[2 * k] = m->base[(b * n + k) * 2] + d[2 * g];
        v[2 * k + 1] = m->base[(b * n + k) * 2 + 1] + d[2 * g + 1];
      }
      props[b] = polygon(v, n);
    }
    if (void_bar && m->bonds) {        // void angles depend on edge vectors of the centred polygons = of the reference ones (geometry.py void_angles0_vjp)
      const double* vb = void_bar + (size_t)mm * m->n_bonds * 2;
      auto back = [&](int bu, int iu0, int iu1, int bw, int iw0, int iw1, double gbar) {
        // phi = atan2(u x w, u . w), u = p[iu1] - p[iu0] on block bu, w likewise on bw:  dphi/du = -perp(u)/|u|^2, dphi/dw = perp(w)/|w|^2
        if (gbar == 0.0) return;
        const double* pu = ref.data() + (size_t)bu * n * 2;
        const double* pw = ref.data() + (size_t)bw * n * 2;
        const double ux = pu[2 * iu1] - pu[2 * iu0], uy = pu[2 * iu1 + 1] - pu[2 * iu0 + 1], wx = pw[2 * iw1] - pw[2 * iw0], wy = pw[2 * iw1 + 1] - pw[2 * iw0 + 1];
        const double iu = gbar / (ux * ux + uy * uy), iw = gbar / (wx * wx + wy * wy);
        const double dux = uy * iu, duy = -ux * iu, dwx = -wy * iw, dwy = wx * iw;
        double* cu = cb.data() + (size_t)bu * n * 2;
        double* cw = cb.data() + (size_t)bw * n * 2;
        cu[2 * iu1] += dux; cu[2 * iu1 + 1] += duy; cu[2 * iu0] -= dux; cu[2 * iu0 + 1] -= duy;
        cw[2 * iw1] += dwx; cw[2 * iw1 + 1] += dwy; cw[2 * iw0] -= dwx; cw[2 * iw0 + 1] -= dwy;
      };
      for (int e = 0; e < m->n_bonds; ++e) {
        const int a = m->bonds[2 * e], c = m->bonds[2 * e + 1];
        const int b1 = a / n, l1 = a % n, b2 = c / n, l2 = c % n;
        const int n1 = (l1 + 1) % n, q1 = (l1 + n - 1) % n, n2 = (l2 + 1) % n, q2 = (l2 + n - 1) % n;
        back(b2, l2, q2, b1, l1, n1, vb[2 * e]);          // phi1: u = e2m, w = e1p
        back(b1, l1, q1, b2, l2, n2, vb[2 * e + 1]);      // phi2: u = e1m, w = e2p
      }
    }
    for (int b = 0; b < nb; ++b) {
      const double* v = ref.data() + (size_t)b * n * 2;
      const double* c = cb.data() + (size_t)b * n * 2;
      double gx = centroid_bar ? centroid_bar[((size_t)mm * nb + b) * 2] : 0.0, gy = centroid_bar ? centroid_bar[((size_t)mm * nb + b) * 2 + 1] : 0.0;
      for (int k = 0; k < n; ++k) { gx -= c[2 * k]; gy -= c[2 * k + 1]; }
      double ab = 0.0, ib = 0.0;
      if (inertia_bar) {
        const double* q = inertia_bar + ((size_t)mm * nb + b) * 3;
        ab = density * (q[0] + q[1]); ib = density * q[2];
      }
      double out[2 * kMaxNpb];
      polygon_vjp(v, n, props[b], ab, gx, gy, ib, out);
      for (int k = 0; k < n; ++k) {
        const int g = m->gather[b * n + k];
        db[2 * g] += c[2 * k] + out[2 * k];
        db[2 * g + 1] += c[2 * k + 1] + out[2 * k + 1];
      }
    }
  });
  return 0;
}

}  // namespace dfx_design
