"""The hand-derived Hessian-vector product of a ligament (bond_hvp / contact_hvp, difflexmm_amd/csrc/dfx_physics.h; reference energies
difflexmm/energy.py:70-176, 333-361) against the dual-number evaluation of the gradient it replaced in the reverse stage: value, (H w) on
the own block's DOFs and the epsilon part of dE/d(own node vector), on random ligaments, both ligament models, contact inside and outside
its range.  A g++ harness around the header the kernels are compiled from (no GPU needed)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = r"""
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include "dfx_physics.h"
using namespace dfx;
static double rnd(double a, double b) { return a + (b - a) * (double)rand() / RAND_MAX; }
template <int MODEL> static double run(int n, double* worst_contact) {
  double worst = 0.0;
  for (int it = 0; it < n; ++it) {
    BlockRec<double> o, p;
    auto fill = [&](BlockRec<double>& r, double cx, double cy) { r.x = cx + rnd(-1.5, 1.5); r.y = cy + rnd(-1.5, 1.5); r.th = rnd(-1.2, 1.2);
                                                               r.sh = sin(0.5 * r.th); r.ch = cos(0.5 * r.th); };
    fill(o, 0.0, 0.0); fill(p, 0.0, 0.0);
    const double rox = rnd(3, 7), roy = rnd(-7, 7), rpx = -rnd(3, 7), rpy = rnd(-7, 7);
    const double lx = rnd(1.5, 3.0), ly = rnd(-1.0, 1.0), l0 = sqrt(lx * lx + ly * ly), il0 = 1.0 / l0;
    const double ks = rnd(50, 200), ksh = rnd(0.5, 3), kr = rnd(0.5, 3), sgn = (it & 1) ? 1.0 : -1.0;
    const double w[6] = {rnd(-1, 1), rnd(-1, 1), rnd(-1, 1), rnd(-1, 1), rnd(-1, 1), rnd(-1, 1)};
    BondHvp hv;
    bond_hvp<MODEL>(o, p, w[0], w[1], w[2], w[3], w[4], w[5], rox, roy, rpx, rpy, lx, ly, l0, il0, ks, ksh, kr, sgn, hv);
    BlockRec<Dual> od = seed_rec(o, w[0], w[1], w[2]), pd = seed_rec(p, w[3], w[4], w[5]);
    BondGrad<Dual> g;
    bond_grad<MODEL, Dual>(od, pd, rox, roy, rpx, rpy, lx, ly, l0, il0, ks, ksh, kr, sgn, g);
    const double ref[8] = {g.fx.v, g.fy.v, g.fth.v, g.fx.e, g.fy.e, g.fth.e, g.rx.e, g.ry.e};
    const double got[8] = {hv.fx, hv.fy, hv.fth, hv.hx, hv.hy, hv.hth, hv.rx, hv.ry};
    double scale = 1e-300;
    for (int q = 0; q < 8; ++q) scale = std::max(scale, fabs(ref[q]));
    for (int q = 0; q < 8; ++q) worst = std::max(worst, fabs(got[q] - ref[q]) / scale);
    // angle contact: void angles such that the penalty is active for some draws, beyond the cutoff for others
    const double am = -0.26, ac = rnd(-0.1, 0.8), kc = rnd(0.5, 3), phi1 = rnd(0.0, 1.2), phi2 = rnd(0.0, 1.2);
    const double kap = sgn * (o.th - p.th), kapd = sgn * (w[2] - w[5]);
    double dk, dke, p1e, p2e;
    contact_hvp(kap, kapd, phi1, phi2, am, ac, kc, dk, dke, p1e, p2e);
    ContactGrad<Dual> cg;
    contact_grad<Dual>(Dual(kap, kapd), phi1, phi2, am, ac, kc, cg);
    const double cref[4] = {cg.dkap.v, cg.dkap.e, cg.p1.e, cg.p2.e}, cgot[4] = {dk, dke, p1e, p2e};
    double cs = 1e-300;
    for (int q = 0; q < 4; ++q) cs = std::max(cs, fabs(cref[q]));
    if (cs > 1e-200) for (int q = 0; q < 4; ++q) *worst_contact = std::max(*worst_contact, fabs(cgot[q] - cref[q]) / cs);
  }
  return worst;
}
int main() {
  srand(7);
  double wc = 0.0;
  const double a = run<kNonlinear>(200000, &wc), b = run<kLinearized>(200000, &wc);
  printf("%.3e %.3e %.3e\n", a, b, wc);
  return 0;
}
"""


def test_hand_written_hvp_equals_the_dual_number_gradient(tmp_path):
    src = tmp_path / "hvp.cpp"
    src.write_text(HARNESS)
    exe = tmp_path / "hvp"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "difflexmm_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                           "-o", str(exe), str(src)])
    nonlinear, linearized, contact = (float(x) for x in subprocess.check_output([str(exe)], text=True).split())
    assert nonlinear < 1e-12 and linearized < 1e-12 and contact < 1e-12, (nonlinear, linearized, contact)
    assert contact > 0.0                 # the penalty was active for some of the draws
