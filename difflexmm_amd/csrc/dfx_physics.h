// dfx_physics.h -- per-ligament physics of the DifFlexMM hot path, hand-derived.
//
// One header, compiled for gfx950 device code (dfx_kernels.hip) and for the host
// (oracle/cpu/dfx_cpu.cpp, the timed CPU port).  Everything is templated on the scalar
// type T: T = double gives energy gradients (forces); T = Dual (first-order forward-mode
// number) pushed through the SAME hand-written gradient gives, in the epsilon parts, the
// Hessian-vector product and every mixed parameter derivative the adjoint sweep needs
// (grad_{u,p} of  d/d eps E(u + eps w, p)).
//
// Reference formulas restated here (file:line relative to /root/reference):
//   node kinematics      difflexmm/kinematics.py:13-31   U = u_xy + (R(theta) - I) r
//   nonlinear ligament   difflexmm/energy.py:120-176
//   linearised ligament  difflexmm/energy.py:70-117
//   simple spring        difflexmm/energy.py:30-48    E = k_stretch (|dU + l0| - |l0|)^2 / 2
//   zero-length spring   difflexmm/energy.py:51-67    E = k_stretch |dU|^2 / 2 + k_rot (theta_2 - theta_1)^2 / 2
//   angle-based contact  difflexmm/energy.py:204-219,333-361 + geometry.py:181-253
//   distance contact     difflexmm/energy.py:222-330,364-407 (angle_based=False)
//   driving functions    problems/quads_focusing.py:211-222, tests/test_difflexmm.py:85-86,
//                        scripts/pulse_RS.py:49-50, problems/hinge_characterization.py:134-139
//
// Algebraic restatements (mathematically identical, cheaper on the GPU):
//  * shear strain  wrap(atan2(b) - atan2(R(tb) l0))  ==  atan2((R(tb) l0) x b, (R(tb) l0) . b)
//    (differs from the reference's floor-mod only at exactly +-pi);
//  * cos/sin of the mean rotation from the half-angle pair (cos(th/2), sin(th/2)) carried in
//    the per-block stage record, so one sincos per block per RHS instead of three per bond;
//  * a void angle is the angle between two edges of two rigid blocks, so it equals
//    wrap(phi +- (theta_A - theta_B)) with phi the same angle in the undeformed design:
//    block centroids and translations cancel exactly (energy.py:397-404 adds them, then
//    geometry.py:196-199 subtracts them again).  phi is computed once per solve on the host.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define DFX_HD __host__ __device__ __forceinline__
#else
#define DFX_HD inline
#endif

namespace dfx {

constexpr double kPi = 3.14159265358979323846;
constexpr double kTwoPi = 6.28318530717958647692;

// ---------------------------------------------------------------------------------------
// first-order forward-mode number
// ---------------------------------------------------------------------------------------
struct Dual {
  double v, e;
  DFX_HD Dual() : v(0.0), e(0.0) {}
  DFX_HD Dual(double a) : v(a), e(0.0) {}
  DFX_HD Dual(double a, double b) : v(a), e(b) {}
};
DFX_HD Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.e + b.e); }
DFX_HD Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.e - b.e); }
DFX_HD Dual operator-(Dual a) { return Dual(-a.v, -a.e); }
DFX_HD Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, a.v * b.e + a.e * b.v); }
DFX_HD Dual operator*(double a, Dual b) { return Dual(a * b.v, a * b.e); }
DFX_HD Dual operator*(Dual a, double b) { return Dual(a.v * b, a.e * b); }
DFX_HD Dual operator+(Dual a, double b) { return Dual(a.v + b, a.e); }
DFX_HD Dual operator+(double a, Dual b) { return Dual(a + b.v, b.e); }
DFX_HD Dual operator-(Dual a, double b) { return Dual(a.v - b, a.e); }
DFX_HD Dual operator-(double a, Dual b) { return Dual(a - b.v, -b.e); }
DFX_HD Dual operator/(Dual a, Dual b) {
  double r = 1.0 / b.v;
  double q = a.v * r;
  return Dual(q, (a.e - q * b.e) * r);
}
DFX_HD Dual operator/(double a, Dual b) {
  double r = 1.0 / b.v;
  double q = a * r;
  return Dual(q, -q * b.e * r);
}

DFX_HD double val(double a) { return a; }
DFX_HD double val(Dual a) { return a.v; }
DFX_HD double eps(double) { return 0.0; }
DFX_HD double eps(Dual a) { return a.e; }

// reciprocal: on the device v_rcp_f64 + two Newton steps (<= 1 ulp for normal inputs) instead of the ~14-instruction
// IEEE division expansion; exact division on the host.
DFX_HD double trcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}
DFX_HD Dual trcp(Dual a) {
  double r = trcp(a.v);
  return Dual(r, -a.e * r * r);
}
DFX_HD double tsqrt(double a) { return sqrt(a); }
DFX_HD Dual tsqrt(Dual a) {
  double s = sqrt(a.v);
  return Dual(s, 0.5 * a.e / s);
}
// 1/sqrt: one transcendental instead of sqrt + reciprocal (v_rsq_f64 + Newton on the device)
DFX_HD double trsqrt(double a) {
#if defined(__HIP_DEVICE_COMPILE__)
  return rsqrt(a);
#else
  return 1.0 / sqrt(a);
#endif
}
DFX_HD Dual trsqrt(Dual a) {
  double r = trsqrt(a.v);
  return Dual(r, -0.5 * a.e * r * r * r);
}
// atan2 with a short path for small angles in the right half plane: the shear angle of a ligament is almost always a
// few degrees, and the general fp64 atan2 is ~120 instructions.  |y| <= x/8: the alternating series through r^17
// truncates below 3e-18 relative.
DFX_HD double fast_atan2(double y, double x) {
  if (x > 0.0 && fabs(y) <= 0.125 * x) {
    const double r = y * trcp(x), z = r * r;
    double p = -1.0 / 17.0;
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, -1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, -1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, -1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    return fma(-(r * z), p, r);
  }
  return atan2(y, x);
}
DFX_HD double tatan2(double y, double x) { return fast_atan2(y, x); }
DFX_HD Dual tatan2(Dual y, Dual x) {
  return Dual(fast_atan2(y.v, x.v), (x.v * y.e - y.v * x.e) * trcp(x.v * x.v + y.v * y.v));
}
// sin and cos for |x| <= pi/4 by the fdlibm kernel polynomials (< 1 ulp), library sincos otherwise: the half rotation
// angle of a block is small except in extreme configurations, and the general fp64 sincos is ~190 instructions.
DFX_HD void fast_sincos(double x, double* sn, double* cs) {
  if (fabs(x) <= 0.78539816339744830962) {
    const double z = x * x;
    double ps = 1.58969099521155010221e-10;
    ps = fma(ps, z, -2.50507602534068634195e-08);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    *sn = fma(x * z, ps, x);
    double pc = -1.13596475577881948265e-11;
    pc = fma(pc, z, 2.08757232129817482790e-09);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    *cs = 1.0 - (0.5 * z - z * (z * pc));
  } else {
    sincos(x, sn, cs);
  }
}
// cos(th/2) from th and sh = sin(th/2): the stage records keep the sine only (32-byte records, two gathers per partner instead of
// three); |cos| = sqrt(1 - sh^2) is exact to rounding for the moderate rotations a lattice takes and loses digits only within a
// fraction of a degree of |th| = pi (absolute error ~1e-16 / |cos|); the sign follows the quadrant of th/2.
DFX_HD double half_cos(double th, double sh) {
  const double t = th * 0.07957747154594767280;                 // th / (4 pi): cos(th/2) = cos(2 pi t)
  const double f = t - rint(t);
  const double x = fmax(0.0, fma(-sh, sh, 1.0));
#if defined(__HIP_DEVICE_COMPILE__)
  // x is in (0, 1] and far from the denormals unless |th| is within 1e-150 of pi: v_rsq_f64 + two coupled Newton steps (<= 1 ulp)
  // instead of the correctly rounded library sqrt with its range scaling (~14 instructions, twice per lane in both stage kernels)
  const double r = __builtin_amdgcn_rsq(x);                     // (x = 0: inf, selected away below -- no branch)
  double y = x * r, hh = 0.5 * r;
  double e = fma(-hh, y, 0.5);
  y = fma(y, e, y); hh = fma(hh, e, hh);
  e = fma(-y, y, x);
  const double c = x > 1e-300 ? fma(e, hh, y) : 0.0;
#else
  const double c = sqrt(x);
#endif
  return fabs(f) <= 0.25 ? c : -c;
}
// wrap an angle into [-pi, pi] (value only; derivative 1)
DFX_HD double twrap(double a) { return a - kTwoPi * rint(a * (1.0 / kTwoPi)); }
DFX_HD Dual twrap(Dual a) { return Dual(twrap(a.v), a.e); }

// ---------------------------------------------------------------------------------------
// per-block stage record: (x, y, theta) plus the half-angle pair
// ---------------------------------------------------------------------------------------
template <class T>
struct BlockRec {
  T x, y, th, ch, sh;  // ch = cos(th/2), sh = sin(th/2)
};

DFX_HD BlockRec<Dual> seed_rec(const BlockRec<double>& r, double wx, double wy, double wth) {
  BlockRec<Dual> d;
  d.x = Dual(r.x, wx);
  d.y = Dual(r.y, wy);
  d.th = Dual(r.th, wth);
  d.ch = Dual(r.ch, -0.5 * r.sh * wth);
  d.sh = Dual(r.sh, 0.5 * r.ch * wth);
  return d;
}

enum BondModel { kLinearized = 0, kNonlinear = 1, kSimpleSpring = 2, kStretchTorsion = 3 };

// dE_bond / d(everything the OWN end of the bond owns) + bond parameters
template <class T>
struct BondGrad {
  T fx, fy, fth;   // dE/d(x, y, theta) of the own block
  T rx, ry;        // dE/d(centroid_node_vector of the own node)
  T lx, ly;        // dE/d(reference_vector)   (total)
  T ks, ksh, kr;   // dE/d(k_stretch, k_shear, k_rot)
  T e;             // bond energy
};

// What the PARTNER end of the same ligament owns, from the same evaluation (the tile kernels, dfx_tile.h, evaluate every
// ligament once -- energy.py:179-197 -- and hand the partner its half): its translational forces are -(fx, fy).
template <class T>
struct BondPartner {
  T fth;           // dE/d(theta) of the partner block
  T rx, ry;        // dE/d(centroid_node_vector of the partner's node)
};

// Bond (node1 on block A) -> (node2 on block B).  `o` is the own block, `p` the partner;
// sgn = +1 when the own block holds node2 (end B), -1 when it holds node1 (end A).
// (ro) / (rp) are the centroid->node vectors of the two bonded nodes, (lx,ly) the
// reference vector oriented node1 -> node2.
// T: type of the kinematic state (double, or Dual in the reverse sweep); P: type of the parameters -- double in every
// kernel: a parameter carries no epsilon part, and typing it as a Dual with a zero epsilon would cost an extra multiply-add
// per product (x * 0.0 cannot be folded away under IEEE semantics).
template <int MODEL, class T, class P>
DFX_HD void bond_grad(const BlockRec<T>& o, const BlockRec<T>& p, P rox, P roy, P rpx, P rpy,
                      P lx, P ly, double l0v, double il0v, P ks, P ksh, P kr, double sgn, BondGrad<T>& g, BondPartner<T>* pg = nullptr) {
  // rotation of own / partner block from the half angles
  T co = o.ch * o.ch - o.sh * o.sh, so = 2.0 * (o.sh * o.ch);
  T cp = p.ch * p.ch - p.sh * p.sh, sp = 2.0 * (p.sh * p.ch);
  // rotated node vectors and node displacements (kinematics.py:24-31)
  T qox = co * rox - so * roy, qoy = so * rox + co * roy;
  T qpx = cp * rpx - sp * rpy, qpy = sp * rpx + cp * rpy;
  T dUx = sgn * ((o.x + qox - rox) - (p.x + qpx - rpx));
  T dUy = sgn * ((o.y + qoy - roy) - (p.y + qpy - rpy));
  T kap = sgn * (o.th - p.th);  // theta_2 - theta_1
  // l0 = |(lx, ly)| and 1/l0 are per-solve constants (the reference vector is a parameter: it carries no
  // epsilon part, and every derivative w.r.t. it below is written in closed form)
  P l02 = lx * lx + ly * ly;
  const double l0 = l0v, il0 = il0v;
  T gbx, gby, gtb;  // dE/d(dU), dE/d(mean rotation)
  if (MODEL == kNonlinear) {
    // energy.py:139-155,172-176
    T cb = o.ch * p.ch - o.sh * p.sh, sb = o.sh * p.ch + o.ch * p.sh;  // cos/sin of (th_o+th_p)/2
    T bx = dUx + lx, by = dUy + ly;
    T L2 = bx * bx + by * by;
    T Lb = tsqrt(L2);          // correctly rounded: a lattice at rest has Lb == l0 bit for bit, hence exactly zero force
    T iLb = trcp(Lb);
    T iL2 = iLb * iLb;
    T px = cb * lx - sb * ly, py = sb * lx + cb * ly;
    T gam = tatan2(px * by - py * bx, px * bx + py * by);
    T es = Lb - l0;
    T kse = ks * es, kshg = ksh * gam;
    T shear = kshg * l02;  // dE/dgamma
    g.e = 0.5 * (kse * es) + 0.5 * (shear * gam) + 0.5 * (kr * (kap * kap));
    gbx = kse * bx * iLb - shear * by * iL2;
    gby = kse * by * iLb + shear * bx * iL2;
    gtb = -shear;
    g.lx = gbx - kse * (lx * il0) + kshg * gam * lx + kshg * ly;
    g.ly = gby - kse * (ly * il0) + kshg * gam * ly - kshg * lx;
    g.ks = 0.5 * (es * es);
    g.ksh = 0.5 * (l02 * (gam * gam));
  } else if (MODEL == kSimpleSpring) {
    // energy.py:30-48: the axial term of the nonlinear ligament alone (no shear, no bending: k_shear / k_rot are not read)
    T bx = dUx + lx, by = dUy + ly;
    T Lb = tsqrt(bx * bx + by * by);
    T iLb = trcp(Lb);
    T es = Lb - l0;
    T kse = ks * es;
    g.e = 0.5 * (kse * es);
    gbx = kse * bx * iLb;
    gby = kse * by * iLb;
    gtb = T(0.0);
    g.lx = gbx - kse * (lx * il0);
    g.ly = gby - kse * (ly * il0);
    g.ks = 0.5 * (es * es);
    g.ksh = T(0.0);
  } else if (MODEL == kStretchTorsion) {
    // energy.py:51-67: zero-length spring between two coincident nodes (the reference vector is not read)
    T kdx = ks * dUx, kdy = ks * dUy;
    g.e = 0.5 * (kdx * dUx + kdy * dUy) + 0.5 * (kr * (kap * kap));
    gbx = kdx;
    gby = kdy;
    gtb = T(0.0);
    g.lx = T(0.0);
    g.ly = T(0.0);
    g.ks = 0.5 * (dUx * dUx + dUy * dUy);
    g.ksh = T(0.0);
  } else {
    // energy.py:88-96,113-117
    T tb = 0.5 * (o.th + p.th);
    T dot = dUx * lx + dUy * ly;
    T crs = lx * dUy - ly * dUx;
    T es = dot * il0;
    T esh = crs * il0 - tb * l0;
    T kse = ks * es, kshe = ksh * esh;
    g.e = 0.5 * (kse * es) + 0.5 * (kshe * esh) + 0.5 * (kr * (kap * kap));
    gbx = (kse * lx - kshe * ly) * il0;
    gby = (kse * ly + kshe * lx) * il0;
    gtb = -(kshe * l0);
    const double il02 = il0 * il0;
    T c3 = crs * (il0 * il02);
    g.lx = kse * (dUx * il0 - es * (lx * il02)) + kshe * (dUy * il0 - c3 * lx - tb * (lx * il0));
    g.ly = kse * (dUy * il0 - es * (ly * il02)) + kshe * (-(dUx * il0) - c3 * ly - tb * (ly * il0));
    g.ks = 0.5 * (es * es);
    g.ksh = 0.5 * (esh * esh);
  }
  g.kr = MODEL == kSimpleSpring ? T(0.0) : 0.5 * (kap * kap);
  T krk = MODEL == kSimpleSpring ? T(0.0) : kr * kap;
  g.fx = sgn * gbx;
  g.fy = sgn * gby;
  g.fth = sgn * (gby * qox - gbx * qoy) + 0.5 * gtb + sgn * krk;
  g.rx = sgn * (co * gbx + so * gby - gbx);
  g.ry = sgn * (co * gby - so * gbx - gby);
  if (pg) {
    // seen from the partner (sgn' = -sgn, o' = p): dU, kap, gb and gtb are the same numbers
    pg->fth = 0.5 * gtb - sgn * ((gby * qpx - gbx * qpy) + krk);
    pg->rx = -(sgn * (cp * gbx + sp * gby - gbx));
    pg->ry = -(sgn * (cp * gby - sp * gbx - gby));
  }
}

// Angle-based contact of one bond (energy.py:333-361 on the two void angles of energy.py:204-219).
//   a1 = wrap(phi1 - kap), a2 = wrap(phi2 + kap), kap = theta_B - theta_A.
template <class T>
struct ContactGrad {
  T dkap;        // dE/dkap
  T p1, p2;      // dE/dphi1, dE/dphi2
  T am, ac, kc;  // dE/d(min_angle, cutoff_angle, k_contact)
  T e;
};

template <class T, class P, bool WRAP = true>
DFX_HD void contact_one(T a, P am, P ac, P kc, T& e, T& da, T& dam, T& dac, T& dkc) {
  if (WRAP) a = twrap(a);      // angles; the distance-based model passes lengths through the same penalty unwrapped
  if (val(a) >= val(am) && val(a) < val(ac)) {
    P D = ac - am;
    T x = (a - ac) * trcp(D);
    T ip = trcp(x + 1.0), im = trcp(x - 1.0);
    T h = ip - im - 2.0;
    T hp = im * im - ip * ip;
    P qD = 0.25 * (kc * D);
    e = qD * D * h;
    da = qD * hp;
    dac = qD * (2.0 * h - (1.0 + x) * hp);
    dam = qD * (x * hp - 2.0 * h);
    dkc = 0.25 * ((D * D) * h);
  } else {
    e = T(0.0); da = T(0.0); dam = T(0.0); dac = T(0.0); dkc = T(0.0);
  }
}

template <class T, class P>
DFX_HD void contact_grad(T kap, P phi1, P phi2, P am, P ac, P kc, ContactGrad<T>& g) {
  T e1, d1, m1, c1, k1, e2, d2, m2, c2, k2;
  contact_one(phi1 - kap, am, ac, kc, e1, d1, m1, c1, k1);
  contact_one(phi2 + kap, am, ac, kc, e2, d2, m2, c2, k2);
  g.e = e1 + e2;
  g.dkap = d2 - d1;
  g.p1 = d1;
  g.p2 = d2;
  g.am = m1 + m2;
  g.ac = c1 + c2;
  g.kc = k1 + k2;
}

// ---------------------------------------------------------------------------------------
// Hessian-vector product by hand (round 6): what the reverse stage needs from a ligament -- dE/d(own DOFs) (value), (H w)_own and the
// epsilon part of dE/d(own node vector) for the direction w = (w_o, w_p) on the six DOFs of the two blocks -- written out instead of pushed
// through the gradient as dual numbers.  Same formulas as bond_grad / contact_grad (energy.py:70-176, 333-361), differentiated once more:
// in the ligament's own coordinates  b = dU + l  (length L, direction bh, angle phi_b), mean rotation tb, relative rotation kap
//     E = ks (L - l0)^2 / 2 + ksh l0^2 gam^2 / 2 + kr kap^2 / 2,   gam = phi_b - tb - angle(l)      (d gam / d tb = -1 exactly)
//     Ldot = bh . bdot,  phidot_b = (b x bdot) / L^2,  gamdot = phidot_b - tbdot,  Sdot = ksh l0^2 gamdot
// so the second derivative costs ~70 multiply-adds where the dual-number evaluation of the whole gradient costs ~300 (the reverse stage
// kernel is bound by instruction issue: profiles/LABNOTES.md, round 4).  The per-ligament gradients (reference vector, stiffnesses,
// contact constants), the spring models and the distance-based contact keep the dual-number path.  Checked against that path on random
// ligaments to 1e-13 (tests/test_physics_hvp.py) and, inside the kernels, by every gradient test of the GPU suite.
// ---------------------------------------------------------------------------------------
struct BondHvp {
  double fx, fy, fth;    // dE/d(x, y, theta) of the own block                    (bond_grad: fx, fy, fth)
  double hx, hy, hth;    // (H w) on the own block's DOFs                          (their epsilon parts)
  double rx, ry;         // epsilon part of dE/d(centroid_node_vector of own node) (rx.e, ry.e)
};

template <int MODEL>
DFX_HD void bond_hvp(const BlockRec<double>& o, const BlockRec<double>& p, double wox, double woy, double woth, double wpx, double wpy, double wpth,
                     double rox, double roy, double rpx, double rpy, double lx, double ly, double l0, double il0, double ks, double ksh, double kr,
                     double sgn, BondHvp& h) {
  static_assert(MODEL == kNonlinear || MODEL == kLinearized, "bond_hvp: the two ligament models");
  const double co = o.ch * o.ch - o.sh * o.sh, so = 2.0 * (o.sh * o.ch);
  const double cp = p.ch * p.ch - p.sh * p.sh, sp = 2.0 * (p.sh * p.ch);
  const double qox = co * rox - so * roy, qoy = so * rox + co * roy;
  const double qpx = cp * rpx - sp * rpy, qpy = sp * rpx + cp * rpy;
  const double dUx = sgn * ((o.x + qox - rox) - (p.x + qpx - rpx));
  const double dUy = sgn * ((o.y + qoy - roy) - (p.y + qpy - rpy));
  // direction: d(dU)/d eps, mean and relative rotation rates
  const double bdx = sgn * ((wox - woth * qoy) - (wpx - wpth * qpy));
  const double bdy = sgn * ((woy + woth * qox) - (wpy + wpth * qpx));
  const double tbd = 0.5 * (woth + wpth);
  const double kapd = sgn * (woth - wpth);
  const double kap = sgn * (o.th - p.th);
  double gbx, gby, gtb, gbxd, gbyd, gtbd;
  if (MODEL == kNonlinear) {
    const double l02 = lx * lx + ly * ly;
    const double cb = o.ch * p.ch - o.sh * p.sh, sb = o.sh * p.ch + o.ch * p.sh;
    const double bx = dUx + lx, by = dUy + ly;
    const double L2 = bx * bx + by * by;
    const double Lb = sqrt(L2);            // (correctly rounded: a lattice at rest has Lb == l0 bit for bit, as in bond_grad)
    const double iLb = trcp(Lb), iL2 = iLb * iLb;
    const double px = cb * lx - sb * ly, py = sb * lx + cb * ly;
    const double gam = fast_atan2(px * by - py * bx, px * bx + py * by);
    const double es = Lb - l0;
    const double kse = ks * es, S = ksh * gam * l02;
    gbx = kse * bx * iLb - S * by * iL2;
    gby = kse * by * iLb + S * bx * iL2;
    gtb = -S;
    const double u = (bx * bdx + by * bdy) * iLb;            // Ldot
    const double gamd = (bx * bdy - by * bdx) * iL2 - tbd;
    const double Sd = ksh * l02 * gamd;
    const double ui = u * iLb;
    const double a1 = ks * u - kse * ui;                     // d(kse / L) * L
    const double a2 = Sd - 2.0 * (S * ui);                   // d(S / L^2) * L^2
    gbxd = iLb * (a1 * bx + kse * bdx) - iL2 * (a2 * by + S * bdy);
    gbyd = iLb * (a1 * by + kse * bdy) + iL2 * (a2 * bx + S * bdx);
    gtbd = -Sd;
  } else {
    const double tb = 0.5 * (o.th + p.th);
    const double es = (dUx * lx + dUy * ly) * il0;
    const double esh = (lx * dUy - ly * dUx) * il0 - tb * l0;
    const double kse = ks * es, kshe = ksh * esh;
    gbx = (kse * lx - kshe * ly) * il0;
    gby = (kse * ly + kshe * lx) * il0;
    gtb = -(kshe * l0);
    const double ksed = ks * ((bdx * lx + bdy * ly) * il0);
    const double kshed = ksh * ((lx * bdy - ly * bdx) * il0 - tbd * l0);
    gbxd = (ksed * lx - kshed * ly) * il0;
    gbyd = (ksed * ly + kshed * lx) * il0;
    gtbd = -(kshed * l0);
  }
  h.fx = sgn * gbx;
  h.fy = sgn * gby;
  h.fth = sgn * (gby * qox - gbx * qoy) + 0.5 * gtb + sgn * (kr * kap);
  h.hx = sgn * gbxd;
  h.hy = sgn * gbyd;
  // d(qo)/d eps = woth * (-qoy, qox)
  h.hth = sgn * (gbyd * qox - gbxd * qoy - woth * (gby * qoy + gbx * qox)) + 0.5 * gtbd + sgn * (kr * kapd);
  // d(co, so)/d eps = woth * (-so, co)
  h.rx = sgn * (co * gbxd + so * gbyd - gbxd + woth * (co * gby - so * gbx));
  h.ry = sgn * (co * gbyd - so * gbxd - gbyd - woth * (so * gby + co * gbx));
}

// Angle-based contact, the same way: dE/dkap (value), its epsilon part for kapdot, and the epsilon parts of dE/dphi1, dE/dphi2.
//   e'(a) = qD hp(x), x = (a - ac) / D, qD = kc D / 4   =>   e''(a) = kc (ip^3 - im^3) / 2,  ip = 1 / (x + 1), im = 1 / (x - 1)
DFX_HD void contact_hvp_one(double a, double am, double ac, double kc, double& da, double& dda) {
  a = twrap(a);
  da = 0.0; dda = 0.0;
  if (a >= am && a < ac) {
    const double D = ac - am;
    const double x = (a - ac) * trcp(D);
    const double ip = trcp(x + 1.0), im = trcp(x - 1.0);
    da = 0.25 * (kc * D) * (im * im - ip * ip);
    dda = 0.5 * kc * (ip * ip * ip - im * im * im);
  }
}
DFX_HD void contact_hvp(double kap, double kapd, double phi1, double phi2, double am, double ac, double kc, double& dkap, double& dkap_e, double& p1_e,
                        double& p2_e) {
  double d1, dd1, d2, dd2;
  contact_hvp_one(phi1 - kap, am, ac, kc, d1, dd1);
  contact_hvp_one(phi2 + kap, am, ac, kc, d2, dd2);
  dkap = d2 - d1;
  p1_e = -(dd1 * kapd);
  p2_e = dd2 * kapd;
  dkap_e = p2_e - p1_e;
}

// ---------------------------------------------------------------------------------------
// distance-based contact (energy.py:222-330; build_contact_energy(angle_based=False) :364-407)
// ---------------------------------------------------------------------------------------
// point_to_edge_distance (energy.py:222-251) in closest-point form: with t = (p - a).(b - a) / |b - a|^2 clamped to [0, 1] the
// three branches of the reference are d = |p - (a + t (b - a))| (inside: sqrt(|p-a|^2 - t^2 |b-a|^2) is the same number, written
// without the cancellation), and in every branch   d d/dp = n,  d d/da = -(1 - t) n,  d d/db = -t n,   n = (p - closest) / d.
template <class T>
DFX_HD void point_segment(T px, T py, T ax, T ay, T bx, T by, T& d, T& nx, T& ny, T& t) {
  T ex = bx - ax, ey = by - ay, rx = px - ax, ry = py - ay;
  t = (rx * ex + ry * ey) * trcp(ex * ex + ey * ey);
  if (val(t) < 0.0) t = T(0.0);
  else if (val(t) > 1.0) t = T(1.0);
  T cx = rx - t * ex, cy = ry - t * ey;
  d = tsqrt(cx * cx + cy * cy);
  T id = trcp(d);
  nx = cx * id;
  ny = cy * id;
}

// edges_distance (energy.py:254-276): the smallest of the four end-point-to-other-edge distances, candidates in the reference's
// order (B0 -> A, B1 -> A, A0 -> B, A1 -> B; the first smallest wins), and its gradient w.r.t. the four end points
// (g[0..3] = A0, A1, B0, B1; x, y).  Selection by value; the selected candidate is evaluated in T.
template <class T>
DFX_HD void edges_distance(const T (&A0)[2], const T (&A1)[2], const T (&B0)[2], const T (&B1)[2], T& d, T (&g)[4][2]) {
  const T* pts[4] = {B0, B1, A0, A1};
  const T* e0[4] = {A0, A0, B0, B0};
  const T* e1[4] = {A1, A1, B1, B1};
  const int pi[4] = {2, 3, 0, 1}, ei0[4] = {0, 0, 2, 2}, ei1[4] = {1, 1, 3, 3};
  int best = 0;
  double dbest = 0.0;
  for (int k = 0; k < 4; ++k) {
    double dk, nx, ny, t;
    point_segment<double>(val(pts[k][0]), val(pts[k][1]), val(e0[k][0]), val(e0[k][1]), val(e1[k][0]), val(e1[k][1]), dk, nx, ny, t);
    if (k == 0 || dk < dbest) { best = k; dbest = dk; }
  }
  T nx, ny, t;
  point_segment<T>(pts[best][0], pts[best][1], e0[best][0], e0[best][1], e1[best][0], e1[best][1], d, nx, ny, t);
  for (int k = 0; k < 4; ++k) { g[k][0] = T(0.0); g[k][1] = T(0.0); }
  g[pi[best]][0] = nx; g[pi[best]][1] = ny;
  T w0 = 1.0 - t;
  g[ei0[best]][0] = -(w0 * nx); g[ei0[best]][1] = -(w0 * ny);
  g[ei1[best]][0] = -(t * nx); g[ei1[best]][1] = -(t * ny);
}

// One ligament, seen from one of its ends: derivatives of the two void-edge-distance penalties w.r.t. everything the OWN block
// owns.  Node order in ro / rp: the bonded node, its next node, its previous node on the block (geometry.py:192-202 numbering).
template <class T>
struct DistContactGrad {
  T fx, fy, fth;     // dE/d(x, y, theta) of the own block
  T r[3][2];         // dE/d(centroid_node_vector) of the bonded node [0], its next [1] and its previous [2] node
  T cx, cy;          // dE/d(block_centroid) of the own block
  T am, ac, kc;      // dE/d(min, cutoff, k_contact)   (lengths here: utils.py:101)
  T e;
};

template <class T, class P>
DFX_HD void distance_contact_grad(const BlockRec<T>& o, const BlockRec<T>& p, P cox, P coy, P cpx, P cpy, const P (&ro)[3][2],
                                  const P (&rp)[3][2], int own_is_end2, P am, P ac, P kc, DistContactGrad<T>& g) {
  T co = o.ch * o.ch - o.sh * o.sh, so = 2.0 * (o.sh * o.ch);
  T cp = p.ch * p.ch - p.sh * p.sh, sp = 2.0 * (p.sh * p.ch);
  T qo[3][2], O[3][2], Q[3][2];
  for (int k = 0; k < 3; ++k) {           // current node positions: centroid + displacement + rotated node vector (energy.py:397-404)
    qo[k][0] = co * ro[k][0] - so * ro[k][1];
    qo[k][1] = so * ro[k][0] + co * ro[k][1];
    O[k][0] = o.x + qo[k][0] + cox;
    O[k][1] = o.y + qo[k][1] + coy;
    Q[k][0] = p.x + (cp * rp[k][0] - sp * rp[k][1]) + cpx;
    Q[k][1] = p.y + (sp * rp[k][0] + cp * rp[k][1]) + cpy;
  }
  // energy.py:301-326: block 1 holds node n1, block 2 node n2;  d1 = dist(edge(p1, p1_next), edge(p2, p2_prev)),
  //                                                             d2 = dist(edge(p1, p1_prev), edge(p2, p2_next))
  T G[3][2];                               // dE/d(own node positions)
  for (int k = 0; k < 3; ++k) { G[k][0] = T(0.0); G[k][1] = T(0.0); }
  g.e = T(0.0); g.am = T(0.0); g.ac = T(0.0); g.kc = T(0.0);
  for (int j = 0; j < 2; ++j) {
    // which own / partner neighbour node closes the edge: (next, prev) for d1 and (prev, next) for d2 as seen from block 1
    const int n1 = j == 0 ? 1 : 2, n2 = j == 0 ? 2 : 1;
    const int own_nb = own_is_end2 ? n2 : n1, par_nb = own_is_end2 ? n1 : n2;
    T d, gg[4][2];
    if (!own_is_end2) edges_distance<T>(O[0], O[own_nb], Q[0], Q[par_nb], d, gg);      // A = own edge, B = partner edge
    else edges_distance<T>(Q[0], Q[par_nb], O[0], O[own_nb], d, gg);                   // A = partner edge, B = own edge
    T e, da, dam, dac, dkc;
    contact_one<T, P, false>(d, am, ac, kc, e, da, dam, dac, dkc);
    g.e = g.e + e; g.am = g.am + dam; g.ac = g.ac + dac; g.kc = g.kc + dkc;
    const int base = own_is_end2 ? 2 : 0;  // where the own edge's end points sit in gg
    G[0][0] = G[0][0] + da * gg[base][0];      G[0][1] = G[0][1] + da * gg[base][1];
    G[own_nb][0] = G[own_nb][0] + da * gg[base + 1][0];  G[own_nb][1] = G[own_nb][1] + da * gg[base + 1][1];
  }
  g.fx = T(0.0); g.fy = T(0.0); g.fth = T(0.0);
  for (int k = 0; k < 3; ++k) {
    g.fx = g.fx + G[k][0];
    g.fy = g.fy + G[k][1];
    g.fth = g.fth + (G[k][1] * qo[k][0] - G[k][0] * qo[k][1]);     // dX_k/dtheta = (-q_y, q_x)
    g.r[k][0] = co * G[k][0] + so * G[k][1];                      // R^T G
    g.r[k][1] = co * G[k][1] - so * G[k][0];
  }
  g.cx = g.fx; g.cy = g.fy;
}

// ---------------------------------------------------------------------------------------
// closed library of scalar time functions; value, d/dt and d/dparams
// ---------------------------------------------------------------------------------------
enum TimeFnType {
  kFnZero = 0,
  kFnPulse = 1,       // p = (A, f, t_d):  A/2 (1 - cos 2 pi f tau) on 0 < tau < 1/f, tau = t - t_d
  kFnHarmonic = 2,    // same, on tau > 0
  kFnRamp = 3,        // p = (A, r):       A * (t r  if t < 1/r else 1)
  kFnSech2Tanh = 4,   // p = (A, s):       2A/s^2 sech^2(t/s - 3) tanh(3 - t/s)
  kFnConstant = 5,    // p = (A)
  kFnRampCap = 6,     // p = (L, r, cap):  L * min(t r, cap)   (static compression: quads_kinetic_energy_static_tuning.py:176-182)
  kFnTable = 7        // p = (A, t_d): A * piecewise-linear table(t - t_d), end values held
};
constexpr int kMaxFnParams = 5;

struct TimeFn {
  int type;
  int n_tab;            // kFnTable: breakpoints
  double p[kMaxFnParams];
  const double* tab;    // kFnTable: n_tab times then n_tab values (device memory in the HIP engine)
};

// g = value, gt = dg/dt, gp[i] = dg/dp[i]
DFX_HD void eval_time_fn(const TimeFn& f, double t, double& g, double& gt, double* gp) {
  for (int i = 0; i < kMaxFnParams; ++i) gp[i] = 0.0;
  g = 0.0;
  gt = 0.0;
  switch (f.type) {
    case kFnPulse:
    case kFnHarmonic: {
      double A = f.p[0], fr = f.p[1], tau = t - f.p[2];
      bool on = (tau > 0.0) && (f.type == kFnHarmonic || tau * fr < 1.0);
      if (on) {
        double ph = kTwoPi * fr * tau;
        double s = sin(ph), c = cos(ph);
        g = 0.5 * A * (1.0 - c);
        gt = A * kPi * fr * s;
        gp[0] = 0.5 * (1.0 - c);
        gp[1] = A * kPi * tau * s;
        gp[2] = -gt;
      }
    } break;
    case kFnRamp:
    case kFnRampCap: {
      // A min(t r, cap): the plain ramp has cap = 1; the capped one is the reference's jnp.where(t < cap / r, t * r, cap) scaled by
      // L = A (static compression) -- the ramp branch is taken where t r < cap
      const double A = f.p[0], r = f.p[1], cap = f.type == kFnRampCap ? f.p[2] : 1.0;
      if (t * r < cap) { g = A * t * r; gt = A * r; gp[0] = t * r; gp[1] = A * t; }
      else { g = A * cap; gp[0] = cap; if (f.type == kFnRampCap) gp[2] = A; }
    } break;
    case kFnSech2Tanh: {
      double A = f.p[0], s = f.p[1];
      double z = t / s - 3.0;
      double th = tanh(z), se2 = 1.0 - th * th;
      double pre = 2.0 * A / (s * s);
      g = -pre * se2 * th;                       // tanh(3 - t/s) = -tanh(z)
      double dgdz = -pre * se2 * (1.0 - 3.0 * th * th);
      gt = dgdz / s;
      gp[0] = g / A;
      gp[1] = -2.0 * g / s + dgdz * (-t / (s * s));
    } break;
    case kFnConstant:
      g = f.p[0];
      gp[0] = 1.0;
      break;
    case kFnTable: {
      const double A = f.p[0], tau = t - f.p[1];
      const double* T = f.tab;
      const double* Y = f.tab + f.n_tab;
      double y, dy = 0.0;
      if (tau <= T[0]) y = Y[0];
      else if (tau >= T[f.n_tab - 1]) y = Y[f.n_tab - 1];
      else {
        int lo = 0, hi = f.n_tab - 1;            // T[lo] <= tau < T[hi]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (T[mid] <= tau) lo = mid; else hi = mid; }
        dy = (Y[hi] - Y[lo]) / (T[hi] - T[lo]);
        y = Y[lo] + dy * (tau - T[lo]);
      }
      g = A * y; gt = A * dy;
      gp[0] = y; gp[1] = -gt;
    } break;
    default:
      break;
  }
}

// ---------------------------------------------------------------------------------------
// explicit Runge-Kutta tableaux in "acceleration form"
//   k_j = (V_j, A_j);  V_i = v_n + h sum_j a_ij A_j;  Q_i = q_n + h c_i v_n + h^2 sum_l aa_il A_l
//   with aa = a * a  (so only accelerations are stored between stages).
// ---------------------------------------------------------------------------------------
constexpr int kMaxStages = 7;
struct Tableau {
  int s;                                 // stages per step (dopri5 fixed-step: 6, rk4: 4)
  double c[kMaxStages + 1];              // stage times; c[s] = 1 (the step end)
  double a[kMaxStages + 1][kMaxStages];  // rows 0..s-1: stage coefficients, row s: solution weights b
  double aa[kMaxStages + 1][kMaxStages]; // (a*a) rows incl. the b row:  sum_j a_ij a_jl
};

inline Tableau make_tableau(int s, const double* c, const double* a /* (s+1) x s, last row = b */) {
  Tableau t;
  t.s = s;
  for (int i = 0; i <= kMaxStages; ++i) {
    t.c[i] = 0.0;
    for (int j = 0; j < kMaxStages; ++j) { t.a[i][j] = 0.0; t.aa[i][j] = 0.0; }
  }
  for (int i = 0; i <= s; ++i) {
    t.c[i] = (i < s) ? c[i] : 1.0;
    for (int j = 0; j < s; ++j) t.a[i][j] = a[i * s + j];
  }
  for (int i = 0; i <= s; ++i)
    for (int l = 0; l < s; ++l) {
      double acc = 0.0;
      for (int j = 0; j < s; ++j) acc += t.a[i][j] * t.a[j][l];
      t.aa[i][l] = acc;
    }
  return t;
}

inline Tableau tableau_dopri5() {
  // Dormand-Prince 5(4), fixed step: 6 stages, solution weights = row 7 of the tableau (FSAL)
  static const double c[6] = {0.0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0};
  static const double a[7 * 6] = {
      0, 0, 0, 0, 0, 0,
      1.0 / 5, 0, 0, 0, 0, 0,
      3.0 / 40, 9.0 / 40, 0, 0, 0, 0,
      44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0, 0,
      19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0, 0,
      9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656, 0,
      35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
  return make_tableau(6, c, a);
}

inline Tableau tableau_rk4() {
  static const double c[4] = {0.0, 0.5, 0.5, 1.0};
  static const double a[5 * 4] = {0, 0, 0, 0, 0.5, 0, 0, 0, 0, 0.5, 0, 0, 0, 0, 1.0, 0,
                                  1.0 / 6, 1.0 / 3, 1.0 / 3, 1.0 / 6};
  return make_tableau(4, c, a);
}


// Dormand-Prince 5(4) with embedded error and dense output, as used by jax.experimental.ode (7 rows incl. FSAL).
struct Dopri {
  double c[7];        // stage times of k_0..k_6 (k_6 is evaluated at the new state, t + h)
  double a[7][7];     // a[i][j]: stage i uses k_j (row 6 = 5th-order solution weights)
  double e[7];        // error weights
  double cm[7];       // mid-point weights of the dense output
  // acceleration form (k_j = (V_j, A_j), V_j = v_n + h sum_l a[j][l] A_l):
  double aa[7][7];    // sum_j a[i][j] a[j][l]
  double ee[7];       // sum_j e[j] a[j][l]      (position error = h^2 sum_l ee[l] A_l, since sum_j e[j] = 0)
  double cma[7];      // sum_j cm[j] a[j][l]     (q_mid = q_n + h/2 v_n + h^2 sum_l cma[l] A_l)
};

inline Dopri make_dopri() {
  Dopri d;
  const double c[7] = {0.0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0, 1.0};
  const double a[7][7] = {
      {0, 0, 0, 0, 0, 0, 0},
      {1.0 / 5, 0, 0, 0, 0, 0, 0},
      {3.0 / 40, 9.0 / 40, 0, 0, 0, 0, 0},
      {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0, 0, 0},
      {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0, 0, 0},
      {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656, 0, 0},
      {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84, 0}};
  const double e[7] = {35.0 / 384 - 1951.0 / 21600, 0, 500.0 / 1113 - 22642.0 / 50085, 125.0 / 192 - 451.0 / 720,
                       -2187.0 / 6784 - -12231.0 / 42400, 11.0 / 84 - 649.0 / 6300, -1.0 / 60};
  const double cm[7] = {6025192743.0 / 30085553152.0 / 2, 0, 51252292925.0 / 65400821598.0 / 2,
                        -2691868925.0 / 45128329728.0 / 2, 187940372067.0 / 1594534317056.0 / 2,
                        -1776094331.0 / 19743644256.0 / 2, 11237099.0 / 235043384.0 / 2};
  for (int i = 0; i < 7; ++i) {
    d.c[i] = c[i]; d.e[i] = e[i]; d.cm[i] = cm[i];
    for (int j = 0; j < 7; ++j) d.a[i][j] = a[i][j];
  }
  for (int l = 0; l < 7; ++l) {
    d.ee[l] = 0.0; d.cma[l] = 0.0;
    for (int j = 0; j < 7; ++j) { d.ee[l] += e[j] * a[j][l]; d.cma[l] += cm[j] * a[j][l]; }
    for (int i = 0; i < 7; ++i) {
      d.aa[i][l] = 0.0;
      for (int j = 0; j < 7; ++j) d.aa[i][l] += a[i][j] * a[j][l];
    }
  }
  return d;
}

// step-size controller of jax.experimental.ode.optimal_step_size (safety 0.9, ifactor 10, dfactor 0.2, order 5)
DFX_HD double dopri_next_step(double h, double ratio) {
  if (ratio == 0.0) return h * 10.0;
  const double dfac = ratio < 1.0 ? 1.0 : 0.2;
  double f = 0.9 * pow(ratio, -0.2);
  f = f > dfac ? f : dfac;
  f = f < 10.0 ? f : 10.0;
  return h * f;
}

// quartic through (y0, y1, y_mid, dy0, dy1) evaluated at relative time r in [0,1]  (jax: fit_4th_order_polynomial + polyval)
DFX_HD double dopri_dense(double y0, double y1, double ymid, double dy0, double dy1, double dt, double r) {
  const double a = -2.0 * dt * dy0 + 2.0 * dt * dy1 - 8.0 * y0 - 8.0 * y1 + 16.0 * ymid;
  const double b = 5.0 * dt * dy0 - 3.0 * dt * dy1 + 18.0 * y0 + 14.0 * y1 - 32.0 * ymid;
  const double c = -4.0 * dt * dy0 + dt * dy1 - 11.0 * y0 - 5.0 * y1 + 16.0 * ymid;
  return (((a * r + b) * r + c) * r + dt * dy0) * r + y0;
}


// The same dense output as weights of the stage slopes: out(r) = y0 + dt sum_j B_j(r) k_j, j = 0..6 (k_6 = f(y1), the FSAL slope), with
// y1 = y0 + dt sum_j b_j k_j and y_mid = y0 + dt sum_j cm_j k_j put into the quartic above.  What the reverse sweep of an adaptive solve
// needs: the cotangent g of an output inside a step adds dt B_j(r) g to Kbar_j and g to lambda at the start of the step.
DFX_HD void dopri_dense_weights(double r, const double* b /* 7: solution weights */, const double* cm /* 7 */, double* B /* 7 */) {
  const double r2 = r * r, r3 = r2 * r, r4 = r3 * r;
  const double w1 = -8.0 * r4 + 14.0 * r3 - 5.0 * r2, wm = 16.0 * r4 - 32.0 * r3 + 16.0 * r2;
  for (int j = 0; j < 7; ++j) B[j] = w1 * b[j] + wm * cm[j];
  B[0] += -2.0 * r4 + 5.0 * r3 - 4.0 * r2 + r;
  B[6] += 2.0 * r4 - 3.0 * r3 + r2;
}

}  // namespace dfx
