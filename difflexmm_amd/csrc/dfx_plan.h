// dfx_plan.h -- host-side problem plan shared by the HIP engine (dfx_engine.hip) and the CPU port
// (oracle/cpu/dfx_cpu.cpp): turns the reference's bond list into the slot-indexed tables the
// kernels read, packs ControlParams into that layout and scatters slot gradients back.
//
// Layout ("gather form", one lane per (block, node slot), 4 slots per block):
//   slot = 4*block + local_node          (kagome blocks use slots 0..2, slot 3 is padding)
//   slot_info[slot] = 2*partner_slot + own_is_end2   or -1 when the node carries no ligament
// Every ligament is therefore evaluated by both of its ends; each end keeps only the derivatives
// of quantities it owns, so forces and parameter gradients are accumulated without atomics and
// bit-reproducibly.  Ligament parameters are duplicated to both slots (coalesced loads).
#pragma once
#include "dfx_hostpar.h"
#include <stdint.h>
#include <string.h>

#include <string>
#include <map>
#include <vector>

#include "../../include/dfx.h"
#include "dfx_physics.h"

namespace dfx {

constexpr int kSlots = 4;        // lanes per block
constexpr int kRec = 8;          // doubles per block stage record: x y th cos(th/2) sin(th/2) vx vy vth
constexpr int kSlotParams = 9;   // r(2) l0(2) k(3) phi(2)

struct Plan {
  int n_blocks = 0, n_npb = 0, n_bonds = 0, batch = 1, n_slots = 0;
  int model = 0, contact = 0, n_fns = 0, n_special = 0;
  int fn_type[DFX_MAX_FNS] = {0, 0};
  std::vector<double> fn_table[DFX_MAX_FNS];        // DFX_FN_TABLE: times then values
  const double* fn_table_ptr[DFX_MAX_FNS] = {nullptr, nullptr};   // where the kernels read it (device copy in the HIP engine)
  Tableau tab;
  std::vector<int32_t> slot_info;      // n_slots
  std::vector<int32_t> slot_bond;      // n_slots, bond id or -1
  int pred_delta[4] = {0, 0, 0, 0};    // most frequent (partner slot - own slot) per node slot: the kernels gather from the guess
                                       // while slot_info is still in flight (regular lattices: right everywhere but the rim)
  std::vector<int32_t> block_special;  // n_blocks, index into special or -1
  std::vector<dfx_special> special;
  // Nodes that carry MORE than one ligament (jax_md.smap.bond takes any bond list, energy.py:179-197; none of the reference's
  // lattices has such nodes): the first ligament of a node lives in slot_info as always, the others in per-slot lists in CSR form.
  // A lane loops over its node's extra ligaments after the first one; lattices without them never enter the loop.
  int n_ovf = 0;
  std::vector<int32_t> ovf_ptr;        // n_slots + 1 offsets into ovf_info (empty when n_ovf == 0)
  std::vector<int32_t> ovf_info;       // 2 * partner_slot + own_is_end2
  std::vector<int32_t> ovf_bond;       // bond id
};
constexpr int kOvfParams = 8;          // lx ly k_stretch k_shear k_rot phi1 phi2 pad   (per extra ligament end)
constexpr int kOvfGrads = 12;          // CPU layout of a slot's bond part: - - l0(2) k(3) phi(2) contact(3)

inline int build_plan(const dfx_problem* p, Plan& pl, std::string& err) {
  if (!p || p->n_blocks <= 0 || (p->n_npb != 3 && p->n_npb != 4)) { err = "invalid problem: n_blocks/n_npb"; return 1; }
  if (p->batch <= 0) { err = "invalid problem: batch must be >= 1"; return 1; }
  if (p->n_fns < 0 || p->n_fns > DFX_MAX_FNS) { err = "invalid problem: n_fns out of range"; return 1; }
  if (p->bond_model < DFX_BOND_LINEARIZED || p->bond_model > DFX_BOND_STRETCH_TORSION) { err = "invalid bond_model"; return 1; }
  if (p->contact < DFX_CONTACT_NONE || p->contact > DFX_CONTACT_DISTANCE) { err = "invalid contact model"; return 1; }
  pl.n_blocks = p->n_blocks; pl.n_npb = p->n_npb; pl.n_bonds = p->n_bonds; pl.batch = p->batch;
  pl.n_slots = p->n_blocks * kSlots;
  pl.model = p->bond_model; pl.contact = p->contact; pl.n_fns = p->n_fns; pl.n_special = p->n_special;
  for (int i = 0; i < DFX_MAX_FNS; ++i) pl.fn_type[i] = i < p->n_fns ? p->fn_type[i] : 0;
  for (int i = 0; i < p->n_fns; ++i) {
    pl.fn_table[i].clear();
    if (pl.fn_type[i] != DFX_FN_TABLE) continue;
    const int n = p->fn_table_n[i];
    if (n < 2 || !p->fn_table[i]) { err = "create: a table time function needs >= 2 breakpoints"; return 1; }
    pl.fn_table[i].assign(p->fn_table[i], p->fn_table[i] + 2 * n);
    for (int k = 0; k + 1 < n; ++k)
      if (!(pl.fn_table[i][k + 1] > pl.fn_table[i][k])) { err = "create: table times must be strictly increasing"; return 1; }
    pl.fn_table_ptr[i] = pl.fn_table[i].data();
  }
  if (p->tableau == DFX_TABLEAU_DOPRI5) pl.tab = tableau_dopri5();
  else if (p->tableau == DFX_TABLEAU_RK4) pl.tab = tableau_rk4();
  else { err = "invalid tableau"; return 1; }
  pl.slot_info.assign(pl.n_slots, -1);
  pl.slot_bond.assign(pl.n_slots, -1);
  const int n_nodes = p->n_blocks * p->n_npb;
  std::vector<std::vector<std::pair<int32_t, int32_t>>> extra;     // per slot: (info, bond) of the ligaments after the first
  for (int b = 0; b < p->n_bonds; ++b) {
    int n1 = p->bonds[2 * b], n2 = p->bonds[2 * b + 1];
    if (n1 < 0 || n2 < 0 || n1 >= n_nodes || n2 >= n_nodes || n1 == n2) { err = "bond with invalid node id"; return 1; }
    int s1 = (n1 / p->n_npb) * kSlots + n1 % p->n_npb;
    int s2 = (n2 / p->n_npb) * kSlots + n2 % p->n_npb;
    if (n1 / p->n_npb == n2 / p->n_npb) { err = "bond joins two nodes of one block"; return 1; }
    const int ends[2][2] = {{s1, 2 * s2 + 0}, {s2, 2 * s1 + 1}};
    for (auto& e : ends) {
      if (pl.slot_info[e[0]] == -1) { pl.slot_info[e[0]] = e[1]; pl.slot_bond[e[0]] = b; continue; }
      if (p->contact == DFX_CONTACT_DISTANCE) { err = "node with more than one ligament: not supported with the distance-based contact"; return 1; }
      if (extra.empty()) extra.resize(pl.n_slots);
      extra[e[0]].push_back({e[1], b});
    }
  }
  pl.n_ovf = 0; pl.ovf_ptr.clear(); pl.ovf_info.clear(); pl.ovf_bond.clear();
  if (!extra.empty()) {
    pl.ovf_ptr.assign(pl.n_slots + 1, 0);
    for (int s = 0; s < pl.n_slots; ++s) {
      pl.ovf_ptr[s] = (int32_t)pl.ovf_info.size();
      for (auto& e : extra[s]) { pl.ovf_info.push_back(e.first); pl.ovf_bond.push_back(e.second); }
    }
    pl.ovf_ptr[pl.n_slots] = (int32_t)pl.ovf_info.size();
    pl.n_ovf = (int)pl.ovf_info.size();
  }
  for (int k = 0; k < kSlots; ++k) {
    std::map<int, int> votes;
    for (int s = k; s < pl.n_slots; s += kSlots)
      if (pl.slot_info[s] >= 0) ++votes[(pl.slot_info[s] >> 1) - s];
    int best = 0, n_best = 0;
    for (auto& kv : votes) if (kv.second > n_best) { best = kv.first; n_best = kv.second; }
    pl.pred_delta[k] = best;
  }
  pl.block_special.assign(p->n_blocks, -1);
  pl.special.assign(p->special, p->special + p->n_special);
  for (int i = 0; i < p->n_special; ++i) {
    int blk = p->special[i].block;
    if (blk < 0 || blk >= p->n_blocks) { err = "special block id out of range"; return 1; }
    if (pl.block_special[blk] != -1) { err = "block listed twice in special"; return 1; }
    pl.block_special[blk] = i;
  }
  return 0;
}

// Packed per-member parameter image (host copy; the HIP engine uploads it verbatim).
struct PackedParams {
  std::vector<double> slot;     // batch * n_slots * kSlotParams
  std::vector<double> inv_m;    // batch * n_blocks * 3
  std::vector<double> damping;  // batch * n_blocks * 3
  std::vector<double> contact;  // batch * 3
  std::vector<double> centroid; // batch * n_blocks * 2 (distance-based contact)
  std::vector<TimeFn> fns;      // batch * DFX_MAX_FNS
  // GPU image (structure of arrays, 16-byte rows so every lane issues aligned dwordx4 loads):
  std::vector<double> p_r;      // batch * n_slots * 2 : own node vector
  std::vector<double> p_l;      // batch * n_slots * 2 : reference vector (oriented node1 -> node2)
  std::vector<double> p_k;      // batch * n_slots * 4 : k_stretch, k_shear, k_rot, 0
  std::vector<double> p_phi;    // batch * n_slots * 2 : the two undeformed void angles (phi1, phi2) of the slot's first ligament
  std::vector<double> ovf;      // batch * n_ovf * kOvfParams : parameters of the extra ligaments (CPU port and GPU image alike)
  std::vector<double> cst;      // batch * 16 (first 9 used) : min_angle, cutoff_angle, k_contact, k_stretch, k_shear, k_rot (if uniform), 0, 0
  bool k_uniform = true;        // every ligament of a member has the same three stiffnesses
  // dictionary compression of per-slot constants that take few distinct values (lattices have 2-3 reference vectors):
  std::vector<uint8_t> l_idx;   // batch * n_slots : index into l_dict
  std::vector<double> l_dict;   // batch * 256 * 4 : lx, ly, |l0|, 1/|l0|
  bool l_dict_ok = true;        // <= 256 distinct reference vectors in every member
  int n_dict_max = 0;           // the largest dictionary of any member (lattices have 2-3 entries: small ones are looked up in LDS)
  bool damping_uniform = true;  // the three per-DOF damping coefficients are the same for every block of a member (cst[6..8])
};

// ControlParams arrays -> the packed images.  The members of an ensemble are independent: a handful of host threads, one contiguous chunk
// of members each (dfx_hostpar.h), every entry of every image written (nothing is zero-filled first: at 16 x 128 x 128 the images are
// 150 MB, and filling them twice was most of the 12 ms this function took before round 6).
inline int pack_params(const Plan& pl, const dfx_params* q, PackedParams& out, std::string& err, bool gpu_image = true) {
  if (!q || !q->centroid_node_vectors || !q->reference_vector || !q->k_bond || !q->inertia) {
    err = "set_params: centroid_node_vectors, reference_vector, k_bond and inertia are required"; return 1;
  }
  if (pl.contact == DFX_CONTACT_ANGLE && (!q->void_angle0 || !q->contact)) { err = "set_params: contact model needs void_angle0 and contact"; return 1; }
  if (pl.contact == DFX_CONTACT_DISTANCE && (!q->block_centroids || !q->contact)) { err = "set_params: distance-based contact needs block_centroids and contact"; return 1; }
  if (pl.n_fns && !q->fn_params) { err = "set_params: fn_params required"; return 1; }
  const int B = pl.batch, NS = pl.n_slots, NB = pl.n_blocks;
  if (!gpu_image) out.slot.resize((size_t)B * NS * kSlotParams);      // (the GPU image is built from a per-thread scratch copy of one member's)
  out.inv_m.resize((size_t)B * NB * 3);
  out.damping.resize((size_t)B * NB * 3);
  out.contact.assign((size_t)B * 3, 0.0);
  if (q->block_centroids) out.centroid.assign(q->block_centroids, q->block_centroids + (size_t)B * NB * 2);
  else out.centroid.assign((size_t)B * NB * 2, 0.0);
  out.fns.assign((size_t)B * DFX_MAX_FNS, TimeFn{0, 0, {0, 0, 0, 0, 0}});
  if (pl.n_ovf) out.ovf.resize((size_t)B * pl.n_ovf * kOvfParams);
  if (gpu_image) {
    out.p_r.resize((size_t)B * NS * 2); out.p_l.resize((size_t)B * NS * 2);
    out.p_k.resize((size_t)B * NS * 4); out.p_phi.resize((size_t)B * NS * 2);
    out.cst.assign((size_t)B * 16, 0.0);
    out.l_idx.resize((size_t)B * NS); out.l_dict.assign((size_t)B * 1024, 0.0);
  }
  struct Flags { bool k_uniform = true, l_dict_ok = true, damping_uniform = true; int n_dict = 0, bad = 0; };
  std::vector<Flags> flags(B);
  // the nine per-slot values (r, l0, k, phi) of slot (b, k) of member m, straight from the ControlParams arrays
  auto slot_values = [&](int m, int b, int k, double* s) {
    const double* cnv = q->centroid_node_vectors + (size_t)m * NB * pl.n_npb * 2;
    const double* l0 = q->reference_vector + (size_t)m * pl.n_bonds * 2;
    const double* kb = q->k_bond + (size_t)m * pl.n_bonds * 3;
    const double* ph = q->void_angle0 ? q->void_angle0 + (size_t)m * pl.n_bonds * 2 : nullptr;
    for (int i = 0; i < kSlotParams; ++i) s[i] = 0.0;
    if (k >= pl.n_npb) return;
    s[0] = cnv[(b * pl.n_npb + k) * 2];
    s[1] = cnv[(b * pl.n_npb + k) * 2 + 1];
    const int bond = pl.slot_bond[b * kSlots + k];
    if (bond >= 0) {
      s[2] = l0[2 * bond]; s[3] = l0[2 * bond + 1];
      s[4] = kb[3 * bond]; s[5] = kb[3 * bond + 1]; s[6] = kb[3 * bond + 2];
      if (ph) { s[7] = ph[2 * bond]; s[8] = ph[2 * bond + 1]; }
    }
  };
  // GPU image, the two arrays that are uploaded only when a member's ligaments do NOT share their stiffnesses / have > 256 distinct reference
  // vectors: filled in a second pass when that turns out to be so (they are 3 of the 5 MB a member's images take)
  auto fill_l_k = [&](int m, bool want_l, bool want_k) {
    double s[kSlotParams];
    for (int s_ = 0; s_ < NS; ++s_) {
      slot_values(m, s_ / kSlots, s_ % kSlots, s);
      const bool lig = pl.slot_info[s_] >= 0;
      if (want_l) { double* l = out.p_l.data() + ((size_t)m * NS + s_) * 2; l[0] = lig ? s[2] : 0.0; l[1] = lig ? s[3] : 0.0; }
      if (want_k) { double* k = out.p_k.data() + ((size_t)m * NS + s_) * 4; k[0] = lig ? s[4] : 0.0; k[1] = lig ? s[5] : 0.0; k[2] = lig ? s[6] : 0.0; k[3] = 0.0; }
    }
  };
  auto member = [&](int m) {
    Flags& F = flags[m];
    const double* l0 = q->reference_vector + (size_t)m * pl.n_bonds * 2;
    const double* kb = q->k_bond + (size_t)m * pl.n_bonds * 3;
    const double* ph = q->void_angle0 ? q->void_angle0 + (size_t)m * pl.n_bonds * 2 : nullptr;
    if (!gpu_image) {
      double* sp = out.slot.data() + (size_t)m * NS * kSlotParams;
      for (int b = 0; b < NB; ++b)
        for (int k = 0; k < kSlots; ++k) slot_values(m, b, k, sp + (size_t)(b * kSlots + k) * kSlotParams);
    }
    for (int e = 0; e < pl.n_ovf; ++e) {
      double* o = out.ovf.data() + ((size_t)m * pl.n_ovf + e) * kOvfParams;
      for (int i = 0; i < kOvfParams; ++i) o[i] = 0.0;
      const int bond = pl.ovf_bond[e];
      o[0] = l0[2 * bond]; o[1] = l0[2 * bond + 1];
      if (!(o[0] * o[0] + o[1] * o[1] > 0.0)) { F.bad = 1; return; }
      o[2] = kb[3 * bond]; o[3] = kb[3 * bond + 1]; o[4] = kb[3 * bond + 2];
      if (ph) { o[5] = ph[2 * bond]; o[6] = ph[2 * bond + 1]; }
    }
    for (int i = 0; i < NB * 3; ++i) {
      double mass = q->inertia[(size_t)m * NB * 3 + i];
      if (!(mass > 0.0)) { F.bad = 2; return; }
      out.inv_m[(size_t)m * NB * 3 + i] = 1.0 / mass;
      out.damping[(size_t)m * NB * 3 + i] = q->damping ? q->damping[(size_t)m * NB * 3 + i] : 0.0;
    }
    if (q->contact) for (int i = 0; i < 3; ++i) out.contact[m * 3 + i] = q->contact[m * 3 + i];
    for (int f = 0; f < pl.n_fns; ++f) {
      TimeFn& tf = out.fns[(size_t)m * DFX_MAX_FNS + f];
      tf.type = pl.fn_type[f];
      tf.n_tab = (int)(pl.fn_table[f].size() / 2);
      tf.tab = pl.fn_table_ptr[f];
      for (int i = 0; i < DFX_FN_PARAMS; ++i) tf.p[i] = q->fn_params[((size_t)m * pl.n_fns + f) * DFX_FN_PARAMS + i];
    }
    // ---- GPU structure-of-arrays image: one pass over the slots -- node vector, void angles, dictionary index of the reference vector, the
    // range of the void angles (culling bound), and whether the stiffnesses are uniform
    if (!gpu_image) return;
    int n_dict = 0;
    double* dict = out.l_dict.data() + (size_t)m * 1024;
    double lo = 1e300, hi = -1e300;
    double s[kSlotParams];
    for (int s_ = 0; s_ < NS; ++s_) {
      slot_values(m, s_ / kSlots, s_ % kSlots, s);
      double* r = out.p_r.data() + ((size_t)m * NS + s_) * 2;
      double* ph2 = out.p_phi.data() + ((size_t)m * NS + s_) * 2;
      r[0] = s[0]; r[1] = s[1];
      ph2[0] = ph2[1] = 0.0;
      out.l_idx[(size_t)m * NS + s_] = 0;
      if (pl.slot_info[s_] < 0) continue;
      if (!(s[2] * s[2] + s[3] * s[3] > 0.0)) { F.bad = 1; return; }
      ph2[0] = s[7]; ph2[1] = s[8];
      lo = s[7] < lo ? s[7] : lo; hi = s[7] > hi ? s[7] : hi;
      lo = s[8] < lo ? s[8] : lo; hi = s[8] > hi ? s[8] : hi;
      if (s[4] != kb[0] || s[5] != kb[1] || s[6] != kb[2]) F.k_uniform = false;
      if (F.l_dict_ok) {      // dictionary of reference vectors
        int hit = -1;
        for (int d = 0; d < n_dict; ++d) if (dict[4 * d] == s[2] && dict[4 * d + 1] == s[3]) { hit = d; break; }
        if (hit < 0) {
          if (n_dict == 256) F.l_dict_ok = false;
          else {
            hit = n_dict++; dict[4 * hit] = s[2]; dict[4 * hit + 1] = s[3];
            dict[4 * hit + 2] = sqrt(s[2] * s[2] + s[3] * s[3]); dict[4 * hit + 3] = 1.0 / dict[4 * hit + 2];
          }
        }
        if (hit >= 0) out.l_idx[(size_t)m * NS + s_] = (uint8_t)hit;
      }
    }
    F.n_dict = n_dict;
    for (int e = 0; e < pl.n_ovf; ++e) {
      const double* o = out.ovf.data() + ((size_t)m * pl.n_ovf + e) * kOvfParams;
      if (o[2] != kb[0] || o[3] != kb[1] || o[4] != kb[2]) F.k_uniform = false;
    }
    if (q->contact) for (int i = 0; i < 3; ++i) out.cst[(size_t)m * 16 + i] = q->contact[m * 3 + i];
    if (q->contact && pl.contact == 1) {
      // Angle contact, culling bound: with phi in [phi_lo, phi_hi] over the member's ligament ends, a relative rotation |kappa| <= kappa_safe
      // = min(phi_lo - cutoff, pi - phi_hi) leaves both void angles wrap(phi -+ kappa) in [cutoff, pi]: the penalty and all its
      // derivatives are exactly zero whatever phi is, so the kernels do not load phi for such ligaments (cst[9] = kappa_safe with a
      // rounding margin, <= 0: never skip; cst[10] = phi_lo, the stand-in value).
      double safe = -1.0;
      if (lo <= hi) { const double a = lo - q->contact[m * 3 + 1], b2 = 3.14159265358979323846 - hi; safe = (a < b2 ? a : b2) * (1.0 - 1e-12) - 1e-12; }
      out.cst[(size_t)m * 16 + 9] = safe;
      out.cst[(size_t)m * 16 + 10] = lo <= hi ? lo : 0.0;
    }
    if (pl.n_bonds > 0) for (int i = 0; i < 3; ++i) out.cst[(size_t)m * 16 + 3 + i] = kb[i];
    for (int d = 0; d < 3; ++d) {
      const double d0 = out.damping[(size_t)m * NB * 3 + d];
      out.cst[(size_t)m * 16 + 6 + d] = d0;
      for (int b = 1; b < NB && F.damping_uniform; ++b) if (out.damping[(size_t)m * NB * 3 + b * 3 + d] != d0) F.damping_uniform = false;
    }
  };
  dfx_hostpar::for_each(B, (size_t)NS * 16, member);
  out.k_uniform = true; out.l_dict_ok = true; out.damping_uniform = true; out.n_dict_max = 0;
  for (int m = 0; m < B; ++m) {
    const Flags& F = flags[m];
    if (F.bad == 1) { err = "set_params: zero-length reference vector"; return 1; }
    if (F.bad == 2) { err = "set_params: inertia must be positive"; return 1; }
    out.k_uniform = out.k_uniform && F.k_uniform;
    out.l_dict_ok = out.l_dict_ok && F.l_dict_ok;
    out.damping_uniform = out.damping_uniform && F.damping_uniform;
    if (F.n_dict > out.n_dict_max) out.n_dict_max = F.n_dict;
  }
  if (gpu_image && (!out.l_dict_ok || !out.k_uniform)) {
    const bool want_l = !out.l_dict_ok, want_k = !out.k_uniform;
    dfx_hostpar::for_each(B, (size_t)NS * 16, [&](int m) { fill_l_k(m, want_l, want_k); });
  }
  return 0;
}

// Gradient accumulators in device layout (host mirror); unpacked into dfx_grads.
//   slot_g:  batch * n_slots * kSlotGrads :  r(2) | l0(2) k(3) phi(2) contact(3)  (bond part only on end-1 slots)
//   blk_g:   batch * n_blocks * 6         :  d/d m (3), d/d damping (3)
//   spec_g:  batch * n_special * 3 * 2    :  u_bar of constrained DOFs folded with fn partials is done
//            on the fly: fn_g = batch * DFX_MAX_FNS * DFX_FN_PARAMS per special block
constexpr int kSlotGrads = 12;

inline void unpack_grads(const Plan& pl, const std::vector<double>& slot_g, const std::vector<double>& blk_g,
                         const std::vector<double>& fn_g /* batch * max(1,n_special) * MAX_FNS*FN_PARAMS */,
                         const std::vector<double>& inv_m, dfx_grads* g, const std::vector<double>* ovf_g = nullptr /* batch * n_ovf * kOvfGrads */) {
  const int B = pl.batch, NS = pl.n_slots, NB = pl.n_blocks;
  for (int m = 0; m < B; ++m) {
    const double* sg = slot_g.data() + (size_t)m * NS * kSlotGrads;
    if (g->centroid_node_vectors)
      for (int b = 0; b < NB; ++b)
        for (int k = 0; k < pl.n_npb; ++k)
          for (int c = 0; c < 2; ++c)
            g->centroid_node_vectors[((size_t)m * NB * pl.n_npb + b * pl.n_npb + k) * 2 + c] =
                sg[(size_t)(b * kSlots + k) * kSlotGrads + c];
    double con[3] = {0, 0, 0};
    if (g->reference_vector) memset(g->reference_vector + (size_t)m * pl.n_bonds * 2, 0, sizeof(double) * pl.n_bonds * 2);
    if (g->k_bond) memset(g->k_bond + (size_t)m * pl.n_bonds * 3, 0, sizeof(double) * pl.n_bonds * 3);
    if (g->void_angle0) memset(g->void_angle0 + (size_t)m * pl.n_bonds * 2, 0, sizeof(double) * pl.n_bonds * 2);
    for (int s = 0; s < NS; ++s) {
      int info = pl.slot_info[s];
      if (info < 0 || (info & 1)) continue;  // bond parameters live on the end-1 slot
      int bond = pl.slot_bond[s];
      const double* q = sg + (size_t)s * kSlotGrads;
      if (g->reference_vector) { g->reference_vector[((size_t)m * pl.n_bonds + bond) * 2] = q[2]; g->reference_vector[((size_t)m * pl.n_bonds + bond) * 2 + 1] = q[3]; }
      if (g->k_bond) for (int c = 0; c < 3; ++c) g->k_bond[((size_t)m * pl.n_bonds + bond) * 3 + c] = q[4 + c];
      if (g->void_angle0) { g->void_angle0[((size_t)m * pl.n_bonds + bond) * 2] = q[7]; g->void_angle0[((size_t)m * pl.n_bonds + bond) * 2 + 1] = q[8]; }
      for (int c = 0; c < 3; ++c) con[c] += q[9 + c];
    }
    for (int e = 0; e < pl.n_ovf && ovf_g; ++e) {          // extra ligaments: like the slots, everything of a bond on its end-0 entry
      const int bond = pl.ovf_bond[e], end = pl.ovf_info[e] & 1;
      const double* q = ovf_g->data() + ((size_t)m * pl.n_ovf + e) * kOvfGrads;
      if (end) continue;
      if (g->void_angle0) { g->void_angle0[((size_t)m * pl.n_bonds + bond) * 2] = q[7]; g->void_angle0[((size_t)m * pl.n_bonds + bond) * 2 + 1] = q[8]; }
      if (g->reference_vector) { g->reference_vector[((size_t)m * pl.n_bonds + bond) * 2] = q[2]; g->reference_vector[((size_t)m * pl.n_bonds + bond) * 2 + 1] = q[3]; }
      if (g->k_bond) for (int c = 0; c < 3; ++c) g->k_bond[((size_t)m * pl.n_bonds + bond) * 3 + c] = q[4 + c];
      for (int c = 0; c < 3; ++c) con[c] += q[9 + c];
    }
    if (g->contact) for (int c = 0; c < 3; ++c) g->contact[m * 3 + c] = con[c];
    for (int i = 0; i < NB * 3; ++i) {
      int b = i / 3, d = i % 3;
      if (g->inertia) g->inertia[(size_t)m * NB * 3 + i] = blk_g[((size_t)m * NB + b) * 6 + d];
      if (g->damping) g->damping[(size_t)m * NB * 3 + i] = blk_g[((size_t)m * NB + b) * 6 + 3 + d];
    }
    if (g->fn_params) {
      const int W = DFX_MAX_FNS * DFX_FN_PARAMS;
      for (int f = 0; f < pl.n_fns; ++f)
        for (int i = 0; i < DFX_FN_PARAMS; ++i) {
          double acc = 0.0;
          for (int sidx = 0; sidx < pl.n_special; ++sidx) acc += fn_g[((size_t)m * pl.n_special + sidx) * W + f * DFX_FN_PARAMS + i];
          g->fn_params[((size_t)m * pl.n_fns + f) * DFX_FN_PARAMS + i] = acc;
        }
    }
  }
  (void)inv_m;
}

}  // namespace dfx
