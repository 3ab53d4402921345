"""profiles/pmc_traffic.json from a pmc_summary.json of tools/profile_round.sh: HBM-side bytes per member and launch of the two
stage kernels, FETCH_SIZE doubled as MI355X_MICROARCH.md (HBM section) prescribes for gfx950, WRITE_SIZE as counted.
usage: python tools/make_pmc_traffic.py gpurun_out/prof_TAG/pmc_summary.json TAG MEMBERS CHECKPOINT"""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_digest      # noqa: E402   (the sources the counters were collected for: run this right after the GPU session)
src, tag, members, ck = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
try:
    commit = subprocess.run(["git", "log", "-1", "--format=%h", "--", "difflexmm_amd/csrc"], capture_output=True, text=True, check=True).stdout.strip()
    dirty = subprocess.run(["git", "status", "--porcelain", "--", "difflexmm_amd/csrc"], capture_output=True, text=True).stdout.strip()
    commit += "+uncommitted" if dirty else ""
except Exception:       # noqa: BLE001
    commit = None
p = json.load(open(src))
out = {"source": f"tools/profile_round.sh {tag} {members} on MI355X: rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | TCC_HIT TCC_MISS | 8 SQ counters | "
                 f"6 SQ counters> --kernel-trace --output-format csv, separate passes; python3 bench.py --streams 1 --members {members} "
                 f"--steps 250 --warmup 0 --no-cpu-baseline --no-single --no-roofline-leg --no-as-written (DFX_DUAL_CHAIN=0); checkpoint level: {ck}",
       "units": "mean per dispatch; FETCH_SIZE/WRITE_SIZE in KiB; gfx950 correction (MI355X_MICROARCH.md, HBM): read bytes = 2 * FETCH_SIZE * 1024",
       "calibration": "the guide calibrates the x2 for 16-B-per-lane streaming reads only and calls other widths uncalibrated; these kernels mix "
                      "16-B (records, node vectors, lambda / Ybar pairs) and 8-B (per-DOF arrays, void angles) loads.  Calibrated on the kernels' own "
                      "access pattern instead: a by-hand count of the bytes each build REQUESTS per unit (DESIGN.md section 4) matched the corrected "
                      "counter to 1 % for two builds that differ by 166 B/unit (r02_v3: 777 B counted by hand vs 777 B; r02_v9: 611 vs 611), i.e. "
                      "the x2 holds for this mix; ratios between builds do not depend on it",
       "members": members, "checkpoint": ck, "csrc_sha256": csrc_digest(), "commit": commit, "counters": p}
for k, name in (("fwd", "k_fwd_stage"), ("adj", "k_adj_stage")):
    rd, wr = 2 * p[k]["FETCH_SIZE"] * 1024 / members, p[k]["WRITE_SIZE"] * 1024 / members
    out[f"{name}_read_bytes_per_member_launch"] = rd
    out[f"{name}_write_bytes_per_member_launch"] = wr
    out[f"{name}_bytes_per_member_launch"] = rd + wr
    out[f"{name}_bytes_per_unit_launch"] = (rd + wr) / (p[k]["SQ_WAVES"] * 16 / members) if "SQ_WAVES" in p[k] else None
    if "TCC_HIT" in p[k]:
        out[f"{name}_l2_hit_rate"] = p[k]["TCC_HIT"] / (p[k]["TCC_HIT"] + p[k]["TCC_MISS"])
out["algorithmic_bytes_per_member_launch"] = {"k_fwd_stage": 344 * 16384, "k_adj_stage": 624 * 16384}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if "per_" in k or "hit" in k}, indent=1))
