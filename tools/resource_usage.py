#!/usr/bin/env python3
"""Registers / scratch / LDS / occupancy of the gfx950 kernels in libdfx (hipcc -Rpass-analysis=kernel-resource-usage), one line per
kernel whose demangled-ish name contains any of the given substrings (default: the stage and pair kernels with nonlinear + contact)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pats = sys.argv[1:] or ["k_fwd_stageILi1ELi1E", "k_adj_stageILi1ELi1ELi0ELi0E", "k_adj_stage_rbILi1ELi1E", "k_fwd_pairILi1ELi1E", "k_adj_pairILi1ELi1E"]
import shlex
BASE = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-disable-machine-licm", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null"]
EXTRA = os.environ.get("DFX_EXTRA_FLAGS", "").split()
CSRC = os.path.join(ROOT, "difflexmm_amd", "csrc")
# the per-stage builds of the reverse kernel are a translation unit of their own with the Makefile's ADJFLAGS
ADJ = shlex.split(re.search(r"^ADJFLAGS = (.*)$", open(os.path.join(CSRC, "Makefile")).read(), re.M).group(1))
out = subprocess.run(BASE + [os.path.join(CSRC, "engine_launch.hip")] + EXTRA, capture_output=True, text=True).stderr
out += subprocess.run(BASE + ADJ + [os.path.join(CSRC, "stage_builds_adj.hip")] + EXTRA, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    txt = m.group(1).strip()
    if txt.startswith("Function Name:") or txt.startswith("Name:"):
        cur = txt.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in txt:
        k, v = txt.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    if any(p in name for p in pats):
        print(f"{name[:70]:70s} VGPR {r.get('VGPRs')} AGPR {r.get('AGPRs')} scratch {r.get('ScratchSize [bytes/lane]')} "
              f"occ {r.get('Occupancy [waves/SIMD]')} SGPR {r.get('TotalSGPRs') or r.get('SGPRs')} LDS {r.get('LDS Size [bytes/block]')}")
