"""Two PROCESSES on one GPU, both with the persistent stage loop switched on (which INTEGRATION.md tells several processes per GPU not to
do: the account that keeps concurrent persistent launches within the chip is per process).  What happens when their workgroups do not
all fit together: every wave's spins are bounded, so each solve must either finish or come back with the engine's error -- never hang.
usage: python tools/two_process_persist.py [LATTICE N MEMBERS STEPS] [REPEATS]      (every child runs under its own timeout)"""
import os, subprocess, sys, time
here = os.path.dirname(os.path.abspath(__file__))
args = sys.argv[1:5] if len(sys.argv) >= 5 else ["kagome", "64", "8", "2000"]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
child = r'''
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[0]))) if False else %r
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from common import Case
lattice, n, B, steps, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
c = Case(lattice, n, True, True, seed=100, lib=None, cutoff_deg=125.0 if lattice == "kagome" else -10.0, batch=B)
cp = c.cp._replace(constraint_params=dict(amplitude=7.5, loading_rate=1000.0, input_delay=1e-5))
nb = c.geo.n_blocks
ts = np.linspace(0.0, 1e-3, 5)
target = np.array([nb // 2 + 1, nb // 2 + 2], dtype=np.int32)
y0 = np.zeros((2, nb, 3))
for rep in range(reps):
    t0 = time.perf_counter()
    try:
        c.solver(y0, ts, [cp] * B if B > 1 else cp, keep_trajectory=True, steps_per_interval=steps // 4, want_fields=False)
        obj, raw = c.solver.kinetic_energy_value_and_raw(target)
        print("pid", os.getpid(), "rep", rep, "ok: objective %%.12e, builds %%s / %%s, %%.0f ms" %% (float(np.atleast_1d(obj)[0]), c.solver.stats["tile_kernels"],
              c.solver.adjoint_stats["tile_kernels"], 1e3 * (time.perf_counter() - t0)), flush=True)
    except RuntimeError as e:
        print("pid", os.getpid(), "rep", rep, "engine error after %%.0f ms: %%s" %% (1e3 * (time.perf_counter() - t0), str(e)[:200]), flush=True)
''' % os.path.dirname(here)
env = dict(os.environ, DFX_PERSIST="1")
t0 = time.perf_counter()
procs = []
for k in range(2):
    procs.append(subprocess.Popen(["timeout", "240", sys.executable, "-c", child] + args + [str(reps)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    time.sleep(float(os.environ.get("STAGGER_S", "0")))
for p in procs:
    out, _ = p.communicate()
    print(out.strip()); print("exit code", p.returncode)
print("both children done after %.1f s" % (time.perf_counter() - t0))
# and one process alone, for the reference objective
p = subprocess.run(["timeout", "240", sys.executable, "-c", child] + args + ["1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
print("alone:", p.stdout.strip())
