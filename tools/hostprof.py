import time, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
fw, obj, des = bench.c3_problem(128, 3, int(os.environ.get("M", "8")))
eng = fw.solve_dynamics.engine
eng.reserve(2500, 11, True)
bench.run_once(fw, obj, des, 250)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
t = time.perf_counter(); r = bench.run_once(fw, obj, des, 2500); w = time.perf_counter() - t
pr.disable()
print("wall", w * 1e3, "dev", r["fwd_ms"] + r["adj_ms"])
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
