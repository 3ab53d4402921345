"""Experiment: does running two independent engine handles concurrently (two streams) hide launch bubbles?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

K = 2500
def setup(members, seed):
    fw, obj, des = bench.c3_problem(128, seed, members)
    fw.solve_dynamics.engine.reserve(K, K // bench.SPI + 1, True)
    bench.run_once(fw, obj, des, 250)
    return fw, obj, des

for groups, members in ((1, 4), (2, 2), (4, 1), (2, 4), (1, 8)):
    hs = [setup(members, 3 + 10 * g) for g in range(groups)]
    res = [None] * groups
    def work(i):
        res[i] = bench.run_once(*hs[i], K)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(groups)]
    [t.start() for t in th]; [t.join() for t in th]
    w = time.perf_counter() - t0
    tot = K * 16384 * members * groups
    print(f"groups {groups} x members {members}: wall {w*1e3:.0f} ms  value {tot/w:.3e}  dev fwd {res[0]['fwd_ms']:.0f} adj {res[0]['adj_ms']:.0f}", flush=True)
    del hs
