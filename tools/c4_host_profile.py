"""cProfile of config 4 through the problem layer at its per-GPU width of an 8-GPU run (8 designs): where the wall time outside the two
sweeps goes.  usage: python tools/c4_host_profile.py [B] [steps]"""
import cProfile, io, os, pstats, runpy, sys
sys.argv = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "c4_problem_timing.py")] + sys.argv[1:]
pr = cProfile.Profile()
pr.enable()
runpy.run_path(sys.argv[0], run_name="__main__")
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue())
