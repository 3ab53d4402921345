import cProfile, pstats, sys, io, runpy
sys.argv = ["tools/c4_problem_timing.py", "8", "4000"]
pr = cProfile.Profile()
pr.enable()
runpy.run_path("tools/c4_problem_timing.py", run_name="__main__")
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
