#!/usr/bin/env python3
"""cProfile of examples/multi_input_ensemble.py (config 5) on the GPU box: where the wall time outside the engines goes."""
import cProfile, pstats, sys, runpy, os
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [os.path.join(here, "examples/multi_input_ensemble.py")] + (sys.argv[1:] or ["--members", "256", "--iterations", "4"])
pr = cProfile.Profile(); pr.enable()
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
except SystemExit:
    pass
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(25)
st.sort_stats("cumtime").print_stats(45)
