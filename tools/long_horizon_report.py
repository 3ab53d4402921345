"""Deviation of an engine library from the long-horizon goldens of the torch oracle (tests/long_horizon.py): prints what the tests bound."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import long_horizon  # noqa: E402

lib = None
if len(sys.argv) > 1 and sys.argv[1] == "cpu":
    from oracle.cpu import load
    lib = load()
for lattice in ("quads", "kagome"):
    print(lattice, {k: float(v) for k, v in long_horizon.check(lib, lattice).items()})
