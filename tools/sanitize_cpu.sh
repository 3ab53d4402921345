#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer run of the engine's host side (plan builder, parameter packing, per-slot stage
# arithmetic: dfx_plan.h / dfx_stage.h / dfx_physics.h), through the CPU port that compiles those same headers with g++.
# GPU sanitizers are not available on the pool; this is the CPU-build sanitizer run.   usage: tools/sanitize_cpu.sh [pytest args]
set -e
cd "$(dirname "$0")/.."
make -C oracle/cpu asan
ASAN_SO=$(gcc -print-file-name=libasan.so)
export DFX_CPU_PORT_LIBRARY="$PWD/oracle/cpu/libdfx_cpu_asan.so"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-4}
tests=${@:-tests/test_cpu_port.py tests/test_edge_cases.py tests/test_general_bonds.py tests/test_spring_models.py tests/test_distance_contact.py}
LD_PRELOAD="$ASAN_SO" python -m pytest -x -q -m "not gpu" -p no:cacheprovider $tests
