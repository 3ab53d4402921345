cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02bo; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "adaptive" > $O/pytest_adaptive.log 2>&1; echo "rc $?" >> $O/pytest_adaptive.log)
tail -3 $O/pytest_adaptive.log
timeout 600 python tools/adaptive_probe.py 16 2 1e-8 > $O/adaptive_probe.txt 2>&1
DFX_STREAMS=1 timeout 600 python tools/adaptive_probe.py 16 2 1e-8 > $O/adaptive_probe_1stream.txt 2>&1
grep "adaptive rtol" $O/adaptive_probe.txt | tail -1 | cut -c1-200; grep "adaptive rtol" $O/adaptive_probe_1stream.txt | tail -1 | cut -c1-200
python - <<'PY'
# two groups vs one: identical fields?
import os, sys, numpy as np, subprocess
PY
