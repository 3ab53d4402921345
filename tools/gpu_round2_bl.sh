cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02bl; mkdir -p $O
timeout 600 python bench.py --gpus 2 --backend socket --all-ranks-device 0 --steps 20 --warmup 5 --members 4 --no-cpu-baseline > $O/two_rank.json 2> $O/two_rank.err
python - <<PY
import json
l=json.loads(open("$O/two_rank.json").read().strip().splitlines()[-1])
print("two ranks (socket rehearsal on one GPU):", "%.3e"%l["value"], "n_gpus", l["n_gpus"], l["config"]["collective"], len(l["objective"]))
PY
tail -3 $O/two_rank.err
