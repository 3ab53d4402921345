cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02q; mkdir -p $O
echo "== torch.distributed.run as launcher, 2 ranks on one GPU, socket backend" > $O/multi.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend socket --all-ranks-device 0 --steps 20 --warmup 5 --members 4 --no-single --no-roofline-leg 2>> $O/multi.err | cut -c1-700 >> $O/multi.log; echo "rc ${PIPESTATUS[0]}" >> $O/multi.log
echo "== same, rccl backend: both ranks on device 0 -> RCCL must refuse (duplicate GPU) and the ranks must agree on the fallback, not hang" >> $O/multi.log
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --all-ranks-device 0 --steps 20 --warmup 5 --members 4 --no-single --no-roofline-leg 2>> $O/multi.err | cut -c1-700 >> $O/multi.log; echo "rc ${PIPESTATUS[0]}" >> $O/multi.log
cat $O/multi.log; grep -v "^$" $O/multi.err | tail -12 | cut -c1-300
