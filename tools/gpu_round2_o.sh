cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02o; mkdir -p $O
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log)
echo "== graphs (DFX_EAGER_STEPS=0), 96 members, 2 groups per engine, three threads: round-1 stall?" >> $O/stall_graph.log
DFX_EAGER_STEPS=0 timeout 240 python tools/stall_probe.py 96 100 1 >> $O/stall_graph.log 2>&1; echo "rc $?" >> $O/stall_graph.log
echo "== graphs, 1 stream per engine, three threads" >> $O/stall_graph.log
DFX_EAGER_STEPS=0 DFX_STREAMS=1 timeout 240 python tools/stall_probe.py 96 100 1 >> $O/stall_graph.log 2>&1; echo "rc $?" >> $O/stall_graph.log
tail -3 $O/smoke.log; cat $O/stall_graph.log
