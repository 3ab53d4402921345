cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02ab; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -o k20 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-leg --no-single > $O/bench.json 2> $O/bench.err
ls -la $O/trace/* | head; 
python3 - <<'PY'
import csv, glob, os
O="gpurun_out/r02ab"
kt=glob.glob(O+"/trace/**/*kernel_trace.csv", recursive=True)[0]
mc=glob.glob(O+"/trace/**/*memory_copy_trace.csv", recursive=True)
rows=[]
for r in csv.DictReader(open(kt)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], "K"))
if mc:
    for r in csv.DictReader(open(mc[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction","copy")+" "+r.get("Name","")[:30], "M"))
rows.sort()
# keep the tail: last 2500 events, compressed
tail=rows[-700:]
t0=tail[0][0]
with open(O+"/timeline_tail.txt","w") as f:
    for s,e,n,k in tail:
        f.write(f"{(s-t0)/1e3:10.1f} {(e-s)/1e3:8.1f} {k} {n}\n")
print(len(rows))
PY
tail -5 $O/timeline_tail.txt
