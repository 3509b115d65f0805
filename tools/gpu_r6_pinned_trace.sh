#!/bin/bash
# round 6: kernel + memory-copy trace of bench.py's pinned Pedersen leg (the 14 ms case of profiles/r06_s41) on the final tree: the rows
# around three gated launches.  AKP_TE_PINNED_FORM=gated so that every call of the leg is a gated one.
O=gpurun_out/${1:-r06_s56}; mkdir -p $O
MIN="--merkle-log2 0 --bh-merkle-log2 0 --proofs-log2 0 --ragged-log2 0 --no-sweep --sustain-seconds 0 --no-cpu-baseline"
export AKP_BENCH_FULL=$GRAFT_REPO_ROOT/$O/full.json AKP_TE_PINNED_FORM=gated
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr -- python $GRAFT_REPO_ROOT/bench.py $MIN > $GRAFT_REPO_ROOT/$O/line.json 2> $GRAFT_REPO_ROOT/$O/err.txt
cd $GRAFT_REPO_ROOT
python - $O <<'P'
import csv,glob,json,sys
O=sys.argv[1]
j=json.load(open(O+"/full.json"))["host_path"]
print("host_path under the tracer", {a:round(b["ms_per_batch"],2) for a,b in j.items() if isinstance(b,dict) and "ms_per_batch" in b})
rows=[]
for f in glob.glob("/tmp/tr/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"K",r["Kernel_Name"][:60],r.get("Queue_Id",""),r.get("Stream_Id","")))
for f in glob.glob("/tmp/tr/**/*memory_copy_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"M",r.get("Direction","")[12:],"",r.get("Stream_Id","")))
rows.sort()
g=[i for i,r in enumerate(rows) if "gated" in r[3]]
out=open(O+"/trace_around_gated.txt","w")
for idx in g[4:6]:
    t0=rows[idx][0]
    out.write("---- gated kernel at row %d (start [ms] relative to it, duration [ms], K kernel / M copy, name, queue, stream)\n"%idx)
    for r in rows[max(0,idx-6):idx+75]:
        if r[0]-t0 > rows[idx][1]-t0+300000: break
        out.write("%9.3f %8.3f %s %-62s q=%s s=%s\n"%((r[0]-t0)/1e6,(r[1]-r[0])/1e6,r[2],r[3],r[4],r[5]))
out.close()
w=[ (r[1]-r[0])/1e6 for r in rows if "streamOpsWrite" in r[3]]
w.sort()
print("streamOpsWrite kernels:",len(w),"median %.4f ms, p90 %.4f, max %.4f"%(w[len(w)//2],w[int(len(w)*0.9)],w[-1]))
k=[(r[1]-r[0])/1e6 for r in rows if "gated" in r[3]]
print("gated kernels:",len(k),"median %.3f ms"%sorted(k)[len(k)//2])
P
head -60 $O/trace_around_gated.txt
