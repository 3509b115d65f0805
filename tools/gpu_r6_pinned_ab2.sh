#!/bin/bash
# round 6: why bench.py's pinned Pedersen leg takes 16 ms per 2^20 hashes when the same call takes 3.9 ms in tools/gpu_r5_gated.py (profiles/r06_s40):
# (A) more hardware queues, (B) only the legs the host path needs, (C) a kernel + memory-copy trace of the full run around the gated kernel.
O=gpurun_out/r06_s41; mkdir -p $O
export AKP_BENCH_FULL=$PWD/$O/full.json
show() { python - "$1" <<'P'
import json,sys
j=json.load(open("gpurun_out/r06_s41/full.json"))["host_path"]
print(sys.argv[1], {a:round(b["ms_per_batch"],2) for a,b in j.items() if isinstance(b,dict) and "ms_per_batch" in b})
P
}
MIN="--merkle-log2 0 --bh-merkle-log2 0 --proofs-log2 0 --ragged-log2 0 --no-sweep --sustain-seconds 0 --no-cpu-baseline"
GPU_MAX_HW_QUEUES=16 timeout 300 python bench.py --no-sweep --sustain-seconds 0 --no-cpu-baseline > $O/A.json 2> $O/A.err; show A_hwq16
timeout 300 python bench.py $MIN > $O/B.json 2> $O/B.err; show B_minimal
timeout 300 python bench.py $MIN --merkle-log2 24 > $O/B2.json 2> $O/B2.err; show B2_minimal_plus_merkle
timeout 300 python bench.py $MIN --bh-merkle-log2 23 > $O/B3.json 2> $O/B3.err; show B3_minimal_plus_bh
timeout 300 python bench.py $MIN --proofs-log2 20 > $O/B4.json 2> $O/B4.err; show B4_minimal_plus_proofs
timeout 300 python bench.py $MIN --ragged-log2 20 > $O/B5.json 2> $O/B5.err; show B5_minimal_plus_ragged
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr -- python $GRAFT_REPO_ROOT/bench.py --no-sweep --sustain-seconds 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/C.json 2> $GRAFT_REPO_ROOT/$O/C.err
cd $GRAFT_REPO_ROOT; show C_traced
python - <<'P'
import csv,glob,os
rows=[]
for f in glob.glob("/tmp/tr/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"K",r["Kernel_Name"][:60],r.get("Queue_Id",""),r.get("Stream_Id","")))
for f in glob.glob("/tmp/tr/**/*memory_copy_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"M",r.get("Direction","")+" "+r.get("Bytes","")[:12] if "Bytes" in r else str(r)[:80],"",r.get("Stream_Id","")))
rows.sort()
g=[i for i,r in enumerate(rows) if "gated" in r[3]]
print("rows",len(rows),"gated launches",len(g))
out=open("gpurun_out/r06_s41/trace_around_gated.txt","w")
for idx in g[2:5]:
    t0=rows[idx][0]
    out.write("---- gated kernel at row %d\n"%idx)
    for r in rows[max(0,idx-40):idx+60]:
        out.write("%10.3f %10.3f %s %-62s q=%s s=%s\n"%((r[0]-t0)/1e6,(r[1]-r[0])/1e6,r[2],r[3],r[4],r[5]))
out.close()
P
head -c 3000 gpurun_out/r06_s41/trace_around_gated.txt
