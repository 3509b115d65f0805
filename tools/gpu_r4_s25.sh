OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s25; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-path --sustain-seconds 0 --no-sweep --proofs-log2 0 > $OUT/bench_under_rocprofv3.json 2> $OUT/bench_under_rocprofv3.err
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/rocprof_kernel_stats_bench_py.csv; head -8 $f | cut -c1-260; rm -rf $OUT/prof
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
for name in ('bench_under_rocprofv3','bench'):
    d=json.loads([l for l in open('gpurun_out/r04_s25/%s.json'%name) if l.startswith('{')][-1])
    r=d['roofline']
    print(name,'value %.4g ms/step %.3f kern_avg %.4f frac %.5f eff %.1f power %s'%(d['value'],d['ms_per_step'],r['kernel_avg_ms'],r['frac'],r['effective_sclk_mhz'] or 0,r['power_w_under_load']))
PY
