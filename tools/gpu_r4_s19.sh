OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s19; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee $OUT/pytest_gpu_full.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_s19/bench.json') if l.startswith('{')][-1])
r=d['roofline']
print('value %.4g ms/step %.3f kern %.3f frac %.5f eff %.1f pre %s post %s power %s mad_frac %.3f'%(d['value'],d['ms_per_step'],r['kernel_avg_ms'],r['frac'],r['effective_sclk_mhz'],r['effective_sclk']['before_timed_steps_mhz'],r['effective_sclk']['after_timed_steps_mhz'],r['power_w_under_load'],r['valu']['frac_of_mad_issue_peak']))
hp=d['host_path']; print({k:(round(v['ms_per_batch'],3),round(v['ms_min'],3),round(v['ms_max'],3)) for k,v in hp.items() if isinstance(v,dict) and 'ms_per_batch' in v})
print('sust', {k:(round(v['permutations_per_s']/1e8,3), v['launch_ms_median']) for k,v in d['sustained'].items()})
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
