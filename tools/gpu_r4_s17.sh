OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s17; mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_s17/bench.json') if l.startswith('{')][-1])
r=d['roofline']
print('value %.4g ms/step %.3f frac %.5f eff_sclk %s before %s cycles/mad %s power %s cap %s temp %s mad_frac %.3f (nominal %.3f)'%(d['value'],d['ms_per_step'],r['frac'],r['effective_sclk']['during_timed_steps_mhz'],r['effective_sclk']['before_mhz'],r['effective_sclk']['cycles_per_dependent_mad'],r['power_w_after_timed_steps'],r['power_cap_w'],r['temp_c_max'],r['valu']['frac_of_mad_issue_peak'],r['valu']['frac_of_mad_issue_peak_at_nominal_2400mhz']))
print('sweep', json.dumps(d.get('sweep'))[:900])
print('pred', json.dumps(d.get('predicted_scaling'))[:1200])
print('host', json.dumps({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if 'ms' in kk or 'per_s' in kk}) for k,v in d['host_path'].items()})[:1200])
print('merkle', d['merkle']['seconds'], json.dumps(d['merkle'].get('one_process_c_abi'))[:1500])
print('ped', d['pedersen']['hashes_per_s'], 'bh', d['bh_merkle']['leaves_per_s'], 'verify', d['proofs']['poseidon']['verify_paths'])
print('sust', json.dumps(d['sustained'])[:700])
print('curve_parity', d['curve_parity'])
PY
timeout 1500 python -m pytest tests/test_gpu_bench_contract.py -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest_contract.txt
