#!/bin/bash
# The GPU sessions of round 4, one case arm per session: `gpurun -- bash tools/gpu_r4.sh s17`.  Outputs go to gpurun_out/r04_sNN/; what was kept is
# under profiles/r04_s1 .. r04_s9 (index: profiles/README.md).  Some arms select A/B builds through environment variables that existed only
# in the library as built for that session (AKP_TE_MSG_LDS, AKP_TE_ZERO_COPY_IN, AKP_TE_PIPE_*, AKP_VERIFY_WALK, AKP_POSEIDON_STAGED_IO): each
# was removed by the commit that settled it; the arms stay as the record of what was run.
case "$1" in
s1)
# round 4 session 1: parity of the LDS-staged accumulate kernel, then its A/B
mkdir -p gpurun_out/r04_s1
timeout 900 python -m pytest tests/test_gpu_curves.py tests/test_gpu_canaries.py tests/test_gpu_merkle.py tests/test_gpu_features.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_s1/pytest_curves.txt
cat gpurun_out/r04_s1/pytest_curves.txt
for A in 0 1; do AKP_TE_MSG_LDS=$A timeout 600 python tools/gpu_te_msg_lds.py > gpurun_out/r04_s1/te_msg_lds_arm$A.txt 2>&1; cat gpurun_out/r04_s1/te_msg_lds_arm$A.txt; done
;;
s3)
mkdir -p gpurun_out/r04_s3
timeout 900 python -m pytest tests/test_gpu_curves.py tests/test_gpu_canaries.py tests/test_gpu_features.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_s3/pytest_curves.txt
cat gpurun_out/r04_s3/pytest_curves.txt
AKP_TE_MSG_LDS=1 timeout 600 python tools/gpu_te_msg_lds.py > gpurun_out/r04_s3/te_msg_lds_arm1.txt 2>&1; grep "host path" gpurun_out/r04_s3/te_msg_lds_arm1.txt
;;
s4)
mkdir -p gpurun_out/r04_s4
for Z in 0 1; do echo "# AKP_TE_ZERO_COPY_IN=$Z"; AKP_TE_ZERO_COPY_IN=$Z timeout 600 python tools/gpu_te_msg_lds.py 2>&1 | grep "host path"; done > gpurun_out/r04_s4/te_hostpath_ab.txt 2>&1
cat gpurun_out/r04_s4/te_hostpath_ab.txt
;;
s5)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s5; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for combo in "pinned pinned" "pageable pinned" "pinned pageable"; do
  tag=$(echo $combo | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tr_$tag -o t -- python $GRAFT_REPO_ROOT/tools/te_host_trace_driver.py $combo > $OUT/trace_$tag.log 2>&1
  tail -3 $OUT/trace_$tag.log
  python $GRAFT_REPO_ROOT/tools/trace_timeline.py $OUT/tr_$tag 36 > $OUT/timeline_$tag.txt; cat $OUT/timeline_$tag.txt
  rm -rf $OUT/tr_$tag
done
;;
s6)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s6; mkdir -p $OUT
timeout 600 python tools/gpu_te_msg_lds.py 2>&1 | grep "host path" > $OUT/te_hostpath.txt; cat $OUT/te_hostpath.txt
cd /tmp && export TMPDIR=/tmp
for combo in "pinned pinned"; do
  tag=$(echo $combo | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tr_$tag -o t -- python $GRAFT_REPO_ROOT/tools/te_host_trace_driver.py $combo > $OUT/trace_$tag.log 2>&1
  tail -3 $OUT/trace_$tag.log
  python $GRAFT_REPO_ROOT/tools/trace_timeline.py $OUT/tr_$tag 36 > $OUT/timeline_$tag.txt; cat $OUT/timeline_$tag.txt
  rm -rf $OUT/tr_$tag
done
cd $GRAFT_REPO_ROOT && timeout 600 python -m pytest tests/test_gpu_canaries.py tests/test_gpu_curves.py -m gpu -x -q 2>&1 | tail -3
;;
s7)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s7; mkdir -p $OUT
timeout 600 python tools/gpu_te_msg_lds.py 2>&1 | grep "host path" > $OUT/te_hostpath.txt; cat $OUT/te_hostpath.txt
timeout 900 python -m pytest tests/test_gpu_lifetimes.py tests/test_gpu_canaries.py tests/test_gpu_tree_handle.py tests/test_gpu_poseidon.py -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest.txt
;;
s8)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s8; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for combo in "pinned pinned"; do
  tag=$(echo $combo | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tr_$tag -o t -- python $GRAFT_REPO_ROOT/tools/te_host_trace_driver.py $combo > $OUT/trace_$tag.log 2>&1
  tail -3 $OUT/trace_$tag.log
  python $GRAFT_REPO_ROOT/tools/trace_timeline.py $OUT/tr_$tag 40 > $OUT/timeline_$tag.txt; cat $OUT/timeline_$tag.txt
  rm -rf $OUT/tr_$tag
done
cd $GRAFT_REPO_ROOT; for i in 1 2 3; do python tools/te_host_trace_driver.py pinned pinned 2>&1 | tail -2; done
;;
s9)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s9; mkdir -p $OUT
for L in 16 17 18 19; do for i in 1 2; do echo "chunk 2^$L: $(AKP_TE_PIPE_CHUNK_LOG2=$L python tools/gpu_te_msg_lds.py 2>&1 | grep 'pinned in/out')"; done; done | tee $OUT/pipe_chunk_sweep.txt
;;
s10)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s10; mkdir -p $OUT
(for P in 1 0; do for L in 17 18; do echo "prio=$P chunk=2^$L: $(AKP_TE_PIPE_PRIO=$P AKP_TE_PIPE_CHUNK_LOG2=$L python tools/te_host_calls.py pinned pinned 14 2>&1 | tail -1)"; done; done
python tools/te_host_calls.py pageable pageable 14 2>&1 | tail -1
python tools/te_host_calls.py pinned pageable 14 2>&1 | tail -1
python tools/te_host_calls.py pageable pinned 14 2>&1 | tail -1) | tee $OUT/te_host_calls.txt
;;
s11)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s11; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_tree_handle.py -m gpu -x -q -k "lane_walk or poseidon_vs_oracle" 2>&1 | tail -5 | tee $OUT/pytest.txt
for W in 0 1; do for M in 16 18; do
  AKP_VERIFY_WALK=$W timeout 300 python tools/bench_proofs.py --config poseidon --log2-m $M 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])['poseidon']['verify_paths']
print('walk=$W m=2^$M  device %.3f ms  wall %.3f ms  hashes/s(device) %.4g  all_accepted %s neg_control %s' % (d['device_ms'], d['wall_ms'], d['hashes_per_s_device'], d['all_accepted'], d['negative_control_rejected_only_the_wrong_leaf']))"
done; done | tee $OUT/verify_walk_ab.txt
;;
s12)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s12; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_multi_slots.py tests/test_gpu_tree_handle.py -m gpu -x -q 2>&1 | tail -25 | tee $OUT/pytest.txt
;;
s13)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s13; mkdir -p $OUT
tools/clock_probe | tee $OUT/clock_probe.txt
timeout 1500 python -m pytest tests/test_gpu_multi_slots.py tests/test_gpu_tree_handle.py tests/test_gpu_lifetimes.py -m gpu -x -q 2>&1 | tail -25 | tee $OUT/pytest.txt
;;
s14)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s14; mkdir -p $OUT
export AKP_LIB=$GRAFT_REPO_ROOT/crypto_primitives_amd/lib/libakp_testhooks.so
for K in "byte_digests" "resident_tree_poseidon" "sharded_build_logic"; do
  echo "== -k $K"; PYTHONFAULTHANDLER=1 timeout 600 python -m pytest tests/test_gpu_multi_slots.py -m gpu -x -q -p no:cacheprovider -k "$K" 2>&1 | tail -12; echo "rc=$?"
done 2>&1 | tee $OUT/exit_crash.txt
unset AKP_LIB
ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>/dev/null | head -40 > $OUT/hwmon.txt; for f in /sys/class/drm/card*/device/hwmon/hwmon*/power1_*; do echo "$f $(cat $f 2>/dev/null)"; done >> $OUT/hwmon.txt 2>&1
(rocm-smi --showpower --json; rocm-smi --showtemp --json; which amd-smi && amd-smi metric --json | head -c 3000) >> $OUT/hwmon.txt 2>&1
cat $OUT/hwmon.txt | head -80
;;
s15)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s15; mkdir -p $OUT
export AKP_LIB=$GRAFT_REPO_ROOT/crypto_primitives_amd/lib/libakp_testhooks.so
echo "== all, faulthandler"; PYTHONFAULTHANDLER=1 timeout 600 python -X dev -m pytest tests/test_gpu_multi_slots.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -40; echo "rc=$?"
echo "== build_logic + poseidon"; PYTHONFAULTHANDLER=1 timeout 600 python -m pytest tests/test_gpu_multi_slots.py -m gpu -x -q -p no:cacheprovider -k "sharded_build_logic or resident_tree_poseidon" 2>&1 | tail -5
echo "== poseidon + bytes"; PYTHONFAULTHANDLER=1 timeout 600 python -m pytest tests/test_gpu_multi_slots.py -m gpu -x -q -p no:cacheprovider -k "byte_digests or resident_tree_poseidon" 2>&1 | tail -5
echo "== build_logic + bytes"; PYTHONFAULTHANDLER=1 timeout 600 python -m pytest tests/test_gpu_multi_slots.py -m gpu -x -q -p no:cacheprovider -k "byte_digests or sharded_build_logic" 2>&1 | tail -5
;;
s16)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s16; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_multi_slots.py tests/test_gpu_tree_handle.py tests/test_gpu_lifetimes.py -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest.txt
cd /tmp && export TMPDIR=/tmp
for W in 0 1; do
  AKP_VERIFY_WALK=$W timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_walk$W -o p -- python $GRAFT_REPO_ROOT/tools/bench_proofs.py --config poseidon --log2-m 16 > $OUT/proofs_walk$W.json 2>/dev/null
  f=$(find $OUT/prof_walk$W -name "*kernel_stats.csv" | head -1); cp $f $OUT/proofs_poseidon_kernel_stats_walk$W.csv; head -12 $f
  rm -rf $OUT/prof_walk$W
done
;;
s17)
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
;;
s18)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s18; mkdir -p $OUT
python tools/gpu_te_dma_interference.py 2>&1 | grep -v amdgpu.ids | tee $OUT/te_dma_interference.txt
for c in "pinned pinned" "pageable pageable" "pinned pinned" "pageable pageable"; do python tools/te_host_calls.py $c 14 2>&1 | tail -1; done | tee $OUT/te_host_calls.txt
timeout 1500 python -m pytest tests/test_gpu_bench_contract.py -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest_contract.txt
;;
s19)
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
;;
s20)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s20; mkdir -p $OUT
for c in none torch torch,heat; do python tools/te_host_calls_ctx.py $c 2>&1 | grep conditions; done | tee $OUT/te_host_calls_conditions.txt
;;
s21)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s21; mkdir -p $OUT
timeout 600 python tools/bench_proofs.py --config poseidon > $OUT/proofs.json 2>$OUT/proofs.err; tail -2 $OUT/proofs.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_s21/proofs.json').read().strip().splitlines()[-1])['poseidon']
print(json.dumps(d['verify_all_leaves_dev'])); print(json.dumps(d['verify_paths']))
PY
timeout 900 python -m pytest tests/test_abi.py tests/test_gpu_tree_handle.py tests/test_gpu_multi_slots.py tests/test_gpu_poseidon.py -m gpu -x -q 2>&1 | tail -4
;;
s22)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s22; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -8 | tee $OUT/pytest_gpu_full.txt
;;
s23)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s23; mkdir -p $OUT
for A in 0 1; do AKP_POSEIDON_STAGED_IO=$A python tools/gpu_poseidon_hostpath.py 2>&1 | grep -v amdgpu.ids; done | tee $OUT/poseidon_hostpath_staged_ab.txt
timeout 600 python -m pytest tests/test_gpu_poseidon.py tests/test_gpu_canaries.py -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -3
;;
s24)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s24; mkdir -p $OUT
AKP_LIB=$GRAFT_REPO_ROOT/crypto_primitives_amd/lib/libakp_testhooks.so timeout 900 python -m pytest tests/test_gpu_multi_slots.py -m gpu -x -q -p no:cacheprovider -k "config5_shape" --durations=3 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -15 | tee $OUT/pytest.txt
;;
s25)
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
;;
s27)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s27; mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_s27/bench.json') if l.startswith('{')][-1])
print('value %.4g' % d['value']); print(json.dumps(d['sweep']['points'])[:1500]); print(json.dumps(d['merkle']['one_process_c_abi']['resident_tree'])[:500])
PY
timeout 900 python -m pytest tests/test_gpu_tree_handle.py tests/test_gpu_bench_contract.py -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -3
;;
s28)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s28; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -20 | tee $OUT/pytest_gpu_full.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
;;
s29)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s29; mkdir -p $OUT
AKP_LIB=$GRAFT_REPO_ROOT/crypto_primitives_amd/lib/libakp_testhooks.so timeout 900 python -m pytest tests/test_gpu_multi_slots.py -m gpu -x -q -p no:cacheprovider --durations=15 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -24 | tee $OUT/pytest_hooks_child.txt
;;
s32)
# wide curve tables (24-bit digits / 8-chunk groups): full suite, counters of the curve kernels, bench.py alone and under rocprofv3
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s32; mkdir -p $OUT
(timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -12) > $OUT/pytest_gpu_full.txt; tail -3 $OUT/pytest_gpu_full.txt
PMC_COUNTERS="FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU VALUBusy MemUnitStalled" bash tools/gpu_pmc_r4.sh r04_s32 te > /dev/null 2>&1; cat $OUT/pmc_te.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-path --sustain-seconds 0 --no-sweep --proofs-log2 0 > $OUT/bench_under_rocprofv3.json 2> $OUT/bench_under_rocprofv3.err
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/rocprof_kernel_stats_bench_py.csv; head -8 $f | cut -c1-200; rm -rf $OUT/prof
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_s32/bench.json') if l.startswith('{')][-1])
r=d['roofline']
print('value %.4g ms/step %.3f eff %.1f power %s' % (d['value'], d['ms_per_step'], r['effective_sclk_mhz'] or 0, r['power_w_under_load']))
print('pedersen', d['pedersen']['hashes_per_s'], d['pedersen']['ms_per_batch'], json.dumps(d['pedersen']['roofline']['table']), d['pedersen'].get('sustained'))
print('bh', d['bh_merkle']['seconds'], d['bh_merkle']['leaves_per_s'])
print('proofs bh', json.dumps(d['proofs']['bh'])[:700])
print('host', {k: v.get('ms_per_batch') for k, v in d['host_path'].items() if isinstance(v, dict)})
PY
;;
s34)
# after the packed pair serialisation: full suite, bench.py alone and under rocprofv3
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s34; mkdir -p $OUT
(timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -12) > $OUT/pytest_gpu_full.txt; tail -3 $OUT/pytest_gpu_full.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-path --sustain-seconds 0 --no-sweep --proofs-log2 0 > $OUT/bench_under_rocprofv3.json 2> $OUT/bench_under_rocprofv3.err
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/rocprof_kernel_stats_bench_py.csv; rm -rf $OUT/prof
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.err
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_s34/bench.json') if l.startswith('{')][-1])
r=d['roofline']
print('value %.4g ms/step %.3f eff %.1f power %s' % (d['value'], d['ms_per_step'], r['effective_sclk_mhz'] or 0, r['power_w_under_load']))
p=d['pedersen']; print('pedersen', p['hashes_per_s'], p['ms_per_batch'], p['roofline']['traffic_over_algorithmic'], p['roofline']['valu']['frac_of_mad_issue_peak'], json.dumps(p.get('sustained')))
b=d['bh_merkle']; print('bh', b['seconds'], b['leaves_per_s'], b['roofline']['traffic_over_algorithmic'], b['roofline']['valu']['frac_of_mad_issue_peak'])
print('sweep', {k: (round(v['bh_tree_ms'],2), round(v['tree_ms'],2)) for k, v in d['sweep']['points'].items()})
print('host', {k: v.get('ms_per_batch') for k, v in d['host_path'].items() if isinstance(v, dict)})
print('predicted', json.dumps(d['predicted_scaling']['bh_merkle_weak'])[:400])
PY
;;
s39)
# final state of round 4 (folded tail constant, pipelined finalize loads): full suite, bench.py alone and under rocprofv3
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s39; mkdir -p $OUT
(timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -12) > $OUT/pytest_gpu_full.txt; tail -3 $OUT/pytest_gpu_full.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-path --sustain-seconds 0 --no-sweep --proofs-log2 0 > $OUT/bench_under_rocprofv3.json 2> $OUT/bench_under_rocprofv3.err
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/rocprof_kernel_stats_bench_py.csv; rm -rf $OUT/prof
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.err
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_s39/bench.json') if l.startswith('{')][-1])
r=d['roofline']
print('value %.4g ms/step %.3f eff %.1f power %s' % (d['value'], d['ms_per_step'], r['effective_sclk_mhz'] or 0, r['power_w_under_load']))
p=d['pedersen']; print('pedersen', p['hashes_per_s'], p['ms_per_batch'], p['roofline']['traffic_over_algorithmic'], p['roofline']['valu']['frac_of_mad_issue_peak'], json.dumps(p.get('sustained')))
b=d['bh_merkle']; print('bh', b['seconds'], b['leaves_per_s'], b['roofline']['traffic_over_algorithmic'], b['roofline']['valu']['frac_of_mad_issue_peak'])
print('sweep', {k: (round(v['bh_tree_ms'],2), round(v['tree_ms'],2)) for k, v in d['sweep']['points'].items()})
print('host', {k: v.get('ms_per_batch') for k, v in d['host_path'].items() if isinstance(v, dict)})
print('predicted', json.dumps(d['predicted_scaling']['bh_merkle_weak'])[:400])
PY
;;
s40)
# final state of round 4 (tables built for the message lengths that arrive): full suite, bench.py alone and under rocprofv3
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s40; mkdir -p $OUT
(timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -12) > $OUT/pytest_gpu_full.txt; tail -3 $OUT/pytest_gpu_full.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-path --sustain-seconds 0 --no-sweep --proofs-log2 0 > $OUT/bench_under_rocprofv3.json 2> $OUT/bench_under_rocprofv3.err
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/rocprof_kernel_stats_bench_py.csv; rm -rf $OUT/prof
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.err
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_s40/bench.json') if l.startswith('{')][-1])
r=d['roofline']
print('value %.4g ms/step %.3f eff %.1f power %s' % (d['value'], d['ms_per_step'], r['effective_sclk_mhz'] or 0, r['power_w_under_load']))
p=d['pedersen']; print('pedersen', p['hashes_per_s'], p['ms_per_batch'], p['roofline']['traffic_over_algorithmic'], p['roofline']['valu']['frac_of_mad_issue_peak'], json.dumps(p.get('sustained')))
b=d['bh_merkle']; print('bh', b['seconds'], b['leaves_per_s'], b['roofline']['traffic_over_algorithmic'], b['roofline']['valu']['frac_of_mad_issue_peak'])
print('sweep', {k: (round(v['bh_tree_ms'],2), round(v['tree_ms'],2)) for k, v in d['sweep']['points'].items()})
print('host', {k: v.get('ms_per_batch') for k, v in d['host_path'].items() if isinstance(v, dict)})
print('predicted', json.dumps(d['predicted_scaling']['bh_merkle_weak'])[:400])
PY
;;
*) echo "usage: $0 s1 .. s40"; exit 2;;
esac
