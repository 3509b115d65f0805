OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s14; mkdir -p $OUT
export AKP_LIB=$GRAFT_REPO_ROOT/crypto_primitives_amd/lib/libakp_testhooks.so
for K in "byte_digests" "resident_tree_poseidon" "sharded_build_logic"; do
  echo "== -k $K"; PYTHONFAULTHANDLER=1 timeout 600 python -m pytest tests/test_gpu_multi_slots.py -m gpu -x -q -p no:cacheprovider -k "$K" 2>&1 | tail -12; echo "rc=$?"
done 2>&1 | tee $OUT/exit_crash.txt
unset AKP_LIB
ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>/dev/null | head -40 > $OUT/hwmon.txt; for f in /sys/class/drm/card*/device/hwmon/hwmon*/power1_*; do echo "$f $(cat $f 2>/dev/null)"; done >> $OUT/hwmon.txt 2>&1
(rocm-smi --showpower --json; rocm-smi --showtemp --json; which amd-smi && amd-smi metric --json | head -c 3000) >> $OUT/hwmon.txt 2>&1
cat $OUT/hwmon.txt | head -80
