#!/bin/bash
# round 5: one point per lane against two points per lane in the fused gated kernel (TE_FUSED_ITEMS), A B A B on one box.
#   make -C crypto_primitives_amd/csrc testhooks && make -C crypto_primitives_amd/csrc arm NAME=fused1 DEF="-DAKP_TEST_HOOKS -DTE_FUSED_ITEMS=1"
out=gpurun_out/${1:-r05_s16}; mkdir -p $out
for round in 1 2; do
  for arm in fused1 testhooks; do
    AKP_LIB=$PWD/crypto_primitives_amd/lib/libakp_$arm.so GATE_KNOBS_ARMS=shipped,no_copies,shipped_again timeout 300 python tools/gpu_r5_gate_knobs.py > $out/fused_items_${arm}_$round.json 2>> $out/fused_items.err
  done
done
python - <<PY
import json
for arm in ("fused1", "testhooks"):
    for r in (1, 2):
        d = json.load(open("$out/fused_items_%s_%d.json" % (arm, r)))
        print(arm, r, {t: {k: v["ms_median"] for k, v in d[t].items()} for t in ("cache_sized", "hbm_sized")})
PY
