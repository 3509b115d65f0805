for D in 16 15 14 13; do
  AKP_PEDERSEN_DIGIT_BITS=$D python bench.py --steps 3 --warmup 1 --merkle-log2 0 --proofs-log2 0 --bh-merkle-log2 0 --no-cpu-baseline --no-host-path --sustain-seconds 4 --sustain-log2-big 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pedersen']
print('D=$D steps',p['roofline']['table']['steps'],'table MB',p['roofline']['table']['table_bytes']/1e6,'ms/batch %.3f'%p['ms_per_batch'],'sustained %.4g/s'%p['sustained']['hashes_per_s'],'sclk during',p['sustained']['sclk_level_mhz_during'],'| poseidon sclk',d['sustained']['2^20']['sclk_level_mhz_during'])"
done
rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk" | head -6
