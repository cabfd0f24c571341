#!/bin/bash
# per-kernel ms per step of the QC stage for the tree's library and for each exp/lib_<name>.so given, then the bit-exact
# headline test with the LAST one (GPU box): tools/quick_qc.sh name1 name2 ...
run() {
  timeout -k 10 100 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --lut-log2b 0 --alsz-log2n 0 | python -c "
import json,sys
d=json.load(sys.stdin); k=d['kernels_ms_per_step']
print('$1', round(d['ms_per_step'],3), {a:round(b,3) for a,b in k.items() if b>0.2})"
}
run tree
for n in "$@"; do FLUTE_SOT_LIB=$PWD/exp/lib_$n.so run $n; done
last=${@: -1}
if [ -n "$last" ]; then FLUTE_SOT_LIB=$PWD/exp/lib_$last.so timeout -k 10 200 python -m pytest tests/test_gpu_scale.py -x -q -k "headline or fused_conversion or 2pow20" 2>&1 | tail -2; fi
