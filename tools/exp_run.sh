#!/bin/bash
# runs bench.py against each exp/lib_<name>.so given (GPU box); prints ms/step and per-kernel ms
for name in "$@"; do
  FLUTE_SOT_LIB=$PWD/exp/lib_$name.so timeout -k 10 200 python bench.py --no-cpu-baseline --lut-log2b 0 --alsz-log2n 0 --steps 5 --warmup 1 > gpurun_out/exp_$name.json 2> gpurun_out/exp_$name.err || { echo "$name failed"; tail -3 gpurun_out/exp_$name.err; exit 1; }
  python - "$name" <<'PY'
import json,sys
d=json.load(open("gpurun_out/exp_%s.json"%sys.argv[1]))
print(sys.argv[1], round(d["ms_per_step"],3), {k:v for k,v in d["kernels_ms_per_step"].items() if v>0.3})
PY
done
