#!/bin/bash
# size sweep on the GPU box: bench.py at n = 2^20 .. 2^26 -> gpurun_out/sweep_<tag>_<L>.json
TAG=${1:-run}
for L in 20 21 22 23 24 25 26; do
  timeout -k 10 200 python bench.py --log2n $L --steps 5 --warmup 1 --no-cpu-baseline --lut-log2b 0 --alsz-log2n 0 > gpurun_out/sweep_${TAG}_$L.json 2> gpurun_out/sweep_${TAG}_$L.err || { echo "L=$L failed"; tail -3 gpurun_out/sweep_${TAG}_$L.err; exit 1; }
  python - $L gpurun_out/sweep_${TAG}_$L.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2]))
k=d["kernels_ms_per_step"]
print(sys.argv[1], "ms/step %.3f"%d["ms_per_step"], "OT/s %.3e"%d["value"], "kernel sum %.3f"%sum(k.values()), {a:round(b,3) for a,b in sorted(k.items(), key=lambda x:-x[1])[:8]})
PY
done
