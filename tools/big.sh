#!/bin/bash
# large-size check on the GPU box: parity at 2^25 / 2^26, then bench at both sizes; optional kernel trace at 2^26
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_scale.py -x -q --timeout 200 --timeout-method thread 2>&1 | tail -2 || exit 1
for L in 25 26; do
  timeout -k 10 200 python bench.py --log2n $L --steps 3 --warmup 1 --no-cpu-baseline --lut-log2b 0 > gpurun_out/big_$L.json 2> gpurun_out/big_$L.err || { echo "L=$L failed"; tail -3 gpurun_out/big_$L.err; exit 1; }
  python - $L gpurun_out/big_$L.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2]))
k=d["kernels_ms_per_step"]
print(sys.argv[1], "ms/step %.3f"%d["ms_per_step"], "OT/s %.3e"%d["value"], {a:round(b,3) for a,b in sorted(k.items(), key=lambda x:-x[1])[:12]})
PY
done
if [ -n "$1" ]; then
  export TMPDIR=/tmp
  timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_big_$1 -o trace -- python3 bench.py --log2n 26 --steps 2 --warmup 1 --no-cpu-baseline --lut-log2b 0 --no-prof > gpurun_out/prof_big_$1.log 2>&1 || exit 2
  python - gpurun_out/prof_big_$1 <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<22: print(r['Name'][:60], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
fi
