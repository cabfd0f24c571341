#!/bin/bash
# GPU box: bit-exact scale tests (n = 2^23 .. 2^26 both roles) + one bench line; prints ms/step and per-kernel ms
set -o pipefail
TAG=${1:-q}
timeout -k 10 500 python -u -m pytest tests/test_gpu_scale.py tests/test_gpu_fafft.py -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --lut-log2b 0 --alsz-log2n 0 > gpurun_out/$TAG.json 2> gpurun_out/$TAG.err || { tail -3 gpurun_out/$TAG.err; exit 2; }
python - $TAG <<'PY'
import json,sys
d=json.loads(open('gpurun_out/%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step %.3f'%d['ms_per_step'], 'OT/s %.4e'%d['value'])
print({k:v for k,v in d['kernels_ms_per_step'].items() if v>0.2})
PY
