#!/bin/bash
# builds exp/lib_<name>.so from ONE recompiled translation unit + the other objects of the regular build:
#   tools/exp_one.sh <name> <unit (ggm|qc|alsz|...)> "<extra flags>"
set -e
cd "$(dirname "$0")/../flute_amd/csrc"
mkdir -p ../../exp
make -s >/dev/null 2>&1
name=$1; unit=$2; flags=$3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -w $flags -c $unit.hip -o /tmp/expone_$name.o
objs=""
for f in api ggm qc lutsetup alsz shard; do
  if [ $f = $unit ]; then objs="$objs /tmp/expone_$name.o"; else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../exp/lib_$name.so $objs
echo built exp/lib_$name.so
