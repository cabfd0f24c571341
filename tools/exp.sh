#!/bin/bash
# builds experiment variants of the library into exp/ (git-ignored): tools/exp.sh "<name> <extra flags>" ...
set -e
cd "$(dirname "$0")/../flute_amd/csrc"
mkdir -p ../../exp
make -s >/dev/null 2>&1
for spec in "$@"; do
  name=${spec%% *}; flags=${spec#* }
  for f in api ggm qc lutsetup alsz shard; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -w $flags -c $f.hip -o /tmp/exp_$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../exp/lib_$name.so /tmp/exp_api.o /tmp/exp_ggm.o /tmp/exp_qc.o /tmp/exp_lutsetup.o /tmp/exp_alsz.o /tmp/exp_shard.o
  echo built exp/lib_$name.so
done
