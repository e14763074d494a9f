#!/bin/bash
# usage: scripts/build_variant.sh NAME "-DFOO=1 ..."   -> irspack_amd/variants/libirspack_amd_NAME.so
set -e
cd "$(dirname "$0")/../irspack_amd/csrc"
mkdir -p ../variants /tmp/irs_var_$1
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -mllvm -pragma-unroll-threshold=65536 -Xarch_host -ffp-contract=off $2"
for f in ials knn evaluator ceilings device_sort; do
  if [ "$f" = "${VARIANT_FILE:-ials}" ] || [ ! -f /tmp/irs_var_base/$f.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o /tmp/irs_var_$1/$f.o &
  else
    cp /tmp/irs_var_base/$f.o /tmp/irs_var_$1/$f.o
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libirspack_amd_$1.so /tmp/irs_var_$1/*.o
echo built ../variants/libirspack_amd_$1.so
