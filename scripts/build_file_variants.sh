#!/bin/bash
# Developer helper: variants of ONE source file (-D flags) linked into humanliff_amd/exp/lib_<name>.so (HL_LIB_PATH selects one at run time)
# usage: bash scripts/build_file_variants.sh hl_render.hip name1:"-DHL_H2_K=4" name2:"-D..." ...
cd "$(dirname "$0")/.."
file=$1; shift
python -m humanliff_amd.build > /dev/null || exit 1
mkdir -p humanliff_amd/exp
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -Wno-unused-result $flags -c humanliff_amd/csrc/$file -o humanliff_amd/exp/k_$name.o || exit 1
  objs=$(ls humanliff_amd/build/*.o | grep -v "/$file.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o humanliff_amd/exp/lib_$name.so $objs humanliff_amd/exp/k_$name.o || exit 1
  rm humanliff_amd/exp/k_$name.o
  echo built $name
done
