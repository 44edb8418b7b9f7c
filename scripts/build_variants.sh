#!/bin/bash
# Developer helper: build variants of the library with extra -D flags for the UNet kernels file into humanliff_amd/exp/lib_<name>.so
# usage: bash scripts/build_variants.sh name1:"-DX -DY" name2:"-DZ" ...
cd "$(dirname "$0")/.."
python -m humanliff_amd.build > /dev/null || exit 1
mkdir -p humanliff_amd/exp
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $flags -c humanliff_amd/csrc/hl_unet_kernels.hip -o humanliff_amd/exp/k_$name.o || exit 1
  objs=$(ls humanliff_amd/build/*.o | grep -v hl_unet_kernels)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o humanliff_amd/exp/lib_$name.so $objs humanliff_amd/exp/k_$name.o || exit 1
  rm humanliff_amd/exp/k_$name.o
  echo built $name
done
