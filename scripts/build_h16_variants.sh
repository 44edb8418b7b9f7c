#!/bin/bash
# Developer helper: variants of hl_conv_h16.hip (-D flags) linked into humanliff_amd/exp/lib_<name>.so   usage: name1:"-DH16_ABL=1" ...
cd "$(dirname "$0")/.."
python -m humanliff_amd.build > /dev/null || exit 1
mkdir -p humanliff_amd/exp
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -Wno-unused-result $flags -c humanliff_amd/csrc/hl_conv_h16.hip -o humanliff_amd/exp/k_$name.o || exit 1
  objs=$(ls humanliff_amd/build/*.o | grep -v hl_conv_h16)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o humanliff_amd/exp/lib_$name.so $objs humanliff_amd/exp/k_$name.o || exit 1
  rm humanliff_amd/exp/k_$name.o
  echo built $name
done
