#!/bin/bash
# A/B builds of one translation unit: tools/ab_build.sh <file.hip> name1:"-DFLAG ..." name2:"..." ...
# -> agile_grasp2_amd/csrc/exp/libag2hip_<name>.so (select with AG2_LIB=...); *.so / *.o are git-ignored.
set -e
cd "$(dirname "$0")/../agile_grasp2_amd/csrc"
make -s -j8
unit=$1; shift
base=${unit%.hip}
others=$(ls *.o | grep -v "^${base}.o$")
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -I. \
      -Wall -Wno-unused-function -Wno-unused-result $flags -c $unit -o exp/${base}_${name}.o &
done
wait
for spec in "$@"; do
  name=${spec%%:*}
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o exp/libag2hip_${name}.so $others exp/${base}_${name}.o
done
ls -la exp/*.so
