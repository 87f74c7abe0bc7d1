#!/bin/bash
# on the GPU box: tools/ab_run.sh <config> name1 name2 ...   (stage times of every variant, same box)
cfg=$1; shift
for name in "$@"; do
  lib=agile_grasp2_amd/csrc/exp/libag2hip_${name}.so
  [ "$name" = base ] && lib=agile_grasp2_amd/csrc/libag2hip.so
  for rep in 1 2; do
    AG2_LIB=$PWD/$lib python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('$name', 'value %.0f' % d['value'], 'ms %.4f' % d['ms_per_step'], 'sweep %.4f ovf %.4f normals %.4f conv %.4f fc %.4f render %.4f grid %.4f' % (s['sweep_ms'], s['sweep_overflow_ms'], s['normals_ms'], s['lenet_conv_ms'], s['lenet_fc_ms'], s['render_ms'], s['grid_ms']))"
  done
done
