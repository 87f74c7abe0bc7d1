#!/bin/bash
# on the GPU box: the headline step under the three wait policies, same box, interleaved (tools/headline_frame.py)
for rep in 1 2 3; do
  for v in "default" "AG2_POLL_SPIN_US=100000" "AG2_POLL=0"; do
    if [ "$v" = default ]; then r=$(python tools/headline_frame.py stepwise 2>/dev/null | tail -1); else r=$(env $v python tools/headline_frame.py stepwise 2>/dev/null | tail -1); fi
    echo "$v $r"
  done
done
