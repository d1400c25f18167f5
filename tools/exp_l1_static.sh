#!/bin/bash
# usage (GPU box, repo root): tools/exp_l1_static.sh <config> <pct> [<pct> ...]
# Layer-1 convolution times against the share of a wave's work items that it takes in a fixed order before it turns to
# the shared counter (EVENTAD_L1_STATIC_PCT, read once per process by the library).
cfg=$1; shift
for p in "$@"; do
  echo "== $cfg static share $p %"
  EVENTAD_L1_STATIC_PCT=$p timeout -k 10 150 python3 tools/stage_times.py $cfg 2>/dev/null | grep -E "l1_conv" || exit 1
done
