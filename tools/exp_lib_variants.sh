#!/bin/bash
# usage (on the GPU box, from the repo root): tools/exp_lib_variants.sh <config> <filter> <variant> [<variant> ...]
# Runs tools/stage_times.py once per library build found under eventad_amd/lib/variants/<variant>.so (built by hand with
# different -D switches), each copied over the library for its run; prints the stage lines that match <filter>.
cfg=$1; flt=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
cp eventad_amd/lib/libeventad_hip.so /tmp/libeventad_hip.keep
for v in "$@"; do
  cp eventad_amd/lib/variants/$v.so eventad_amd/lib/libeventad_hip.so
  echo "== $v $cfg"
  timeout -k 10 150 python3 tools/stage_times.py $cfg 2>/dev/null | grep -E "$flt" || { echo "run failed"; cp /tmp/libeventad_hip.keep eventad_amd/lib/libeventad_hip.so; exit 1; }
done
cp /tmp/libeventad_hip.keep eventad_amd/lib/libeventad_hip.so
