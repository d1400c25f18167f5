#!/bin/bash
# usage (GPU box, repo root): bash tools/exp_head_alone.sh  -- per-kernel averages of the head update replayed ALONE (whole chip, no
# backbone beside it), to set against the per-kernel durations of the pipelined step (tools/exp_head_timeline.sh)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/head_alone
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cp $root/tools/stage_graphs.py /tmp/stage_graphs_run.py
cd $root
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python3 tools/stage_graphs.py C2 head > $out/log 2>&1 || { tail -3 $out/log; exit 1; }
f=$(find $out/t -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("%-50s calls %5s avg %8.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $out/t
