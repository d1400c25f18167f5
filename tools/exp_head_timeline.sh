#!/bin/bash
# usage (GPU box, repo root): tools/exp_head_timeline.sh
# One steady-state step of the head's stream in the pipelined C2 bench: every kernel with its duration and the gap in front of it.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/head_timeline
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 $root/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-kernel-timing > $out/log 2>&1 || { tail -3 $out/log; exit 1; }
f=$(find $out/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
hq = max(byq, key=lambda q: sum("k_head_stage" in r["Kernel_Name"] for r in byq[q]))
# steps are delimited by k_adamw (the last kernel of a head update), wherever it runs
adam = [r for r in rows if r["Kernel_Name"].startswith("k_adamw")]
t_a, t_b = int(adam[-3]["End_Timestamp"]), int(adam[-2]["End_Timestamp"])      # one whole update near the end of the run
print("update period %.1f us; head-stream queue %s" % ((t_b - t_a) / 1e3, hq))
names = ("k_head", "k_tile", "k_fuse", "k_out_bwd", "k_colsum", "k_sqnorm", "k_adamw", "k_copy_group", "k_roi")
step = [r for r in rows if t_a < int(r["End_Timestamp"]) <= t_b and (r["Queue_Id"] == hq or r["Kernel_Name"].startswith(names))]
prev_end = t_a
tot_k = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  dur %6.1f  gap %6.1f  q%s  %s  grid %s" % ((s - t_a) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Queue_Id"], r["Kernel_Name"][:40], r["Grid_Size_X"]))
    tot_k += e - s; prev_end = e
print("kernels %.1f us, %d launches" % (tot_k / 1e3, len(step)))
PY
rm -rf $out/t
