#!/bin/bash
# kernel trace of a short pipelined bench run: which queue the runtime's copyBuffer launches sit on and what runs around them
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/trace_copies
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 $root/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-timing > $out/log 2>&1 || { tail -3 $out/log; exit 1; }
f=$(find $out/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-900:]
byq = collections.defaultdict(list)
for r in tail:
    byq[(r.get("Queue_Id"), r.get("Stream_Id"))].append(r)
for q, rs in byq.items():
    print("queue/stream", q, "kernels", len(rs))
    names = [r["Kernel_Name"][:40] for r in rs]
    for i, r in enumerate(rs):
        if "copyBuffer" in r["Kernel_Name"] and i > 0 and i + 1 < len(rs):
            print("   ", names[i - 1], "->", names[i], "(%d ns)" % (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])), "->", names[i + 1])
    print()
# steady state: busy share and launch gaps per queue over the last 8 steps' kernels
for q, rs in byq.items():
    if len(rs) < 50: continue
    t0, t1 = int(rs[0]["Start_Timestamp"]), int(rs[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rs, rs[1:])]
    small = [g for g in gaps if 0 <= g < 20000]
    print("queue", q, "span %.3f ms busy %.1f %% kernels %d; gaps < 20 us: n %d mean %.2f us; gaps >= 20 us: n %d total %.1f us" % (
        (t1 - t0) / 1e6, 100.0 * busy / (t1 - t0), len(rs), len(small), sum(small) / max(len(small), 1) / 1e3,
        len([g for g in gaps if g >= 20000]), sum(g for g in gaps if g >= 20000) / 1e3))
PY
rm -rf $out/t
