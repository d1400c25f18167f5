"""Timeline of one steady-state step from a rocprofv3 --kernel-trace CSV: per queue, kernel start/duration and the gap
to the previous kernel of the same queue.  usage: python tools/timeline.py <kernel_trace.csv> [step_index_from_end]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# a step starts at every k_evad_zero / k_prep_events of the backbone graph
marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_prep_events")]
i0, i1 = marks[-back - 1], marks[-back]
t0 = rows[i0]["s"]
seg = [r for r in rows if rows[i0]["s"] - 5000 <= r["s"] < rows[i1]["s"] - 5000]
print("step length %.1f us, %d kernels" % ((rows[i1]["s"] - rows[i0]["s"]) / 1e3, len(seg)))
last = {}
busy = {}
for r in seg:
    q = r["Queue_Id"]
    gap = (r["s"] - last[q]) / 1e3 if q in last else 0.0
    last[q] = r["e"]
    busy[q] = busy.get(q, 0) + (r["e"] - r["s"]) / 1e3
    print("q%-3s t=%8.1f  dur %7.1f  gap %6.1f  %s" % (q, (r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, gap, r["Kernel_Name"][:60]))
print("busy per queue (us):", {q: round(v, 1) for q, v in busy.items()})
