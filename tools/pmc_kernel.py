"""Reduces rocprofv3 --pmc counter_collection CSVs to per-kernel averages per launch (first launch dropped).
usage: python tools/pmc_kernel.py <dir> [<dir> ...] [--filter substring]"""
import collections, csv, glob, sys
flt = None
dirs = []
it = iter(sys.argv[1:])
for a in it:
    if a == "--filter":
        flt = next(it)
    else:
        dirs.append(a)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if flt and flt not in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    row = []
    for c, v in sorted(acc[k].items()):
        v = v[1:] if len(v) > 1 else v
        row.append("%s=%.4g" % (c, sum(v) / len(v)))
    print("%-44s %s" % (k[:44], "  ".join(row)))
