#!/bin/bash
# usage: tools/kernel_resources.sh eventad_amd/csrc/layer1.hip  -> one line per kernel: VGPRs, spills, scratch, occupancy, LDS
f="$1"; shift
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wno-pass-failed -Xclang -target-feature -Xclang -packed-fp32-ops \
  --cuda-device-only -c -o /dev/null "$f" -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import re, sys
cur = {}
for line in sys.stdin:
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        if cur: print(cur)
        cur = {"name": t.split(":",1)[1].strip()}
    else:
        k, _, v = t.partition(":")
        if k.strip() in ("VGPRs", "AGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
            cur[k.strip()] = v.strip()
if cur: print(cur)
' | python3 -c '
import sys, subprocess, ast
for line in sys.stdin:
    d = ast.literal_eval(line)
    name = subprocess.run(["c++filt", d.pop("name")], capture_output=True, text=True).stdout.strip().split("(")[0]
    print("%-60s %s" % (name[-60:], "  ".join("%s=%s" % (k.split(" [")[0], v) for k, v in d.items())))
'

