# usage (GPU box, repo root): tools/exp_gg64.sh [masks ...]  -- C2 step against which of the head's six batched GEMM launches
# run on 64x64 tiles (EVENTAD_GG64 bit i = launch i in program order: input projections, fusion projections, fusion.0,
# fusion.0 backward, projection backward, recurrent weight gradients)
masks=${@:-"0 32 40 56 59 63 0 32"}
for m in $masks; do
  echo "GG64=$m"; EVENTAD_GG64=$m timeout -k 10 120 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
done
