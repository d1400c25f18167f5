# C2 step against the number of hardware queues of the process and the number of backbone chains
for q in 4 8; do for n in 3 4 5; do
  echo "GPU_MAX_HW_QUEUES=$q CHAINS=$n"
  GPU_MAX_HW_QUEUES=$q EVENTAD_BACKBONE_CHAINS=$n timeout -k 10 120 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
done; done
