# usage (GPU box, repo root): tools/exp_head_cus.sh [CUs ...]  -- C2 step against the CUs set aside for the head's stream,
# then against the number of backbone chains
cus=${@:-"24 32 40 48 64"}
for c in $cus; do
  echo "HEAD_CUS=$c"; EVENTAD_HEAD_CUS=$c timeout -k 10 120 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
done
for n in 2 4; do
  echo "CHAINS=$n"; EVENTAD_BACKBONE_CHAINS=$n timeout -k 10 120 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
done
