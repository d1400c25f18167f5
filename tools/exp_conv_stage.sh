for c in C2 C4 C5; do
  timeout -k 10 300 python bench.py --config $c --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', d['ms_per_step'], {k:v for k,v in d['stage_ms'].items() if 'l1_conv' in k})" || exit 1
done
