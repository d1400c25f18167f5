#!/bin/bash
# usage (GPU box, repo root): tools/profile_config.sh <round-tag> <config> <steps>
# bench line + rocprofv3 kernel stats of the same command + PMC passes (SQ, FETCH_SIZE, WRITE_SIZE) over the backbone.
tag=$1; cfg=$2; steps=${3:-10}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_${tag}_${cfg}
mkdir -p $out
cd $root
extra=""; [ "$cfg" != "C2" ] && extra="--no-cpu-baseline"
timeout -k 10 400 python bench.py --config $cfg --steps $steps --warmup 3 $extra > $out/bench.json 2> $out/bench.err || { echo bench failed; tail -5 $out/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --config $cfg --steps $steps --warmup 3 --no-cpu-baseline --no-kernel-timing > $out/stats.log 2>&1 || { echo stats failed; tail -5 $out/stats.log; exit 1; }
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/stats
cd $root && tools/pmc_passes.sh ${tag}_${cfg} $cfg
echo profile done
