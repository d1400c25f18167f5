#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_passes.sh <tag> <config>
# Two SQ counter passes + FETCH_SIZE + WRITE_SIZE over tools/run_backbone_once.py, each in a run of its own.
tag=$1; cfg=${2:-C2}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" -d $out/$name --output-format csv -- python3 $root/tools/run_backbone_once.py $cfg > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/$name.log; exit 1; }; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS &&
run sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_ANY &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
cd $root && python3 tools/pmc_kernel.py $out/sq1 $out/sq2 > $out/sq_summary.txt && python3 tools/pmc_traffic.py $out/hbm_traffic.json $out/fetch $out/write > $out/traffic_summary.txt
find $out -name "*.csv" ! -name "*counter_collection.csv" -delete 2>/dev/null; find $out -name "*counter_collection.csv" -size +20M -delete
echo done
