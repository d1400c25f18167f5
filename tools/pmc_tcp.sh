#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_tcp.sh <tag> <config> [kernel-name filter]
# Vector-memory path counters (L1 accesses, L2 requests, stalls, address-unit busy) over tools/run_backbone_once.py.
tag=$1; cfg=${2:-C3}; flt=${3:-k_}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/tcp_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" -d $out/$name --output-format csv -- python3 $root/tools/run_backbone_once.py $cfg > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/$name.log; exit 1; }; }
run t1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE &&
run t2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM &&
cd $root && python3 tools/pmc_kernel.py $out/t1 $out/t2 --filter "$flt" > $out/summary.txt
find $out -name "*.csv" -delete 2>/dev/null
cat $out/summary.txt
