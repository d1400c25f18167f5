#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_mfma.sh <tag> <config>
# One counter pass (MFMA-busy share and fp32 MFMA operation count per kernel) over tools/run_backbone_once.py.
tag=$1; cfg=${2:-C3}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/mfma_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --kernel-trace --pmc MfmaUtil SQ_INSTS_VALU_MFMA_MOPS_F32 -d $out/pass --output-format csv -- python3 $root/tools/run_backbone_once.py $cfg > $out/pass.log 2>&1 || { echo "pass failed"; tail -3 $out/pass.log; exit 1; }
cd $root && python3 tools/pmc_kernel.py $out/pass > $out/mfma_util.txt
find $out -name "*.csv" -delete 2>/dev/null
grep -E "k_conv|k_gemm" $out/mfma_util.txt
