#!/bin/bash
# Kernel trace of BFV multiply + relinearise on BFVDefault(2^logn).  usage: bash tools/trace_bfv_default.sh <tag> [logn = 15] [batch = 64] [op = mul_relin]
set -e
tag=${1:-run}
logn=${2:-15}
batch=${3:-64}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/trace_$tag
mkdir -p $out
rocprofv3 --kernel-trace -d $out/bd -o bd -- python3 tools/ab_ops.py --logn $logn --bfv-default --op ${4:-mul_relin} --batch $batch --rounds 2 --steps 3 > $out/bd.log 2>&1
python3 tools/rocpd_stats.py $(ls $out/bd/*/*.db $out/bd/*.db 2>/dev/null | head -1) --csv $out/bfv_default_${logn}_kernel_stats.csv > $out/bd.txt
rm -rf $out/bd
head -16 $out/bd.txt
