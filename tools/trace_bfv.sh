#!/bin/bash
# Kernel trace of BFV mul+relin on BFVDefault(16384) at batch 256.  usage: bash tools/trace_bfv.sh <tag> [env assignments...]
set -e
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/trace_$tag
mkdir -p $out
rocprofv3 --kernel-trace -d $out/bfv -o bfv -- python3 tools/bfv_profile.py 16384 256 > $out/bfv.log 2>&1
python3 tools/rocpd_stats.py $(ls $out/bfv/*/*.db $out/bfv/*.db 2>/dev/null | head -1) --csv $out/bfv16384_b256_kernel_stats.csv > $out/bfv.txt
rm -rf $out/bfv
head -12 $out/bfv.txt
