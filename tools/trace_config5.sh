#!/bin/bash
# Kernel trace of config 5 (32 circuits) on the 49-bit (default) or 55-bit chain.  usage: bash tools/trace_config5.sh <tag> [bits]
set -e
tag=${1:-run}
bits=${2:-49}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/trace_$tag
mkdir -p $out
ABC_CONFIG5_BITS=$bits rocprofv3 --kernel-trace -d $out/c5 -o c5 -- python3 tools/run_configs.py --config 5 --batch 32 > $out/c5.log 2>&1
python3 tools/rocpd_stats.py $(ls $out/c5/*/*.db $out/c5/*.db 2>/dev/null | head -1) --csv $out/config5_b32_${bits}bit_kernel_stats.csv > $out/c5.txt
rm -rf $out/c5
head -16 $out/c5.txt
