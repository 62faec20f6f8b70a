#!/bin/bash
# Kernel trace of one of the configs at its stated batch.  usage: bash tools/trace_config.sh <tag> [config = 4]
set -e
tag=${1:-run}
cfg=${2:-4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/trace_$tag
mkdir -p $out
rocprofv3 --kernel-trace -d $out/cx -o cx -- python3 tools/run_configs.py --config $cfg > $out/cx.log 2>&1
python3 tools/rocpd_stats.py $(ls $out/cx/*/*.db $out/cx/*.db 2>/dev/null | head -1) --csv $out/config${cfg}_kernel_stats.csv > $out/cx.txt
rm -rf $out/cx
head -16 $out/cx.txt
