#!/bin/bash
# Kernel trace of the headline call (bench.py at batch 1024): per-kernel durations with one lane (kernels run alone) and with the
# default two lanes.  usage (on the GPU box): bash tools/trace_hot.sh <tag>  ->  gpurun_out/trace_<tag>/{onelane,twolane}_kernel_stats.csv
set -e
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/trace_$tag
mkdir -p $out
ABC_HIP_LANES=1 rocprofv3 --kernel-trace -d $out/one -o one -- python3 bench.py --batch 1024 --steps 5 --warmup 2 --no-cpu --no-60bit --no-bfv > $out/one.log 2>&1
python3 tools/rocpd_stats.py $(ls $out/one/*/*.db $out/one/*.db 2>/dev/null | head -1) --csv $out/onelane_kernel_stats.csv > $out/onelane.txt
rocprofv3 --kernel-trace -d $out/two -o two -- python3 bench.py --batch 1024 --steps 5 --warmup 2 --no-cpu --no-60bit --no-bfv > $out/two.log 2>&1
python3 tools/rocpd_stats.py $(ls $out/two/*/*.db $out/two/*.db 2>/dev/null | head -1) --csv $out/twolane_kernel_stats.csv > $out/twolane.txt
rm -rf $out/one $out/two
head -8 $out/onelane.txt
