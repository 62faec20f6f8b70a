#!/bin/bash
# HBM traffic of the headline call from the L2 counters, the way MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE
# in separate rocprofv3 passes (no tracing domains beside --kernel-trace), calibrated on a kernel of known traffic
# (4096 stand-alone forward transforms of 2^14 points: 512 MiB read, 512 MiB written).
# usage (on the GPU box): bash tools/pmc_traffic.sh <tag>   ->  gpurun_out/pmc_<tag>/{fetch,write,cal_fetch,cal_write}/...
set -e
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o fetch --output-format csv -- python3 bench.py --batch 1024 --steps 2 --warmup 1 --no-cpu --no-60bit --no-bfv > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o write --output-format csv -- python3 bench.py --batch 1024 --steps 2 --warmup 1 --no-cpu --no-60bit --no-bfv > $out/write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/cal_fetch -o cal --output-format csv -- python3 tools/ntt_bench.py 14 4096 2 1 > $out/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/cal_write -o cal --output-format csv -- python3 tools/ntt_bench.py 14 4096 2 1 > $out/cal_write.log 2>&1
python3 tools/pmc_traffic.py $out
