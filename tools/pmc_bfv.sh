#!/bin/bash
# HBM traffic per kernel of a BFV mul+relin on the reference's default ring (same recipe as tools/pmc_traffic.sh)
# usage (on the GPU box): bash tools/pmc_bfv.sh <tag> [n] [batch]
set -e
tag=${1:-bfv}; n=${2:-16384}; B=${3:-128}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o fetch --output-format csv -- python3 tools/bfv_profile.py $n $B > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o write --output-format csv -- python3 tools/bfv_profile.py $n $B > $out/write.log 2>&1
python3 - $out $B <<'PY'
import sys, json
sys.path.insert(0, "tools")
from pmc_traffic import collect, short
out, B = sys.argv[1], int(sys.argv[2])
f, w = collect(out + "/fetch", "FETCH_SIZE"), collect(out + "/write", "WRITE_SIZE")
res = {}
for name in sorted(f):
    if "abc::" not in name: continue
    fs, n = f[name]; ws, _ = w.get(name, (0.0, n))
    # KiB over the run; 5 calls of B pairs; FETCH_SIZE corrected by the factor measured for 8-byte loads (1.885)
    res[short(name)] = {"launches": n, "read_MB_per_pair": fs * 1.885 * 1024 / 5 / B / 1e6, "write_MB_per_pair": ws * 1024 / 5 / B / 1e6}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -(kv[1]["read_MB_per_pair"] + kv[1]["write_MB_per_pair"]))[:12]:
    print("%-60s launches %3d  read %7.2f MB  write %7.2f MB per pair" % (k[:60], v["launches"], v["read_MB_per_pair"], v["write_MB_per_pair"]))
PY
