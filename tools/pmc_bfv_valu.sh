#!/bin/bash
# VALU / wait / LDS counters per kernel of a BFV mul+relin on the reference's default ring (one counter pass each)
# usage (on the GPU box): bash tools/pmc_bfv_valu.sh <tag> [n] [batch]
set -e
tag=${1:-bfv}; n=${2:-16384}; B=${3:-256}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcv_$tag
mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace -d $out/insts -o insts --output-format csv -- python3 tools/bfv_profile.py $n $B > $out/insts.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $out/busy -o busy --output-format csv -- python3 tools/bfv_profile.py $n $B > $out/busy.log 2>&1
python3 - "$out" "$B" <<'PY'
import csv, glob, json, os, sys
out, B = sys.argv[1], int(sys.argv[2])
res = {}
for sub in ("insts", "busy"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").split("(")[0]
            if "abc::k_" not in name:
                continue
            d = res.setdefault(name, {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
summary = {}
for k, d in res.items():
    summary[k] = {c: v / (B * 5) for c, v in d.items()}
json.dump({"note": "counter sums per mul+relin (5 profiled calls of B pairs); SQ_INSTS_* count wave-instructions, SQ_*_CYCLES quad-cycles",
           "batch": B, "per_pair": summary}, open(os.path.join(out, "pmc_valu.json"), "w"), indent=1)
for k, e in sorted(summary.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    wc = e.get("SQ_WAVE_CYCLES", 1)
    print("%-52s valu %8.0f lds %7.0f | of wave cycles: valu %4.1f%% wait %4.1f%% wait_inst %4.1f%% (lds %4.1f%%) | gui %7.0f busy-valu-frac %4.2f" % (
        k[:52], e.get("SQ_INSTS_VALU", 0), e.get("SQ_INSTS_LDS", 0), 100 * e.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * e.get("SQ_WAIT_ANY", 0) / wc,
        100 * e.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * e.get("SQ_WAIT_INST_LDS", 0) / wc, e.get("GRBM_GUI_ACTIVE", 0),
        e.get("SQ_ACTIVE_INST_VALU", 0) / max(1.0, e.get("GRBM_GUI_ACTIVE", 0) / 8 * 1024 / 4)))
PY
rm -rf $out/insts $out/busy
