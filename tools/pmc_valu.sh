#!/bin/bash
# VALU instruction counts and busy cycles per kernel of the headline call (one rocprofv3 counter pass)
# usage (on the GPU box): bash tools/pmc_valu.sh <tag>
set -e
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcv_$tag
mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $out/insts -o insts --output-format csv -- python3 bench.py --batch 1024 --steps 2 --warmup 1 --no-cpu --no-60bit --no-bfv > $out/insts.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $out/busy -o busy --output-format csv -- python3 bench.py --batch 1024 --steps 2 --warmup 1 --no-cpu --no-60bit --no-bfv > $out/busy.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for sub in ("insts", "busy"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").split("(")[0]
            if "abc::k_fused" not in name and "abc::k_split" not in name:
                continue
            d = res.setdefault(name, {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            d.setdefault("_dispatches_" + sub, set()).add(r["Dispatch_Id"])
batch, calls = 1024, 3
summary = {}
for k, d in res.items():
    e = {c: v / (batch * calls) for c, v in d.items() if not c.startswith("_")}
    summary[k] = e
json.dump({"note": "counter sums per multiply (batch 1024, 3 profiled calls); SQ_INSTS_* count wave-instructions",
           "per_multiply": summary}, open(os.path.join(out, "pmc_valu.json"), "w"), indent=1)
for k, e in summary.items():
    print(k, {c: round(v, 1) for c, v in e.items()})
PY
