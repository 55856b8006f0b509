#!/bin/bash
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_calib
rm -rf $out; mkdir -p $out
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/p1 -o pmc -- python $GRAFT_REPO_ROOT/tools/exp/calib_fetch.py > $out/p1.log 2>&1
python - <<'PY'
import csv, glob, os, collections
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_calib/p1/*counter_collection.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and r["Kernel_Name"].startswith("read"):
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, "FETCH_SIZE per launch:", sum(v) / len(v), "KiB for", (1 << 30) / 1024, "KiB read -> ratio", (1 << 30) / 1024 / (sum(v) / len(v)))
PY
