#!/bin/bash
cd /tmp && export TMPDIR=/tmp
SEG_ONLY=1 timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/segtr -- python /root/repo/tools/dbg/seg_time.py > /tmp/segtr.log 2>&1
tail -3 /tmp/segtr.log
f=$(find /tmp/segtr -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
# the tail of one late step: from 12 kernels before the first bucket scaling (mul) to the optimiser's end
mul = [i for i, e in enumerate(ev) if "MulFunctor" in e[2] or "mul" in e[2].lower() and "elementwise" in e[2].lower()]
print("mul kernels", len(mul))
i0 = mul[-10] if len(mul) >= 10 else mul[0]
t0 = ev[i0 - 12][0]
for j in range(i0 - 12, min(len(ev), i0 + 40)):
    s, e, n = ev[j]
    print(f"{(s - t0)/1e3:9.1f} us  +{(e - s)/1e3:7.1f}  {n[:90]}")
PY
