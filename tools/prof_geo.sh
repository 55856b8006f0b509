#!/bin/bash
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_geo
rm -rf $out; mkdir -p $out
P2PB_EXPERIMENT="nn_cells=1" timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $out -o geo -- python $GRAFT_REPO_ROOT/tools/exp_geo.py > $out/log 2>&1
python - <<'PY'
import sqlite3, glob, os
db = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_geo/*.db")[0]
cur = sqlite3.connect(db).cursor()
for r in cur.execute("select substr(name,1,50), grid_x, count(*), avg(end-start)/1e3 from kernels where name like '%nn%' or name like '%ball%' group by 1,2 order by 4 desc"):
    print(r)
PY
