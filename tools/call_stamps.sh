R=$GRAFT_REPO_ROOT; cd $R; o=gpurun_out/stamps; mkdir -p $o
python tools/exp_stamps.py > $o/stamps2.txt 2> $o/err.txt
CHAINS=1 python tools/exp_stamps.py > $o/stamps1.txt 2>> $o/err.txt
for c in 2 1; do for k in 0 4 0 4; do MODE=noops NOOPS=$k CHAINS=$c python tools/exp_stamps.py 2>>$o/err.txt | head -1 >> $o/noops.txt; done; done
cat $o/noops.txt; head -3 $o/stamps2.txt; tail -3 $o/err.txt
