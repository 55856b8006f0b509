cd $GRAFT_REPO_ROOT; o=gpurun_out/c10; mkdir -p $o
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $o/tests.txt
cat $o/tests.txt
