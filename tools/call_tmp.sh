cd $GRAFT_REPO_ROOT; o=gpurun_out/c8; mkdir -p $o
tools/ab_libs.sh 3 tools/exp/lib_pwpkfma.so "" | tee $o/ab_pkfma.txt
