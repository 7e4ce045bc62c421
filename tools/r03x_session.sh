cd $GRAFT_REPO_ROOT
o=gpurun_out/r03x; mkdir -p $o
make -C tools qos_bench > /dev/null 2>&1
(timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -30) > $o/pytest.log; tail -4 $o/pytest.log
run() { echo "== callers=$1 $2" >> $o/qos.txt; env $2 timeout 60 tools/qos_bench $1 2.0 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; }
run 48 ""; run 48 "GEC_BG_LINK_WAIT_US=0"; run 48 ""; run 48 "GEC_BG_LINK_WAIT_US=0"; run 48 "GEC_BG_LINK_WAIT_US=500"
run 3 ""; run 3 "GEC_BG_LINK_WAIT_US=0"; run 3 ""
grep -E "^==|with the class|puts alone|background class  " $o/qos.txt | cut -c1-200
