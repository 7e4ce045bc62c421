cd $GRAFT_REPO_ROOT
o=gpurun_out/r03i; mkdir -p $o
make -C tools qos_bench > /dev/null 2>&1
for i in 1 2 3 4; do echo "== 3 callers $i" >> $o/qos.txt; timeout 60 tools/qos_bench 3 1.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done
for i in 1 2 3 4; do echo "== 48 callers $i" >> $o/qos.txt; timeout 60 tools/qos_bench 48 1.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done
for i in 1 2; do echo "== 48 callers GEC_BG_PACE_PCT=0 $i" >> $o/qos.txt; GEC_BG_PACE_PCT=0 timeout 60 tools/qos_bench 48 1.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done
for i in 1 2; do echo "== 48 callers GEC_BG_PACE_PCT=100 $i" >> $o/qos.txt; GEC_BG_PACE_PCT=100 timeout 60 tools/qos_bench 48 1.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done
grep -E "^==|with the class" $o/qos.txt
(timeout 600 python -m pytest tests/test_gpu_qos.py tests/test_block_native.py -m gpu -q 2>&1 | tail -5) > $o/pytest.log; cat $o/pytest.log
