cd $GRAFT_REPO_ROOT
o=gpurun_out/r03w; mkdir -p $o
make -C tools qos_bench > /dev/null 2>&1
(timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -30) > $o/pytest.log; tail -4 $o/pytest.log
for v in 1 0 1 0 1 0 1; do echo "== callers=3 GEC_PLACE_STREAMS=$v" >> $o/qos.txt; GEC_PLACE_STREAMS=$v timeout 60 tools/qos_bench 3 1.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done
grep -E "^==|with the class|stream placement" $o/qos.txt
for v in 1 0 1 0 1; do
  echo "== GEC_PLACE_STREAMS=$v" >> $o/get.txt
  GEC_PLACE_STREAMS=$v GBM_TRACE=1 timeout 300 python tools/host_path_bench.py 512 2>&1 | grep -E "gbm\] get:" | tail -2 >> $o/get.txt
done
cat $o/get.txt
