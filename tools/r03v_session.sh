cd $GRAFT_REPO_ROOT
o=gpurun_out/r03v; mkdir -p $o
make -C tools qos_bench > /dev/null 2>&1
for v in "GEC_RESIDENT_GRID=1" "GEC_RESIDENT_GRID=0" "GEC_RESIDENT_GRID=1" "GEC_RESIDENT_GRID=0"; do
  echo "== $v (host_path_bench main)" >> $o/get.txt
  env $v GBM_TRACE=1 timeout 300 python tools/host_path_bench.py 512 2>&1 | grep -E "gbm\] get|rpc_get_blocks_4|rpc_put_blocks_GiBps|rpc_get_blocks_GiBps|encode_hash_pinned|verify_pinned" | tail -9 >> $o/get.txt
done
cat $o/get.txt
for i in 1 2 3 4; do echo "== callers=3 $i" >> $o/qos.txt; timeout 60 tools/qos_bench 3 2.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done
for i in 1 2; do echo "== callers=48 $i" >> $o/qos.txt; timeout 60 tools/qos_bench 48 2.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done
grep -E "^==|with the class" $o/qos.txt
