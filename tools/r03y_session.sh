cd $GRAFT_REPO_ROOT
o=gpurun_out/r03y; mkdir -p $o
make -C tools qos_bench batcher_bench > /dev/null 2>&1
(timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -30) > $o/pytest.log; tail -4 $o/pytest.log
for v in quad lane quad lane; do echo "== 3 callers GEC_BLAKE2_KERNEL=$v" >> $o/lat.txt; GEC_BLAKE2_KERNEL=$v timeout 60 tools/batcher_bench 3 100 2>&1 | tail -1 >> $o/lat.txt; done
for v in quad lane; do echo "== 1 caller GEC_BLAKE2_KERNEL=$v" >> $o/lat.txt; GEC_BLAKE2_KERNEL=$v timeout 60 tools/batcher_bench 1 200 2>&1 | tail -1 >> $o/lat.txt; done
for v in quad lane; do echo "== 48 callers GEC_BLAKE2_KERNEL=$v" >> $o/lat.txt; GEC_BLAKE2_KERNEL=$v timeout 60 tools/batcher_bench 48 20 2>&1 | tail -1 >> $o/lat.txt; done
cat $o/lat.txt
timeout 300 python tools/get_latency.py > $o/get_latency.txt 2>&1; tail -12 $o/get_latency.txt
