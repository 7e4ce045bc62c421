cd $GRAFT_REPO_ROOT
o=gpurun_out/r03j; mkdir -p $o
make -C tools qos_bench > /dev/null 2>&1
export GEC_BG_PACE_PCT=0
for mb in 8 16 32; do for i in 1 2; do echo "== 48 callers GEC_BG_CHUNK_MB=$mb $i" >> $o/qos.txt; GEC_BG_CHUNK_MB=$mb timeout 60 tools/qos_bench 48 2.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done; done
for mb in 8 16; do for i in 1 2; do echo "== 3 callers GEC_BG_CHUNK_MB=$mb $i" >> $o/qos.txt; GEC_BG_CHUNK_MB=$mb timeout 60 tools/qos_bench 3 2.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done; done
echo "== 48 callers GEC_BG_CHUNK_MB=16 GEC_BG_YIELD_US=0" >> $o/qos.txt; GEC_BG_YIELD_US=0 GEC_BG_CHUNK_MB=16 timeout 60 tools/qos_bench 48 2.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt
grep -E "^==|with the class|scrub alone|puts alone|background class  " $o/qos.txt | cut -c1-200
