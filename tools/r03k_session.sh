cd $GRAFT_REPO_ROOT
o=gpurun_out/r03k; mkdir -p $o
make -C tools qos_bench > /dev/null 2>&1
for w in 6 4 3 2; do echo "== 48 callers GEC_BG_LINK_WGS=$w" >> $o/qos.txt; GEC_BG_LINK_WGS=$w timeout 60 tools/qos_bench 48 2.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done
for pc in 30 60; do echo "== 48 callers GEC_BG_PACE_PCT=$pc" >> $o/qos.txt; GEC_BG_PACE_PCT=$pc timeout 60 tools/qos_bench 48 2.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done
for w in 4 2; do echo "== 3 callers GEC_BG_LINK_WGS=$w" >> $o/qos.txt; GEC_BG_LINK_WGS=$w timeout 60 tools/qos_bench 3 2.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done
echo "== 3 callers GEC_BG_PACE_PCT=30" >> $o/qos.txt; GEC_BG_PACE_PCT=30 timeout 60 tools/qos_bench 3 2.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt
grep -E "^==|with the class|scrub alone|puts alone|background class  " $o/qos.txt | cut -c1-200
