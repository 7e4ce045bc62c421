cd $GRAFT_REPO_ROOT
o=gpurun_out/r03m; mkdir -p $o
make -C tools qos_bench > /dev/null 2>&1
for c in 8 6 5 4; do echo "== 48 callers GEC_BG_LINK_CUS=$c" >> $o/qos.txt; GEC_BG_LINK_CUS=$c timeout 60 tools/qos_bench 48 2.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done
echo "== 48 callers GEC_UPLOAD_CUS=24" >> $o/qos.txt; GEC_UPLOAD_CUS=24 GEC_BG_LINK_CUS=8 timeout 60 tools/qos_bench 48 2.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt
for c in 6 4; do echo "== 3 callers GEC_BG_LINK_CUS=$c" >> $o/qos.txt; GEC_BG_LINK_CUS=$c timeout 60 tools/qos_bench 3 2.5 512 2>&1 | grep -v "^CU masks" >> $o/qos.txt; done
grep -E "^==|with the class|scrub alone|puts alone|background class  " $o/qos.txt | cut -c1-200
