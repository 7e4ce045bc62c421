cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
lscpu | grep -i "numa\|socket\|model name" > gpurun_out/bm_numa.txt
numactl --hardware >> gpurun_out/bm_numa.txt 2>&1
rocm-smi --showtoponuma >> gpurun_out/bm_numa.txt 2>&1
cat /sys/class/drm/card*/device/numa_node >> gpurun_out/bm_numa.txt 2>&1
for i in 1 2; do
GBM_TRACE=1 timeout 600 python -c "
import json,sys
sys.path.insert(0,'tools')
import host_path_bench as h
print(json.dumps(h.block_manager_rates(512)))
" > gpurun_out/bm_trace_$i.json 2> gpurun_out/bm_trace_$i.err
done
