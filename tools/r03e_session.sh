mkdir -p gpurun_out/r03e
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -80) > gpurun_out/r03e/pytest.log
tail -5 gpurun_out/r03e/pytest.log
for i in 1 2 3; do timeout 60 tools/qos_bench 3 2 512 2>&1 | tail -4 >> gpurun_out/r03e/qos_repeat.txt; done
timeout 60 tools/qos_bench 48 2 512 2>&1 | tail -4 >> gpurun_out/r03e/qos_repeat.txt
cat gpurun_out/r03e/qos_repeat.txt
make -C tests/c put_get_callers > /dev/null 2>&1; timeout 120 tests/c/put_get_callers 16 12 1048576 4 > gpurun_out/r03e/put_get_callers.txt 2>&1; cat gpurun_out/r03e/put_get_callers.txt
GARAGE_DRYRUN_ONE_GPU=1 timeout 300 python bench.py --gpus 2 --mode threads --steps 50 --warmup 10 --batch 256 --no-cpu-baseline 2>gpurun_out/r03e/dry_threads.err | tail -1 > gpurun_out/r03e/dryrun_2_threads_line.json
GARAGE_DRYRUN_ONE_GPU=1 timeout 300 python bench.py --gpus 2 --steps 50 --warmup 10 --batch 256 --no-cpu-baseline 2>gpurun_out/r03e/dry_procs.err | tail -1 > gpurun_out/r03e/dryrun_2_procs_line.json
timeout 400 python bench.py 2>gpurun_out/r03e/bench.err | tail -1 > gpurun_out/r03e/bench.json
cut -c1-300 gpurun_out/r03e/bench.json
