mkdir -p gpurun_out/r03d
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -150) > gpurun_out/r03d/pytest.log
tail -6 gpurun_out/r03d/pytest.log
for t in 1 8 16 32 64 128; do GEC_CPU_THREADS=$t tools/cpu_backend_bench 512 >> gpurun_out/r03d/cpu_backend.txt 2>&1; done
cat gpurun_out/r03d/cpu_backend.txt
timeout 120 tools/batcher_bench 48 20 > gpurun_out/r03d/batcher_48.txt 2>&1; timeout 120 tools/batcher_bench 3 100 > gpurun_out/r03d/batcher_3.txt 2>&1; timeout 120 tools/batcher_bench 192 10 > gpurun_out/r03d/batcher_192.txt 2>&1
cat gpurun_out/r03d/batcher_*.txt
timeout 900 bash tools/profile_round.sh > gpurun_out/r03d/profile_round.log 2>&1
tail -3 gpurun_out/bench_final.json | cut -c1-600
