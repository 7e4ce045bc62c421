#!/bin/bash
# The two comparisons tests/test_gpu_qos.py asserts, N times each: the distribution of "p99 with the class / solo".
n=${1:-6}
make -C tools qos_bench > /dev/null || exit 1
for i in $(seq 1 $n); do tools/qos_bench 3 1.5 256 | grep -E "^with the class|^without the class" | tr '\n' ' '; echo; done
echo "--- degraded gets beside a resync"
for i in $(seq 1 $n); do tools/qos_bench 3 1.5 256 0 4 0 4 resync | grep -E "^with the class|^without the class" | tr '\n' ' '; echo; done
