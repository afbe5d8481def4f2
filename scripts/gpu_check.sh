#!/bin/bash
# full GPU check: tests (fp32 + bf16 builds are parametrised inside), then bench; logs under gpurun_out/
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread $PYTEST_EXTRA > gpurun_out/tests.log 2>&1
rc=$?; tail -5 gpurun_out/tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u scripts/gemm_bench.py > gpurun_out/gemm_bench.log 2>&1 || { tail -20 gpurun_out/gemm_bench.log; exit 1; }
cat gpurun_out/gemm_bench.log
timeout -k 10 400 python -u bench.py > gpurun_out/bench.log 2>&1 || { tail -20 gpurun_out/bench.log; exit 1; }
grep '^{' gpurun_out/bench.log | tail -1
