#!/bin/bash
# round 5: the wave-per-point linearisation of the device-resident optimiser — parity tests, the example's alignment, the example
set -e
python -m pytest tests/test_gpu_optimize.py -x -q > gpurun_out/r05_m_opt_tests.log 2>&1 || { tail -40 gpurun_out/r05_m_opt_tests.log; exit 1; }
tail -3 gpurun_out/r05_m_opt_tests.log
python scratch/opt_example.py > gpurun_out/r05_m_opt_example.txt 2>&1 || { tail -30 gpurun_out/r05_m_opt_example.txt; exit 1; }
cat gpurun_out/r05_m_opt_example.txt
make -C tests/cpp example_registration > /dev/null 2>&1 || true
tests/cpp/example_registration tests/golden/source.ply tests/golden/target.ply 100 10 > gpurun_out/r05_m_example.txt 2>&1
tail -14 gpurun_out/r05_m_example.txt
