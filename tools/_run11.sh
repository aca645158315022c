set -e
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1 || { tail -40 gpurun_out/t_all.log; exit 1; }
tail -2 gpurun_out/t_all.log
timeout -k 10 900 python tools/fuzz_parity.py 700 31337 > gpurun_out/fuzz_r2f.log 2>&1 || { tail -20 gpurun_out/fuzz_r2f.log | cut -c1-600; exit 1; }
tail -1 gpurun_out/fuzz_r2f.log | cut -c1-900
