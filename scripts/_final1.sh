cd /root/repo
mkdir -p gpurun_out/final
timeout 1700 python -m pytest tests -q -m gpu > gpurun_out/final/gpu_tests.txt 2>&1; tail -4 gpurun_out/final/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.txt 2>&1; tail -2 gpurun_out/final/smoke.txt
