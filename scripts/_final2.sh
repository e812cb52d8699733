cd /root/repo
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_n1.json 2> gpurun_out/final/bench_n1.err; tail -c 300 gpurun_out/final/bench_n1.json
bash scripts/profile_round.sh r06 "garden:3:full C2:0:short C4:3:short trained:3:short" > gpurun_out/final/profile_round.log 2>&1; tail -3 gpurun_out/final/profile_round.log
timeout 1700 python -m pytest tests -q -m gpu > gpurun_out/final/gpu_tests.txt 2>&1; grep -n "passed\|failed" gpurun_out/final/gpu_tests.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.txt 2>&1; tail -1 gpurun_out/final/smoke.txt
