cd /root/repo
mkdir -p gpurun_out/final
bash scripts/profile_round.sh r06 "C2H:3:short C5:0:short" > gpurun_out/final/profile_round2.log 2>&1; tail -3 gpurun_out/final/profile_round2.log
cp gpurun_out/profiles_r06/r06_* profiles/ 2>/dev/null; cp gpurun_out/profiles_r06/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
python bench.py > gpurun_out/final/bench_n1.json 2> gpurun_out/final/bench_n1.err; tail -c 200 gpurun_out/final/bench_n1.json
