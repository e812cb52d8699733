cd /root/repo
mkdir -p gpurun_out/final
python scripts/lib_identity.py pairfinal - scan auto > gpurun_out/final/identity_nopair.txt 2>&1; tail -1 gpurun_out/final/identity_nopair.txt | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "scan or composite or dtu or trained or config_sizes" > gpurun_out/final/pytest_nopair.txt 2>&1; tail -2 gpurun_out/final/pytest_nopair.txt
python scripts/ab_libs.py "garden,C4,trained" 3 pair=pairfinal step=- > gpurun_out/final/ab_nopair.jsonl 2> gpurun_out/final/ab_nopair.err; cut -c1-230 gpurun_out/final/ab_nopair.jsonl
