set -x
cd /root/repo
mkdir -p gpurun_out/s7
python scripts/lib_identity.py fl0 - scan > gpurun_out/s7/identity_flush.txt 2>&1; tail -2 gpurun_out/s7/identity_flush.txt
python scripts/ab_libs.py "garden,C4,trained" 3 base=fl0 bits=- > gpurun_out/s7/ab_flush.jsonl 2> gpurun_out/s7/ab_flush.err; cat gpurun_out/s7/ab_flush.jsonl | cut -c1-250
