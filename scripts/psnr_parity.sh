#!/bin/bash
# The PSNR-parity grid on ONE GPU with virtual ranks (Trainer.views_per_step = the averaged N-view step of N view-parallel ranks):
#   bash scripts/psnr_parity.sh [K]
K=${1:-6000}; OUT=gpurun_out/psnr_parity.jsonl; : > $OUT
python scripts/psnr_parity.py steps $K 1 2>/dev/null | grep '^{' >> $OUT
for N in 2 4 8; do for rule in steps images images-lrsqrt images-lrN; do
  python scripts/psnr_parity.py $rule $K $N 2>/dev/null | grep '^{' >> $OUT
done; done
python - <<PY
import json
rows=[json.loads(l) for l in open("$OUT")]
fin={}
for r in rows:
    if "psnr_train" in r: fin[(r["world"],r["rule"])]=r
for k,r in sorted(fin.items()):
    print(k, "steps",r["steps"],"images",r["images_seen"],"points",r["points"],"train",r["psnr_train"],"heldout",r["psnr_heldout"],"wall",r["wall_s"])
PY
