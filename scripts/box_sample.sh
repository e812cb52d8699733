python - <<EOF
import sys, json
sys.path.insert(0, "2d-gaussian-splatting_amd"); sys.path.insert(0, ".")
import torch
from helpers_bench import box_probe
d = box_probe(torch.device("cuda:0"))
d.pop("note", None); d.pop("reference_box", None); d.pop("step_split_assumed", None)
print("PROBE " + json.dumps(d))
EOF
python bench.py --quick --steps 30 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('STEP', d['ms_per_step'], json.dumps(d['roofline']['all_kernels_ms']))
"
