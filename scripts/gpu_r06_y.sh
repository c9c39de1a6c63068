#!/usr/bin/env bash
set -u
O=gpurun_out/r06y; mkdir -p $O
python -m pytest tests/test_side_streams.py tests/test_rccl_world1.py -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -h "gradient gap\|passed\|failed\|rc=" $O/pytest.log | cut -c1-300
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; wc -l $O/bench.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06y/bench.json").readline())
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
t=d["training_step"]; print({k:(round(v,2) if isinstance(v,float) else v) for k,v in t.items() if k in ("ms_per_step","ddp_ms_per_step","in_order_ms_per_step","fused_adam_ms_per_step","steps_ms")})
print({k:(round(v,2) if isinstance(v,float) else v) for k,v in t["train_precision_bf16"].items() if "ms" in k})
PY
tail -3 $O/bench.err
