#!/bin/bash
# full GPU test suite + the default bench line (N = 1)
O=gpurun_out/${1:-r02full}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest.txt
timeout 600 python bench.py 2> $O/bench.err | tail -1 > $O/bench.json
python - <<PY
import json
d = json.loads(open("$O/bench.json").read())
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "roofline", d["roofline"])
print(json.dumps(d.get("e2e"), indent=1))
print("cpu", d.get("cpu_baseline"))
PY
