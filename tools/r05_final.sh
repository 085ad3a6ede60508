#!/bin/bash
# final check of the round on the committed tree: the GPU parity suite and the driver's bench command
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > gpurun_out/r05_final_pytest.log 2>&1
tail -6 gpurun_out/r05_final_pytest.log | cut -c1-200
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r05_final_bench.json 2> gpurun_out/r05_final_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_final_bench.json").read().strip().splitlines()[-1])
print(json.dumps(d["summary"]))
print(d["roofline"]["traffic_source"], d["reference_driven"].get("stale"), d["cpu_baseline"]["sample"][-200:])
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
