#!/bin/bash
# round 5, second GPU call: the driver's bench command with the row-owner kernel in the search
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r05b_bench.json 2> gpurun_out/r05b_bench.err
tail -3 gpurun_out/r05b_bench.err | cut -c1-300
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05b_bench.json").read().strip().splitlines()[-1])
print(json.dumps(d["summary"]))
for r in d["gemm_tune"]["shapes"]:
    if r["M"] in (256, 128, 64):
        print(r["M"], r["name"], r["tuned_us"], r.get("hand_written", "library")[:60], r.get("library_best_us"))
for r in d["gemm_tune"]["refined_in_graph"]:
    print(r["name"], r.get("chosen"), r.get("changed"), json.dumps(r.get("tried", r.get("ms", {})))[:600])
PY
