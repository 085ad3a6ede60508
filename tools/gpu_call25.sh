#!/bin/bash
# end-of-round validation: full GPU parity suite + smoke + one bench line
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > gpurun_out/c25_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/c25_pytest.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/c25_smoke.log 2>&1
tail -1 gpurun_out/c25_smoke.log
( timeout 400 python bench.py --no-cpu-baseline --no-prefill-roofline ) > gpurun_out/c25_bench.log 2>&1
grep '^{"metric' gpurun_out/c25_bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ttft_p50_ms'], d['roofline']['frac'], d['small_batch_ms_per_step'])"
