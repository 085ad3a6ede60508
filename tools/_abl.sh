MSGL_DISABLE_M256=1 timeout 300 python bench.py --no-cpu-baseline --no-prefill-roofline --small-batches > gpurun_out/bench_nom256.json 2> gpurun_out/bench_nom256.err; python -c "
import json; d=json.load(open('gpurun_out/bench_nom256.json')); print('no m256:', d['ms_per_step'], d['value']); print([(s['name'], s['tuned_us'], s.get('hand_written','lib')[:40]) for s in d['gemm_tune']['shapes']])"
timeout 300 python bench.py --no-cpu-baseline --no-prefill-roofline --small-batches > gpurun_out/bench_m256.json 2> gpurun_out/bench_m256.err; python -c "
import json; d=json.load(open('gpurun_out/bench_m256.json')); print('m256:', d['ms_per_step'], d['value']); print([(s['name'], s['tuned_us'], s.get('hand_written','lib')[:40]) for s in d['gemm_tune']['shapes']])"
tail -3 gpurun_out/bench_m256.err
