mkdir -p gpurun_out
timeout 200 python tools/decode_ab.py --shape 14b,14b_tp4,70b_tp8,0.6b --impls 0,73,74 --out gpurun_out/c7_decode_ab.json 2>&1 | grep impl | cut -c1-120
timeout 400 python tools/step_ab.py --rounds 3 --knobs decode:73 decode:74 --out gpurun_out/c7_step_ab.json 2>&1 | tail -2 | cut -c1-500
timeout 100 python tools/decode_trace.py --out gpurun_out/c7_decode_trace_default.json > /dev/null 2>&1
python -c "
import json
d=json.load(open('gpurun_out/c7_decode_trace_default.json')); print('default lifetimes', [round(x) for x in d['lifetime_by_wave_of_the_workgroup']], d['event_us_kernel_plus_merge'])"
