mkdir -p gpurun_out
timeout 200 python tools/bench_ops.py --batch 524288 > gpurun_out/bench_ops_r11.json 2>/dev/null; python -c "
import json; o=json.load(open('gpurun_out/bench_ops_r11.json')); print({k:(v.get('ops_per_s') or v.get('rows_per_s')) for k,v in o.items() if isinstance(v,dict) and k!='geometry'}, {k:v.get('bit_exact_sample') for k,v in o.items() if isinstance(v,dict) and 'bit_exact_sample' in v})"
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r11.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu_r11.log
