mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pytest_gpu_r10.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu_r10.log
timeout 300 python bench.py --batch 262144 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r10.json 2> gpurun_out/bench_r10.err; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/bench_r10.json')); print('2048: enc/s %.0f dec/s %.0f'%(d['value'], d['decrypt']['value']), d['config']['geometry'], d['bit_exact'])"
timeout 300 python bench.py --key-bits 3072 --batch 131072 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r10_3072.json 2> gpurun_out/bench_r10_3072.err; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/bench_r10_3072.json')); print('3072: enc/s %.0f dec/s %.0f'%(d['value'], d['decrypt']['value']), d['config']['geometry'], d['bit_exact'])"
timeout 200 python tools/bench_ops.py --batch 524288 > gpurun_out/bench_ops_r10.json 2>/dev/null; python -c "
import json; o=json.load(open('gpurun_out/bench_ops_r10.json')); print({k:(v.get('ops_per_s') or v.get('rows_per_s')) for k,v in o.items() if isinstance(v,dict) and k!='geometry'})"
