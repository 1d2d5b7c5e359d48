mkdir -p gpurun_out
timeout 300 python bench.py --batch 262144 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r8.json 2> gpurun_out/bench_r8.err; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/bench_r8.json')); print('   enc/s %.0f dec/s %.0f'%(d['value'], d['decrypt']['value']), d['config']['geometry'], d['bit_exact'])"
timeout 300 python bench.py --key-bits 3072 --batch 131072 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r8_3072.json 2> gpurun_out/bench_r8_3072.err; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/bench_r8_3072.json')); print('3072: enc/s %.0f dec/s %.0f'%(d['value'], d['decrypt']['value']), d['config']['geometry'], d['bit_exact'])"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or random or large" > gpurun_out/pytest_gpu_r8.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_r8.log
