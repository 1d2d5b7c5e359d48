mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pytest_gpu_r12.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu_r12.log
timeout 300 python bench.py --batch 262144 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r12.json 2> gpurun_out/bench_r12.err; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/bench_r12.json')); print('2048: enc/s %.0f dec/s %.0f'%(d['value'], d['decrypt']['value']), d['config']['geometry'], d['bit_exact'])"
PHE_HIP_GROUP=4 timeout 300 python bench.py --batch 262144 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r12_g4.json 2> gpurun_out/bench_r12_g4.err; python -c "
import json; d=json.load(open('gpurun_out/bench_r12_g4.json')); print('2048 (G>=4): enc/s %.0f dec/s %.0f'%(d['value'], d['decrypt']['value']), d['config']['geometry'], d['bit_exact'])"
timeout 300 python bench.py --key-bits 3072 --batch 131072 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r12_3072.json 2> gpurun_out/bench_r12_3072.err; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/bench_r12_3072.json')); print('3072: enc/s %.0f dec/s %.0f'%(d['value'], d['decrypt']['value']), d['config']['geometry'], d['bit_exact'])"
timeout 300 python bench.py --key-bits 1024 --batch 524288 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r12_1024.json 2> gpurun_out/bench_r12_1024.err; python -c "
import json; d=json.load(open('gpurun_out/bench_r12_1024.json')); print('1024: enc/s %.0f dec/s %.0f'%(d['value'], d['decrypt']['value']), d['config']['geometry'], d['bit_exact'])"
