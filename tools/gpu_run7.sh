mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pytest_gpu_r7.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu_r7.log
for g in 0 4 8; do
  if [ "$g" = "0" ]; then unset PHE_HIP_GROUP; else export PHE_HIP_GROUP=$g; fi
  timeout 300 python bench.py --batch 262144 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_grp$g.json 2> gpurun_out/bench_grp$g.err; echo "group=$g rc=$?"
  python -c "
import json; d=json.load(open('gpurun_out/bench_grp$g.json')); print('   enc/s %.0f dec/s %.0f'%(d['value'], d['decrypt']['value']), d['config']['geometry'], d['bit_exact'])"
done
unset PHE_HIP_GROUP
timeout 300 python bench.py --key-bits 1024 --batch 524288 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_1024.json 2> gpurun_out/bench_1024.err; python -c "
import json; d=json.load(open('gpurun_out/bench_1024.json')); print('1024-bit: enc/s %.0f dec/s %.0f'%(d['value'], d['decrypt']['value']), d['config']['geometry'], d['bit_exact'], d['roofline']['frac'], d['roofline']['decrypt']['frac'])"
