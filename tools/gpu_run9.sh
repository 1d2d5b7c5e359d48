mkdir -p gpurun_out
for b in 0 2 3; do
timeout 300 python bench.py --batch 262144 --steps 1 --warmup 1 --no-cpu-baseline --blocks-per-cu $b > gpurun_out/bench_r9_$b.json 2> gpurun_out/bench_r9_$b.err; echo "bpc=$b rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/bench_r9_$b.json')); print('   enc/s %.0f dec/s %.0f'%(d['value'], d['decrypt']['value']), d['config']['geometry'], d['bit_exact'])"
done
