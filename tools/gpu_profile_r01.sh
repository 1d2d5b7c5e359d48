mkdir -p gpurun_out; R=$PWD
cat /sys/fs/cgroup/cpu.max > gpurun_out/host_info.txt 2>&1; nproc >> gpurun_out/host_info.txt; python -c "import os; print(len(os.sched_getaffinity(0)))" >> gpurun_out/host_info.txt; lscpu | head -20 >> gpurun_out/host_info.txt
timeout 200 ./python-paillier_amd/lib/phe_microbench > gpurun_out/microbench2.json 2> gpurun_out/microbench2.err; echo "microbench rc=$?"
timeout 600 python bench.py > gpurun_out/bench_1M.json 2> gpurun_out/bench_1M.err; echo "bench rc=$?"; cat gpurun_out/bench_1M.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- python $R/bench.py --batch 262144 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_kt.log 2>&1; echo "rocprof kt rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS -d $R/gpurun_out/prof_pmc1 -- python $R/bench.py --batch 65536 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_pmc1.log 2>&1; echo "rocprof pmc1 rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_pmc2 -- python $R/bench.py --batch 65536 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_pmc2.log 2>&1; echo "rocprof pmc2 rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_pmc3 -- python $R/bench.py --batch 65536 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_pmc3.log 2>&1; echo "rocprof pmc3 rc=$?"
cd $R; find gpurun_out -name "*.csv" | head -30; du -sh gpurun_out
