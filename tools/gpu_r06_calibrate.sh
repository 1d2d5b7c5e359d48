#!/bin/bash
# TAG=<name> bash tools/gpu_r06_calibrate.sh (through gpurun): the measured ladder re-made on the final tree WITH points below 128 rows
# (16, 32, 64: ADVICE round 5 — below the smallest measured size every rung's cost was a constant and the tie-break alone picked)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
O=gpurun_out/${TAG:-r06c}; mkdir -p $O
timeout 900 python tools/calibrate_ladder.py --min 4 > $O/ladder_gfx950.txt 2> $O/calibrate.err; echo "calibrate rc=$?"; head -3 $O/ladder_gfx950.txt; wc -l $O/ladder_gfx950.txt
timeout 400 python tools/calibrate_ladder.py --min 4 --check $O/ladder_gfx950.txt > $O/ladder_check.json 2>/dev/null; echo "check rc=$?"; tail -c 800 $O/ladder_check.json
