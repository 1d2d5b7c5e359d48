#!/bin/bash
# What shader clock does the chip hold while its VALU is saturated?  Runs a load in the background for a few seconds — the
# probe's one-instruction burns (v_mad_u64_u32, v_fma_f32) and the bench's own encrypt kernel — and samples rocm-smi's current
# sclk and socket power beside it.  (tools, not part of the library; output: profiles/rNN_clock_under_load.txt)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
SMI=/opt/rocm/bin/rocm-smi
sample() { $SMI --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket" | sed 's/^GPU\[0\]\t*: //' | tr '\n' ' '; echo; }
echo "== idle"; sample
for kind in mad fma pkfma; do
    python-paillier_amd/lib/phe_latency_probe --burn $kind 6 > /tmp/burn_$kind.json &
    BG=$!
    sleep 1.5
    for i in 1 2 3 4; do echo -n "== burn $kind, sample $i: "; sample; sleep 1; done
    wait $BG; cat /tmp/burn_$kind.json
done
python bench.py --steps 6 --warmup 1 --no-ops --no-config4 --no-cpu-baseline --only encrypt > /tmp/bench_enc.json 2>/dev/null &
BG=$!
for i in $(seq 1 60); do
    line=$(sample)
    w=$(echo "$line" | sed -n 's/.*(W): \([0-9]*\).*/\1/p')
    [ -n "$w" ] && [ "$w" -gt 500 ] && echo "== bench.py encrypt 2^20 (k_modexp_split<4,18,encrypt,scaled>), second $i: $line"
    kill -0 $BG 2>/dev/null || break
    sleep 1
done
wait $BG; cut -c1-260 /tmp/bench_enc.json
