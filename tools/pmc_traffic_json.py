#!/usr/bin/env python3
"""HBM traffic and VALU instruction counts per element, from rocprofv3 --pmc passes over `bench.py --only encrypt|decrypt`.

    python tools/pmc_traffic_json.py --batch B --leg encrypt DIR_VALU DIR_FETCH DIR_WRITE [--leg decrypt DIR DIR DIR] > hbm_traffic_rNN.json

`bench.py --only LEG --batch B --steps K --warmup W` launches the leg's kernels on exactly B rows per dispatch (one encrypt
launch precedes the timed ones in either leg), so rows = dispatches x B with no hand bookkeeping.  FETCH_SIZE is doubled for
gfx950 (MI355X_MICROARCH.md, HBM section: the counter counts 128-byte requests as 64), both counters are KB.  The output
carries the hash of the device sources (tools/csrc_hash.py): bench.py marks a file made from other sources as stale.
rocprofv3 cannot run inside bench.py, hence a committed file; tools/gpu_pmc_traffic.sh makes it on the GPU box."""
import argparse
import glob
import json
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_hash import csrc_hash  # noqa: E402

MODES = {0: "encrypt", 1: "obfuscate", 2: "half_decrypt"}
# legs named ops<key bits> (tools/gpu_pmc_traffic.sh: `tools/bench_sweep.py --ops add,mul,pair_add` at ONE batch size, so that rows =
# dispatches x batch holds for them too): the kernel behind each entry of bench.py's `ops` object (configs[2])
OP_OF = [(r"^k_mulmod_tile<", "raw_add"), (r"^k_mulmod_table<", "raw_add_table_in_lds"), (r"^k_mulmod_staged<", "raw_add_two_montgomery_products"),
         (r"^k_modexp_var_split<\d+,\d+,false>", "raw_mul_float56"), (r"^k_modexp_var_split<\d+,\d+,true>", "raw_mul_float56_resident_pair_form"),
         (r"^k_pair_mul<", "raw_add_resident_pair_form"), (r"^k_to_pair<", "to_pair_form"), (r"^k_from_pair<", "from_pair_form")]


def counters(directory):
    out = {}
    for db in sorted(glob.glob(os.path.join(directory, "**", "*.db"), recursive=True)):
        con = sqlite3.connect(db)
        try:
            rows = con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                               "group by kernel_name, counter_name").fetchall()
        except sqlite3.OperationalError:
            rows = []
        for name, counter, value, n in rows:
            rec = out.setdefault(name, {})
            rec[counter] = rec.get(counter, 0.0) + float(value)
            rec["dispatches:" + counter] = rec.get("dispatches:" + counter, 0) + int(n)
        con.close()
    return out


def key_of(name):
    m = re.search(r"k_modexp_split<(\d+), (\d+), (\d+), (true|false)>", name)
    if m:
        return "k_modexp_split<%s,%s,%s>" % (m.group(1), m.group(2), MODES.get(int(m.group(3)), m.group(3)))
    m = re.search(r"(k_\w+)<([^>]*)>", name)
    if m:
        return "%s<%s>" % (m.group(1), m.group(2).replace(" ", ""))
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, required=True)
    ap.add_argument("--leg", action="append", nargs=4, metavar=("NAME", "DIR_VALU", "DIR_FETCH", "DIR_WRITE"), required=True)
    ap.add_argument("--source", default="")
    args = ap.parse_args()
    out = {"source": args.source or "rocprofv3 --pmc, three passes per leg (VALU group / FETCH_SIZE / WRITE_SIZE) over bench.py --only <leg>",
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); bytes; Infinity-Cache hits are "
                   "included in FETCH_SIZE.  rows = dispatches x --batch (bench.py --only launches the leg's kernels on the whole batch)",
           "csrc_sha256": csrc_hash(), "batch_per_dispatch": args.batch}
    for leg, d_valu, d_fetch, d_write in args.leg:
        valu, fetch, write = counters(d_valu), counters(d_fetch), counters(d_write)
        for name in sorted(set(valu) | set(fetch) | set(write)):
            key = key_of(name)
            if key is None or "phe" not in name:
                continue
            if leg == "decrypt" and "half_decrypt" not in key and "tail" not in key and "halves" not in key:
                continue                                       # (the one encrypt launch that makes the ciphertexts)
            if leg == "encrypt" and ("half_decrypt" in key or "tail" in key):
                continue
            ops_leg = re.match(r"ops(\d+)$", leg)
            if ops_leg and (key.startswith("k_modexp_split<") or "tail" in key):
                continue                                       # (the launches that make the operands)
            n_disp = max(valu.get(name, {}).get("dispatches:SQ_INSTS_VALU", 0), fetch.get(name, {}).get("dispatches:FETCH_SIZE", 0),
                         write.get(name, {}).get("dispatches:WRITE_SIZE", 0))
            rows = n_disp * args.batch
            if not rows:
                continue
            rec = {"leg": leg, "launches": n_disp, "batch": rows}
            if "FETCH_SIZE" in fetch.get(name, {}):
                rec["fetch_bytes"] = fetch[name]["FETCH_SIZE"] * 2 * 1024
            if "WRITE_SIZE" in write.get(name, {}):
                rec["write_bytes"] = write[name]["WRITE_SIZE"] * 1024
            if "SQ_INSTS_VALU" in valu.get(name, {}):
                rec["sq_insts_valu"] = valu[name]["SQ_INSTS_VALU"]
                rec["valu_wave_instructions_per_row"] = rec["sq_insts_valu"] / rows
            if "fetch_bytes" in rec and "write_bytes" in rec:
                rec["bytes_per_row"] = (rec["fetch_bytes"] + rec["write_bytes"]) / rows
            if "GRBM_GUI_ACTIVE" in valu.get(name, {}) and "SQ_BUSY_CYCLES" in valu.get(name, {}):
                rec["grbm_gui_active"] = valu[name]["GRBM_GUI_ACTIVE"]
                rec["sq_busy_cycles"] = valu[name]["SQ_BUSY_CYCLES"]
            if ops_leg:
                rec["kernel"] = key
                op = next((o for pat, o in OP_OF if re.match(pat, key)), None)
                if op:
                    out.setdefault("ops", {}).setdefault(ops_leg.group(1), {})[op] = rec
                continue
            out[key] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
