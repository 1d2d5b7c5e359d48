"""VGPRs / spilled VGPRs / scratch bytes / scratch instructions of every kernel of one csrc translation unit, read from the
gfx950 assembly hipcc emits (no GPU needed):  python tools/kernel_resources.py kernels_s4b [more units...]
With --loops: per kernel, the scratch_* instructions that sit inside a loop body (between a label and a backward branch to it)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "python-paillier_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-pragma-unroll-threshold=1000000", "--offload-device-only", "-S"]


def assembly(unit, extra=()):
    src = unit if os.path.isabs(unit) else os.path.join(CSRC, unit + ".hip")
    unit = os.path.splitext(os.path.basename(unit))[0]
    out = os.path.join(tempfile.gettempdir(), "phe_asm_%s.s" % unit)
    if os.environ.get("PHE_ASM_REUSE") and os.path.exists(out):
        return open(out).read()
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + list(extra) + ["-o", out, "-I", CSRC, src],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def kernels(text):
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S):
        body = m.group(2)
        if "vgpr_count" not in body:
            continue
        get = lambda k: int((re.search(r"\." + k + r":\s+(\d+)", body) or [0, 0])[1])
        meta[m.group(1)] = {"vgpr": get("vgpr_count"), "spilled": get("vgpr_spill_count"), "scratch_bytes": get("private_segment_fixed_size")}
    for name in meta:
        m = re.search(r"^%s:[^\n]*\n(.*?)^\s*s_endpgm" % re.escape(name), text, re.S | re.M)
        code = m.group(1) if m else ""
        lines = code.split("\n")
        meta[name]["instructions"] = sum(1 for l in lines if re.match(r"\s+[a-z]", l) and not l.strip().startswith((".", ";")))
        meta[name]["mads"] = sum(1 for l in lines if "v_mad_u64_u32" in l)
        meta[name]["scratch_insts"] = sum(1 for l in lines if re.match(r"\s+scratch_", l))
        # loops: label .LBBx_y ... s_cbranch* .LBBx_y (backward)
        pos = {}
        loops = []
        for i, l in enumerate(lines):
            lab = re.match(r"^(\.LBB\d+_\d+):", l)
            if lab:
                pos[lab.group(1)] = i
            br = re.match(r"\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
            if br and br.group(1) in pos:
                lo = pos[br.group(1)]
                body = lines[lo:i]
                loops.append({"label": br.group(1), "insts": sum(1 for b in body if re.match(r"\s+[a-z]", b)),
                              "mads": sum(1 for b in body if "v_mad_u64_u32" in b),
                              "scratch": sum(1 for b in body if re.match(r"\s+scratch_", b))})
        meta[name]["loops"] = loops
    return meta


def short(name):
    try:
        out = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        out = name
    return re.sub(r"\(.*", "", out.replace("void phe::", ""))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    show_loops = "--loops" in sys.argv
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    for unit in args:
        for name, k in kernels(assembly(unit, defs)).items():
            print("%-44s vgpr %3d  spilled %3d  scratch %5d B  scratch-insts %4d  insts %6d  mads %6d" % (
                short(name), k["vgpr"], k["spilled"], k["scratch_bytes"], k["scratch_insts"], k["instructions"], k["mads"]))
            if show_loops:
                for lp in k["loops"]:
                    if lp["insts"] >= 200:
                        print("      loop %-12s insts %6d  mads %6d  scratch-insts %4d" % (lp["label"], lp["insts"], lp["mads"], lp["scratch"]))
