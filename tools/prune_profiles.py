#!/usr/bin/env python3
"""Housekeeping: keep under profiles/ what the design of record, the tests, the tools and bench.py name —
plus the newest file of every kind bench.py globs for and everything of the current round — and list the rest in
profiles/REMOVED.md with the commit that still holds them (`git show <commit>:profiles/<name>`).

    python tools/prune_profiles.py            # dry run: what would go
    python tools/prune_profiles.py --apply    # git rm + write profiles/REMOVED.md"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = "r06"


def main():
    os.chdir(ROOT)
    files = sorted(f for f in os.listdir("profiles") if os.path.isfile(os.path.join("profiles", f)))
    tracked = subprocess.check_output(["git", "ls-files"]).decode().split("\n")
    corpus = []
    for f in tracked:
        if not f or f.startswith(("profiles/", "docs/history/", "BENCH_", "GPUTEST_", "SCALE_", "MULTICHIP_")) or f in (
                "VERDICT.md", "ADVICE.md", "SURVEY.md", "tools/prune_profiles.py") or f.endswith(".gz"):
            continue
        try:
            corpus.append(open(f, errors="ignore").read())
        except OSError:
            pass
    corpus.append(open("profiles/README.md").read())            # the index of what is kept (rewritten before pruning)
    text = "\n".join(corpus)
    keep = {f for f in files if f in text}
    for pattern in ("executed_mads_r*.json", "hbm_traffic_r*.json", "kernel_resources_r*.txt", "microbench_r*.json"):
        found = sorted(glob.glob(os.path.join("profiles", pattern)))
        if found:
            keep.add(os.path.basename(found[-1]))
    keep |= {f for f in files if f.startswith(ROUND) or ROUND in f}
    keep |= {"README.md", "REMOVED.md"}
    gone = [f for f in files if f not in keep]
    head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip()
    print("%d files, keeping %d, removing %d (last commit that holds them: %s)" % (len(files), len(files) - len(gone), len(gone), head))
    if "--apply" not in sys.argv:
        print("\n".join(gone))
        return
    with open("profiles/REMOVED.md", "w") as f:
        f.write("# Measurement files of rounds 1-5 that the design of record no longer cites\n\n"
                "Removed from the tree in round 6 (VERDICT round 5 item 9); every one of them is in the history:\n"
                "`git show %s:profiles/<name>`.  `docs/history/DESIGN_round5.md` and the round-5 `profiles/README.md`\n"
                "(`git show %s:profiles/README.md`) say what each one holds.\n\n" % (head, head))
        for name in gone:
            f.write("- `%s`\n" % name)
    subprocess.check_call(["git", "rm", "-q", "--"] + [os.path.join("profiles", g) for g in gone])
    subprocess.check_call(["git", "add", "profiles/REMOVED.md"])


if __name__ == "__main__":
    main()
