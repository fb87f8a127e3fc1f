#!/usr/bin/env python3
"""Stamp a PMC summary under profiles/ with when and from which kernel sources it was taken:
profiles/latest_pmc.json[NAME] = {date, commit, command, csrc_sha16}.  bench.py withholds `traffic` when the
stamp's csrc_sha16 is not the hash of the csrc/ it runs (a byte count of other kernels is not evidence).
usage: python tools/pmc_stamp.py NAME "date text" COMMIT "command" ["rows=N batch=B mode=M"]   (run where the summary was
taken from: the hash is of the working tree's csrc/; the workload key -- bench.workload_key -- says which shapes the
per-launch byte counts belong to: bench.py withholds `traffic` from a record of another workload)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

name, date, commit, command = sys.argv[1:5]
workload = sys.argv[5] if len(sys.argv) > 5 else None
path = os.path.join(ROOT, "profiles", "latest_pmc.json")
meta = json.load(open(path)) if os.path.exists(path) else {}
meta[name] = {"date": date, "commit": commit, "command": command, "csrc_sha16": bench.csrc_hash()}
if workload:
    meta[name]["workload"] = workload
# (several stamps in one GPU call: start from the copy an earlier stamp of this call left under gpurun_out/)
carry = os.path.join(ROOT, "gpurun_out", "latest_pmc.json")
if os.path.exists(carry):
    prev = json.load(open(carry))
    for k, v in prev.items():
        if k != name and v.get("csrc_sha16") == meta[name]["csrc_sha16"]:
            meta[k] = v
json.dump(meta, open(path, "w"), indent=1)
# (the GPU box returns only gpurun_out/: leave a copy there to carry the stamp back into profiles/)
out = os.path.join(ROOT, "gpurun_out")
if os.path.isdir(out):
    json.dump(meta, open(os.path.join(out, "latest_pmc.json"), "w"), indent=1)
print(json.dumps(meta[name]))
