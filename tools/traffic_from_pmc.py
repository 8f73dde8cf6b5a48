#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) over
`bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing` into profiles/traffic.json: HBM-side bytes per
launch for every sfcvit kernel.  rocprofv3 reports both counters in KiB; FETCH_SIZE is doubled (on gfx950 it tallies
the 128-B requests of 16-B/lane streaming reads as 64 B, MI355X_MICROARCH.md "HBM"), WRITE_SIZE is taken as is.

    python tools/traffic_from_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <build tag> [workload]

The file is keyed by bench.py workload (default vit_b16_224_hilbert); other workloads' entries are kept.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            name = re.sub(r"^void ", "", name)
            name = name.replace("sfcvit::(anonymous namespace)::", "").replace("p8::", "")
            name = re.sub(r"\(.*$", "", name)
            tot[name] += float(row["Counter_Value"])
            cnt[name] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    workload = sys.argv[4] if len(sys.argv) > 4 else "vit_b16_224_hilbert"
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        doc = {}
    doc.setdefault("workloads", {})
    out = {"_comment": "HBM-side traffic per launch from rocprofv3 PMC (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE "
                       "passes over `bench.py --steps 3 --warmup 1`), averaged over all launches of the kernel. bytes = counter "
                       "value x 1024; FETCH_SIZE is then doubled as MI355X_MICROARCH.md prescribes for 16-B/lane streaming "
                       "reads on gfx950 (128-B requests tallied as 64 B); WRITE_SIZE is taken as is. Infinity-Cache hits "
                       "are included in FETCH_SIZE.",
           "build": sys.argv[3] if len(sys.argv) > 3 else "", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if not ("gemm" in k or "attn" in k or "ln_" in k or "pe_" in k or "pe2_" in k or "adamw" in k or "colsum" in k or "transpose" in k or "gather" in k or "reduce" in k or "gelu" in k):
            continue
        f, nf = fetch.get(k, (0.0, 0))
        w, _ = write.get(k, (0.0, 0))
        out["kernels"][k] = {"fetch_bytes_raw": f * 1024, "fetch_bytes_corrected": 2 * f * 1024, "write_bytes": w * 1024,
                             "traffic_bytes": 2 * f * 1024 + w * 1024, "launches_sampled": nf}
    doc["_comment"] = out.pop("_comment") + " Keyed by bench.py workload."
    doc["workloads"][workload] = out
    with open(path, "w") as fo:
        json.dump(doc, fo, indent=1)
    for k, v in out["kernels"].items():
        print(f"{k:60s} {v['traffic_bytes'] / 1e6:10.1f} MB/launch ({v['launches_sampled']} launches)")


if __name__ == "__main__":
    main()
