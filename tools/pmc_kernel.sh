#!/bin/bash
# HBM / L2 counters for the kernels matching $1 while running "$2..." (separate passes: FETCH_SIZE, WRITE_SIZE, L2 hit/miss)
set -e
export TMPDIR=/tmp
PAT=$1; shift
OUT=gpurun_out/pmc_k
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -- "$@" > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -- "$@" > $OUT/w.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/h -- "$@" > $OUT/h.log 2>&1
python3 - "$OUT" "$PAT" <<'PY'
import csv, glob, sys, collections, re
out, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("f", "w", "h"):
    for f in glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("sfcvit::", "")
            if not re.search(pat, k):
                continue
            agg[re.sub(r"\(.*", "", k)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(agg.items()):
    m = {c: sum(x) / len(x) for c, x in v.items()}
    fetch = 2 * m.get("FETCH_SIZE", 0) * 1024 / 1e6      # KiB -> bytes, x2 (MI355X_MICROARCH.md: 128-B requests tallied as 64 B)
    wr = m.get("WRITE_SIZE", 0) * 1024 / 1e6
    hit, miss = m.get("TCC_HIT_sum", 0), m.get("TCC_MISS_sum", 0)
    print(f"{k:50s} fetch {fetch:8.1f} MB  write {wr:8.1f} MB  L2 hit rate {hit / max(1.0, hit + miss):.3f}  L2 req {m.get('TCC_REQ_sum', 0):.3e}")
PY
