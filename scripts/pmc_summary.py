#!/usr/bin/env python
"""Per-kernel mean of rocprofv3 --pmc counters from the *_counter_collection.csv files of scripts/gpu_pmc.sh.
  python scripts/pmc_summary.py gpurun_out/<tag>/pmc_<config> > profiles/<name>_pmc.csv
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as
64 bytes (MI355X_MICROARCH.md, HBM section), so hbm_read_bytes = 2 * FETCH_SIZE * 1024 for wide coalesced
streams; WRITE_SIZE is taken at face value (uncalibrated)."""
import collections
import csv
import glob
import os
import sys


def main(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "p*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for k in acc for c in acc[k]})
    print(",".join(["Kernel", "Dispatches"] + counters + ["hbm_read_bytes(2xFETCH)", "hbm_write_bytes"]))
    for k in sorted(acc, key=lambda k: -sum(acc[k].get("SQ_BUSY_CYCLES", acc[k].get("FETCH_SIZE", [0])))):
        n = max(len(v) for v in acc[k].values())
        means = {c: (sum(acc[k][c]) / len(acc[k][c]) if acc[k].get(c) else float("nan")) for c in counters}
        rd = 2 * means.get("FETCH_SIZE", float("nan")) * 1024
        wr = means.get("WRITE_SIZE", float("nan")) * 1024
        print(",".join(['"%s"' % k, str(n)] + ["%.1f" % means[c] for c in counters] + ["%.0f" % rd, "%.0f" % wr]))


if __name__ == "__main__":
    main(sys.argv[1])
