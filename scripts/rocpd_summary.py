#!/usr/bin/env python
"""Per-kernel summary of a rocprofv3 run (`--kernel-trace --stats`, rocpd sqlite output):
what `rocprofv3 --stats` prints as kernel_stats, as CSV, for committing under profiles/.

  python scripts/rocpd_summary.py gpurun_out/<run>/<name>_results.db > profiles/<name>_kernel_stats.csv
"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    q = ("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
         "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
         "max(grid_x), max(workgroup_x) from kernels group by name order by 3 desc")
    rows = list(cur.execute(q))
    total = float(sum(r[2] for r in rows)) or 1.0
    print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPR,AGPR,SGPR,LDS,Scratch,GridX,WorkgroupX")
    for r in rows:
        print('"%s",%d,%d,%.1f,%d,%d,%.2f,%d,%d,%d,%d,%d,%d,%d' % (
            r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10], r[11], r[12]))


if __name__ == "__main__":
    main(sys.argv[1])
