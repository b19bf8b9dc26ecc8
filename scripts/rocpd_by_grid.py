#!/usr/bin/env python
"""Per (kernel, grid size) summary of a rocprofv3 --kernel-trace run (rocpd sqlite): one kernel launched with several geometries.
  python scripts/rocpd_by_grid.py <results.db>"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
q = ("select name, grid_x, count(*), avg(end-start), min(end-start), max(vgpr_count), max(lds_size) from kernels "
     "group by name, grid_x order by name, grid_x")
print("Name,GridX,Calls,AverageUs,MinUs,VGPR,LDS")
for r in cur.execute(q):
    print('"%s",%d,%d,%.1f,%.1f,%d,%d' % (r[0][:90], r[1], r[2], r[3] / 1e3, r[4] / 1e3, r[5], r[6]))
