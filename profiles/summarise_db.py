#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (`-o name` output) into a per-LAUNCH-CLASS table: one row per
(kernel symbol, workgroups in the grid, workgroup size), so that a command which launches one symbol at several sizes (the
default bench.py does: headline + few-window + training legs) never averages them together (VERDICT r03 Weak 6).
Usage: summarise_db.py results.db [--timeline MARKER_KERNEL_SUBSTRING]"""
import sqlite3
import sys


def tables(c):
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    return [t for t in tabs if "kernel_dispatch" in t][0], [t for t in tabs if "kernel_symbol" in t][0]


def main():
    db = sqlite3.connect(sys.argv[1])
    c = db.cursor()
    kd, ks = tables(c)
    wgs = "(d.grid_size_x / d.workgroup_size_x) * (d.grid_size_y / d.workgroup_size_y) * (d.grid_size_z / d.workgroup_size_z)"
    rows = list(c.execute(f"select s.kernel_name, {wgs} as wgs, d.workgroup_size_x * d.workgroup_size_y * d.workgroup_size_z as wg, count(*), "
                          f"sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e6, min(d.end-d.start)/1e6, max(d.end-d.start)/1e6 "
                          f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name, wgs, wg order by 5 desc"))
    tot = sum(r[4] for r in rows)
    print("kernel,workgroups,workgroup_size,calls,total_ms,avg_ms,min_ms,max_ms,percent")
    for r in rows:
        print('"%s",%d,%d,%d,%.3f,%.4f,%.4f,%.4f,%.2f' % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], 100 * r[4] / tot))
    if len(sys.argv) > 3 and sys.argv[2] == "--timeline":
        mark = sys.argv[3]
        d = list(c.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x from {kd} d "
                           f"join {ks} s on d.kernel_id=s.id order by d.start"))
        idx = [i for i, r in enumerate(d) if mark in r[0]]
        seg = d[idx[-2] + 1: idx[-1] + 1]
        t0 = seg[0][1]
        print("\nstart_ms,dur_ms,kernel,grid_x,grid_y,wg")
        for r in seg:
            print("%.3f,%.4f,\"%s\",%d,%d,%d" % ((r[1] - t0) / 1e6, (r[2] - r[1]) / 1e6, r[0], r[3] // r[5], r[4], r[5]))


if __name__ == "__main__":
    main()
