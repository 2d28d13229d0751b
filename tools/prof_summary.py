#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases (kernel-trace stats + PMC passes) into text."""
import glob
import os
import sqlite3
import sys

out, tag = sys.argv[1], sys.argv[2]


def db(sub):
    f = glob.glob(os.path.join(out, sub, "**", "*.db"), recursive=True)
    return sqlite3.connect(f[0]) if f else None


def short(name):
    return name.replace("void ", "").replace("efts::", "")[:72]


print(f"# rocprofv3 summary {tag}: python bench.py --steps {os.environ.get('STEPS', '5')} --warmup 2 --no-cpu-baseline --parity-mode 0 --call-modes 0 "
      f"--precision {os.environ.get('PREC', 'bf16')} --workload {os.environ.get('WL', 'fwd64')} {os.environ.get('BARGS', '')}")
con = db("trace")
if con:
    print("\n## kernel-trace --stats: top kernels (name, calls, total, average [as reported by rocprofv3, us], %)")
    for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 16"):
        print(f"{short(r[0]):72s} {r[1]:6d} {r[2]:14.1f} {r[3]:10.2f} {r[4]:6.2f}")
    print("\n## kernel-trace: efts kernels by grid size (name, grid, calls, avg duration us)")
    try:
        for r in con.execute("select name, grid_x * grid_y * grid_z, count(*), avg(duration) / 1000.0, sum(duration) / 1000.0, "
                             "max(vgpr_count), max(lds_size) from kernels where name like '%efts::%' "
                             "group by name, grid_x, grid_y, grid_z order by 5 desc limit 24"):
            print(f"{short(r[0]):60s} grid={r[1]:9d} n={r[2]:4d} avg={r[3]:9.2f} us total={r[4]:10.1f} us vgpr={r[5]} lds={r[6]}")
    except Exception as e:
        print("(kernels view unavailable:", e, ")")
for sub in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write", "pmc_l2"):
    con = db(sub)
    if not con:
        if os.path.exists(os.path.join(out, sub + ".log")):
            print(f"\n## {sub}: no database (pass failed, see {sub}.log)")
        else:
            print(f"\n## {sub}: not collected (trace-only run: NOPMC=1)")
        continue
    print(f"\n## {sub}: average counter value per dispatch (efts kernels only)")
    q = ("select kernel_name, counter_name, avg(value), count(*), grid_size from counters_collection "
         "where kernel_name like '%efts::%' group by kernel_name, grid_size, counter_name order by kernel_name, grid_size")
    for r in con.execute(q):
        print(f"{short(r[0]):72s} {r[1]:28s} {r[2]:18.1f}  n={r[3]} grid={r[4]:.0f}")

# ---- timeline of the last traced step: kernels in start order with the idle gap before each one on the device
con = db("trace")
if con and os.environ.get("TIMELINE"):
    cur = con.execute("select * from kernels limit 1")
    cols = [c[0] for c in cur.description]
    print("\n## kernels view columns:", ", ".join(cols))
    qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
    rows = con.execute(f"select name, start, end, {qcol or '0'}, grid_x * grid_y * grid_z from kernels order by start").fetchall()
    n = int(os.environ["TIMELINE"])
    rows = rows[-n:]
    t0 = rows[0][1]
    print(f"\n## last {n} kernels in start order: start us (from the first), duration us, gap to the latest end so far us, {qcol}, grid, name")
    latest = rows[0][1]
    busy = 0.0
    for name, s, e, q, g in rows:
        gap = (s - latest) / 1000.0
        print(f"{(s - t0) / 1000.0:10.1f} {(e - s) / 1000.0:8.1f} {gap:8.1f}  q={q} grid={g:8d} {short(name)[:60]}")
        if e > latest:
            busy += (e - max(s, latest)) / 1000.0
            latest = e
    print(f"span {(latest - t0) / 1000.0:.1f} us, device busy (union of kernel intervals) {busy:.1f} us")
