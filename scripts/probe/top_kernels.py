"""Print per-kernel call count / average duration from a rocprofv3 results .db: top_kernels.py results.db [filter]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for name, calls, total, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    if flt in name:
        print(f"{avg / 1e3 if avg > 1e4 else avg:10.2f} {'us' if avg > 1e4 else 'ns?'}  x{calls:5d}  {pct:5.1f}%  {name[:110]}", flush=True)
