#!/usr/bin/env python3
"""Print the per-kernel summary of a rocprofv3 results database (the `top_kernels` view): name, calls, mean us, percent."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for name, calls, total, avg, pct in db.execute("select * from top_kernels"):
    if pat in name:
        print(f"{name[:100]:100s} {calls:6d} {avg:9.1f} us {pct:6.2f} %")
