#!/usr/bin/env python3
"""Per kernel of a gfx950 .s file: global loads, and the number of 'round trips' = wait points (s_waitcnt vmcnt) that follow
at least one global load issued since the previous wait point.  A kernel with about as many round trips as loads waits for
every load separately (conditional loads in their own basic blocks): tools/isa_waits.py file.s [name filter]"""
import re, sys
name = None; loads = trips = pend = 0; rows = []
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for line in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):', line)
    if m:
        name = m.group(1); loads = trips = pend = 0; continue
    if name is None: continue
    if 'global_load' in line or 'buffer_load' in line: loads += 1; pend += 1
    elif 's_waitcnt' in line and 'vmcnt' in line:
        if pend: trips += 1
        pend = 0
    elif 's_endpgm' in line:
        if flt in name: rows.append((name, loads, trips))
        name = None
for n, l, t in rows: print(f"{n[:90]:90s} loads {l:4d}  round trips {t:4d}")
