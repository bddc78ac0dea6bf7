import csv, sys, re, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
gaps, durs = collections.defaultdict(list), collections.defaultdict(list)
for p, r in zip(rows, rows[1:]):
    m = re.search(r"writer<(\d+), (\d+)>", p["Kernel_Name"])
    if m and "tiny" in r["Kernel_Name"]:
        key = (int(m.group(1)), int(m.group(2)))
        gaps[key].append((int(r["Start_Timestamp"]) - int(p["End_Timestamp"])) / 1e3)
        durs[key].append((int(p["End_Timestamp"]) - int(p["Start_Timestamp"])) / 1e3)
for key in sorted(gaps, key=lambda k: (k[1], k[0])):
    g, d = sorted(gaps[key]), sorted(durs[key])
    print("writer %4d MB nt=%d : writer %7.1f us (%.2f TB/s)   gap to next kernel median %5.2f us  min %5.2f max %5.2f" % (
        key[0], key[1], d[len(d) // 2], key[0] * 1.048576 / d[len(d) // 2], g[len(g) // 2], g[0], g[-1]))
