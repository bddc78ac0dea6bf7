import csv, sys, re, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
gaps, durs = collections.defaultdict(list), collections.defaultdict(list)
for p, r in zip(rows, rows[1:]):
    m = re.search(r"mover<(\d+), (\d+), (\d+)>", p["Kernel_Name"])
    if m and "tiny" in r["Kernel_Name"]:
        key = tuple(int(v) for v in m.groups())
        gaps[key].append((int(r["Start_Timestamp"]) - int(p["End_Timestamp"])) / 1e3)
        durs[key].append((int(p["End_Timestamp"]) - int(p["Start_Timestamp"])) / 1e3)
for key in gaps:
    g, d = sorted(gaps[key]), sorted(durs[key])
    print("read %3d MB write %3d MB in %2d-byte pieces: %7.1f us   gap to the next kernel median %5.2f us (min %5.2f, max %5.2f)" % (
        key[0], key[1], key[2], d[len(d) // 2], g[len(g) // 2], g[0], g[-1]))
