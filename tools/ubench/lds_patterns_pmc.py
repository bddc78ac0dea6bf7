"""Per-dispatch bank-conflict share of tools/ubench/lds_patterns under rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE."""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
names = ["tr-read plain 32B rows", "tr-read Oimg half-swap (o_tr)", "b64 read Oimg half-swap (o_rd)", "b64 read plain 32B rows", "tr-read Qimg swizzled (q_tr)",
         "tr-read plain 64B rows", "b128 read Qimg swizzled (q_rd)", "b64 write exch slot (ex_wr)", "tr-read exch slot (ex_tr)", "b128 write dqp", "b128 read dqp",
         "b128 read lse/delta"]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "copyBuffer" in r["Kernel_Name"]:
        continue
    d.setdefault(int(r["Dispatch_Id"]), {}).setdefault(r["Counter_Name"], 0.0)
    d[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
for i, (k, v) in enumerate(sorted(d.items())):
    act = v.get("SQ_LDS_IDX_ACTIVE", 0)
    print("%-36s conflict cycles / LDS-active cycles %.3f   (active %d)" % (names[i] if i < len(names) else "?", v.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, act), act))
