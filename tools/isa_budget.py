#!/usr/bin/env python3
"""Per-basic-block instruction budget of one kernel in a gfx950 .s file (hipcc -S --cuda-device-only).

    tools/isa_budget.py file.s <kernel-name substring> [--trips LABEL=N ...] [--dump LABEL]

Prints every basic block with its instruction counts by issue class (VALU, TRANS = v_exp/v_log/v_rcp/v_rsq/v_sqrt/v_sin/v_cos, MFMA, SALU, LDS, VMEM,
branch, wait/nop) and its branch targets.  With --trips the per-wave totals are the blocks' counts times the given trip
counts (blocks not named count once if they are on the entry path, i.e. default 1; give LABEL=0 to exclude a block), which
is how the per-phase budgets of DESIGN.md section 4.2 were summed against the PMC totals."""
import re
import sys

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "MFMA"
    if op.startswith(TRANS):
        return "TRANS"
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "VMEM"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
        return "BR"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "WAIT"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    return "OTHER"


CLASSES = ["VALU", "TRANS", "MFMA", "SALU", "LDS", "VMEM", "SMEM", "BR", "WAIT"]


def parse(path, flt):
    blocks = []          # (label, counts, targets, lines)
    cur = None
    inside = False
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            inside = flt in m.group(1)
            if inside:
                cur = ["entry", dict.fromkeys(CLASSES, 0), [], []]
                blocks.append(cur)
                kname = m.group(1)
            continue
        if not inside:
            continue
        m = re.match(r'^(\.LBB\w+):', line)
        if m:
            cur = [m.group(1), dict.fromkeys(CLASSES, 0), [], []]
            blocks.append(cur)
            continue
        s = line.strip()
        if not s or s.startswith((";", ".", "//")):
            if s.startswith(".Lfunc_end"):
                inside = False
            continue
        op = s.split()[0]
        c = classify(op)
        if c == "OTHER":
            continue
        cur[1][c] += 1
        cur[3].append(s.split(";")[0].rstrip())
        if c == "BR":
            t = re.search(r'(\.LBB\w+)', s)
            if t:
                cur[2].append(t.group(1))
        if op == "s_endpgm":
            inside = False
    return blocks


def main():
    path, flt = sys.argv[1], sys.argv[2]
    trips = {}
    dump = None
    args = sys.argv[3:]
    i = 0
    while i < len(args):
        if args[i] == "--trips":
            i += 1
            while i < len(args) and "=" in args[i]:
                k, v = args[i].split("=")
                trips[k] = float(v)
                i += 1
            continue
        if args[i] == "--dump":
            dump = args[i + 1]
            i += 2
            continue
        i += 1
    blocks = parse(path, flt)
    tot = dict.fromkeys(CLASSES, 0.0)
    print("%-14s %7s " % ("block", "trips") + " ".join("%5s" % c for c in CLASSES) + "  -> targets")
    for lab, cnt, tg, lines in blocks:
        short = lab.replace(".LBB", "B")
        n = trips.get(short, trips.get(lab, 1.0))
        for c in CLASSES:
            tot[c] += n * cnt[c]
        print("%-14s %7.2f " % (short, n) + " ".join("%5d" % cnt[c] for c in CLASSES) + "  -> " + ",".join(t.replace(".LBB", "B") for t in tg))
        if dump and dump in (short, lab):
            for l in lines:
                print("        " + l)
    print("%-14s %7s " % ("weighted", "") + " ".join("%5.0f" % tot[c] for c in CLASSES))


if __name__ == "__main__":
    main()
