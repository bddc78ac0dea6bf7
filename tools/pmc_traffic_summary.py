"""gpurun_out/traffic_<tag>_* (tools/pmc_traffic.sh) -> the JSON bench.py reads as profiles/r03_traffic.json."""
import csv, glob, json, statistics, sys
tag = sys.argv[1]


def counter(name, ctr, pat):
    vals = {}
    for f in glob.glob(f"gpurun_out/traffic_{tag}_{name}_{ctr}/*/*_counter_collection.csv") + glob.glob(f"/root/repo/gpurun_out/traffic_{tag}_{name}_{ctr}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                vals.setdefault(r["Dispatch_Id"], 0.0)
                vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
    return statistics.median(vals.values()) if vals else None


def entry(name, pat, kernel, alg_bytes, note):
    f, w = counter(name, "FETCH_SIZE", pat), counter(name, "WRITE_SIZE", pat)
    hbm = None if f is None or w is None else (2.0 * f + w) * 1024.0      # counters are in KB; FETCH_SIZE x2: gfx950 correction
    return {"kernel": kernel, "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "hbm_bytes_per_launch": hbm,
            "algorithmic_bytes_per_launch": alg_bytes, "ratio": None if hbm is None else hbm / alg_bytes, "note": note}


B, P, heads, Nqp, hd, A = 4, 343, 4, 352, 12, 20
qkv = 3 * B * P * heads * Nqp * hd * 2
o = B * P * Nqp * heads * hd * 2
out = {
    "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/pmc_traffic.sh), median per launch; FETCH_SIZE doubled per "
           "MI355X_MICROARCH.md (gfx950 tallies 128-byte read requests at 64 bytes); Infinity-Cache hits are included: fabric-side traffic, an "
           "upper bound on HBM traffic.  The attention launches use 4- and 8-byte-per-lane accesses (LDS-DMA dword gathers, 8-byte Q' pieces "
           "and O stores), for which the doubling rule is not established: read the figure as an upper bound.",
    "attn_fwd_stage0": entry("attn", "k_win_attn_fwd", "k_win_attn_fwd<1,1,8,1,...,ZREF,DMA> un-shifted, stage 0: 1372 windows x 4 heads", qkv + o,
                             "algorithmic = q, k, v read once + o written once (bf16); the bias tables (14 KB per head) stay in L2"),
    "attn_fwd_stage0_shifted": entry("attns", "k_win_attn_fwd", "same, shifted block (masked kernel)", qkv + o, ""),
    "conv3d_dec2": entry("conv", "k_conv3d_halo", "k_conv3d_halo<3,6> (6x6x16 bricks), Cin 144 -> Cout 48, 4 x 48^3 voxels", 4 * 48 ** 3 * (144 + 48) * 2 + 27 * 144 * 48 * 2,
                         "algorithmic = input + output once + weights"),
}
print(json.dumps(out, indent=1))
