"""profiles/r03_streaming.md: every kernel of the cfg1 step WITHOUT MFMA work (or with negligible MFMA work) against the bytes it has
to move: algorithmic bytes per step / time per step / 6.3 TB/s (the achievable HBM rate, MI355X_MICROARCH.md).
usage: tools/streaming_table.py <cfg1 kernel_stats.csv> <steps in the trace>"""
import csv, sys
MB = 1e6
B, T0, T1, T2, T3 = 4, 48 ** 3, 24 ** 3, 12 * 12 * 24, 6 * 6 * 24          # tokens per volume at the four resolutions
rows0, rows1, rows2 = 1372 * 352, 256 * 352, 64 * 352                     # window rows (B x P x Nqp) of stages 0 / 1 / 2
x_in = B * 96 ** 3 * 4
a0, a1, a2, a3 = B * T0 * 48 * 2, B * T1 * 96 * 2, B * T2 * 192 * 2, B * T3 * 384 * 2        # bf16 activations
logits = B * 96 ** 3 * 2 * 4
planes = B * T0 * 54 * 2
model = [  # (name substring, launches per step, bytes per step, what)
    ("k_patch_embed<0", 1, x_in, "BatchNorm statistics of conv_k2s2(x): reads the f32 volume"),
    ("k_patch_embed<1", 1, x_in + a0, "conv_k2s2 + BN apply: reads the volume, writes 48-channel tokens"),
    ("k_swin_qkv_fwdILi2", 4, 4 * (a0 + 3 * rows0 * 24 * 4), "C = 48 gather + LN + QKV: reads x, writes q | k | v (24-byte rows)"),
    ("k_proj_mlp_fwd_wideILi3", 4, 4 * (rows0 * 96 + 2 * a0), "C = 48 proj + MLP: reads o and x, writes y"),
    ("k_qkv_fwd_wsILi6", 4, 4 * (a1 + 3 * rows1 * 192), "C = 96 QKV (weight-stationary form, round 3; was k_qkv_fwd_wide<6>: 24.0 us per launch)"),
    ("k_proj_mlp_fwd_wideILi6", 4, 4 * (rows1 * 192 + 2 * a1), "C = 96 proj + MLP"),
    ("k_qkv_fwd_wsILi12", 4, 4 * (a2 + 3 * rows2 * 384), "C = 192 QKV (weight-stationary form; was k_qkv_fwd_wide<12>: 19.4 us per launch at 155 MB of L2 weight reads)"),
    ("k_proj_mlp_fwd_wideILi12", 4, 4 * (rows2 * 384 + 2 * a2), "C = 192 proj + MLP"),
    ("k_patch_merge_fwdILi12", 2, (a0 + a1) + (a1 + a2), "patch merging enc0 -> enc1, enc1 -> enc2"),
    ("k_patch_merge_fwdILi24", 1, a2 + a3, "patch merging enc2 -> bottleneck"),
    ("k_upcat_fwd", 3, (a3 + a2 + B * T2 * 576 * 2) + (a2 + a1 + B * T1 * 288 * 2) + (a1 + a0 + B * T0 * 144 * 2), "upsample + concat + BN affine + LeakyReLU, three decoder stages: write of the normalised concat tensor"),
    ("k_upcat_stats<2>", 1, a1 + a0, "BatchNorm statistics of the last stage's concat tensor from its sources (forms 127 MB of values)"),
    ("k_upcat_stats<4>", 1, a2 + a1, "... middle stage"),
    ("k_upcat_stats<8>", 1, a3 + a2, "... first stage"),
    ("k_uphead_stats", 1, a0, "head BatchNorm statistics on the low-resolution tensor (27-point product stencil)"),
    ("k_uphead_taps", 1, a0 + planes, "head: per-voxel GEMM to 54 tap planes (fp16)"),
    ("k_uphead_gather", 1, planes + logits, "head: 27-point gather of the planes into f32 logits at 96^3"),
    ("k_uphead_adjoint_brick", 1, logits + planes, "head backward: adjoint of the gather"),
    ("k_gemm_tn", 1, planes + a0, "head weight gradient (TN GEMM over 442 k tokens)"),
    ("k_dice_focal_stats", 1, logits + B * 96 ** 3 * 4, "loss value: reads logits + labels"),
    ("k_dice_focal_grad", 1, 2 * logits + B * 96 ** 3 * 4, "loss gradient: reads logits + labels, writes dlogits"),
    ("k_bn_stats", 1, a0, "BatchNorm statistics (patch embedding output)"),
]
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
print("| kernel | launches / step | µs / step | MB / step | TB/s | of 6.3 TB/s | what / why |")
print("|---|---|---|---|---|---|---|")
tot_us = 0.0
for key, n, nbytes, what in model:
    m = [r for r in rows if key in r["Name"]]
    if not m:
        continue
    us = sum(float(r["TotalDurationNs"]) for r in m) / steps / 1e3
    tot_us += us
    tbs = nbytes / (us * 1e-6) / 1e12
    print(f"| `{key}` | {n} | {us:.1f} | {nbytes / MB:.0f} | {tbs:.2f} | {tbs / 6.3:.2f} | {what} |")
small = [("k_bn_finalize", "5 launches: one workgroup per channel, latency of the partial-sum round trip"), ("k_dice_focal_finalize", "single workgroup"),
         ("k_head_grads", "single workgroup: head parameter-gradient algebra"), ("k_gemm_tn_reduce", "split partial reduction"), ("k_adamw_multi", "142 978 parameters"),
         ("k_reduce_rows", ""), ("k_uphead_fold", "single workgroup"), ("multi_tensor_apply", "BatchNorm step counters")]
for key, why in small:
    m = [r for r in rows if key in r["Name"]]
    if m:
        us = sum(float(r["TotalDurationNs"]) for r in m) / steps / 1e3
        tot_us += us
        print(f"| `{key}` | {sum(int(r['Calls']) for r in m) / steps:.0f} | {us:.1f} | - | - | - | latency-bound: {why} |")
print(f"\nsum of these rows: {tot_us:.0f} µs per step")
