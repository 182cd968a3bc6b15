"""Compact summary of a rocprofv3 *_kernel_stats.csv: total, family roll-up and the top kernels (ms per step).
usage: python tools/prof_summary.py <kernel_stats.csv> <steps> [top_n] [name-regex: list only matching kernels]"""
import csv
import re
import sys

FAMILIES = [
    ("bf16-storage conv fwd+dgrad (conv_gather16_kernel, conv_igemm16_kernel, conv_tap16_kernel, conv_wide16_kernel, pack16 / pack_tap16)", r"conv_gather16|conv_igemm16|conv_tap16|conv_wide16|pack16_kernel|pack_tap16"),
    ("bf16-storage weight gradient (wgrad16_kernel, wgrad_ring16_kernel)", r"wgrad16|wgrad_ring16"),
    ("direct conv fwd+dgrad (conv_gather_kernel, conv_igemm_kernel, conv_tap_kernel, conv_taph_kernel, conv_tapdh_kernel, conv_tap2_kernel, conv_tap2up_kernel, conv_pw32_kernel, conv_thin*_kernel)", r"conv_gather_kernel|conv_igemm_kernel|conv_tap_kernel|conv_taph_kernel|conv_tapdh_kernel|conv_tap2_kernel|conv_tap2up_kernel|conv_pw32_kernel|conv_thin"),
    ("own MFMA GEMM family (gemm_nn / gemm_tn / skinny / sum: BRI products, k == s deconvs, pointwise weight gradients)", r"gemm_"),
    ("depth-fused Winograd contraction, own MFMA kernels (wino_df_kernel fwd/dgrad, wino_dfw_kernel wgrad, + pack / sum / reduce)", r"wino_df"),
    ("Winograd transforms (wino*_input/output/output_adjoint/weight*)", r"wino"),
    ("rocBLAS / hipBLASLt GEMMs (Cijk_*: Winograd frequency GEMMs, BRI products, image-branch pointwise convs)", r"Cijk_"),
    ("weight gradient, direct (wgrad_lds/wgrad_thin/wgrad_1x1/wgrad_cf/wgrad_kernel + reduce)", r"wgrad"),
    ("GroupNorm / BatchNorm (gn_*, gn2_*, bn_*)", r"gn_|gn2_|bn_"),
    ("weight packing (pack_*)", r"pack_"),
    ("cost volume, lift/splat, scatter prep, DCN, softmax, losses, trilinear, image-branch ops", r"gwc_|pool_|lift_|voxel_index|histogram|scan_|fill_kernel|canonicalise|dcn_|softmax_axis|softmax_row|occ_loss|trilinear|bri_|dw_|swish|chan_|adamw|sumsq"),
]


def main():
    path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    rows = list(csv.DictReader(open(path)))
    if len(sys.argv) > 4:
        for r in rows:
            if re.search(sys.argv[4], r["Name"]):
                print(f"{float(r['TotalDurationNs']) / 1e6 / steps:9.3f} ms/step {int(r['Calls']) / steps:7.1f} calls/step  avg "
                      f"{float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:110]}")
        return
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"total kernel time: {tot / 1e6 / steps:.1f} ms/step over {steps:g} steps")
    print("\nfamily roll-up (ms/step, kernel launches/step):")
    left = list(rows)
    for label, pat in FAMILIES:
        hit = [r for r in left if re.search(pat, r["Name"])]
        left = [r for r in left if not re.search(pat, r["Name"])]
        print(f"  {sum(float(r['TotalDurationNs']) for r in hit) / 1e6 / steps:8.2f} ms {sum(int(r['Calls']) for r in hit) / steps:8.1f}  {label}")
    print(f"  {sum(float(r['TotalDurationNs']) for r in left) / 1e6 / steps:8.2f} ms {sum(int(r['Calls']) for r in left) / steps:8.1f}  ATen elementwise / reductions / copies and everything else")
    print()
    for r in rows[:top]:
        print(f"{float(r['TotalDurationNs']) / 1e6 / steps:9.2f} ms/step {int(r['Calls']) / steps:7.1f} calls/step  avg {float(r['AverageNs']) / 1e3:9.1f} us  "
              f"{float(r['Percentage']):6.2f}%  {r['Name'][:130]}")


if __name__ == "__main__":
    main()
