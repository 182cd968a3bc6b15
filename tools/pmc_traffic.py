"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of bench.py, KiB units) to HBM-side bytes
per launch per kernel family.  Usage: python tools/pmc_traffic.py <fetch.csv> <write.csv> <out.json>

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): WRITE_SIZE is exact -- it calibrates on the two pure
streaming kernels of this path (gwc_warp_fwd writes 188.7 MB, pool_gather 134.2 MB); FETCH_SIZE under-reports the
16-byte-per-lane streaming reads of these kernels by 2x on gfx950 (gwc_warp_fwd reads 3.9 MB and reports 2.0 MB), so
it is doubled."""
import collections, csv, json, sys

FAMILIES = [("conv_wide16_kernel", ("conv_wide16_kernel",)), ("conv_tap16_kernel", ("conv_tap16_kernel",)),
            ("conv_gather16_kernel", ("conv_gather16_kernel",)), ("wgrad16_kernel", ("wgrad16_kernel", "wgrad_ring16_kernel")),
            ("wgrad_ring16_kernel", ("wgrad_ring16_kernel",)),
            ("wino_df_kernel", ("wino_df_kernel",)), ("wino_dfw_kernel", ("wino_dfw_kernel",)),
            ("gemm_nn_kernel", ("gemm_nn_kernel",)), ("gemm_tn_kernel", ("gemm_tn_kernel", "gemm_tn_skinny_kernel")),
            ("gwc_warp_bwd", ("gwc_warp_bwd2_kernel", "gwc_warp_bwd3_kernel")), ("lift_splat_bwd2", ("lift_splat_bwd2_kernel",)),
            ("conv_thinin_kernel", ("conv_thinin_kernel",)), ("conv_thinout_u_kernel", ("conv_thinout_u_kernel",)),
            ("wgrad_thinside_kernel", ("wgrad_thinside_kernel",)), ("softmax_row", ("softmax_row_",)),
            ("conv_fwd_dgrad", ("conv_gather_kernel", "conv_tap_kernel", "conv_taph_kernel", "conv_tapdh_kernel", "conv_thin_kernel")),
            ("conv_gather_kernel", ("conv_gather_kernel",)),
            ("conv_tap_kernel", ("conv_tap_kernel",)), ("conv_taph_kernel", ("conv_taph_kernel",)), ("conv_tapdh_kernel", ("conv_tapdh_kernel",)),
            ("wgrad_tapdh_kernel", ("wgrad_tapdh_kernel",)), ("conv_tap2_kernel", ("conv_tap2_kernel",)), ("conv_tap2up_kernel", ("conv_tap2up_kernel",)), ("wgrad_lds_kernel", ("wgrad_lds_kernel",)),
            ("gwc_warp_fwd", ("gwc_warp_fwd_kernel", "gwc_warp_fwd4_kernel", "gwc_warp_fwd5_kernel")), ("pool_gather", ("pool_gather_kernel", "pool_gather2_kernel", "pool_gather3_kernel", "pool_gather5_kernel")),
            ("gn_apply_fwd", ("gn_apply_fwd_kernel",)), ("gn_apply_bwd", ("gn_apply_bwd_kernel",)),
            ("gn2_apply_fwd", ("gn2_apply_fwd_kernel",)), ("gn2_partial_bwd", ("gn2_partial_bwd_kernel",)), ("gn2_apply_bwd", ("gn2_apply_bwd_kernel",)),
            ("wino_input_kernel", ("wino_input_kernel", "wino43_input_kernel")),
            ("wino_output_kernel", ("wino_output_kernel", "wino43_output_kernel")), ("rocblas_gemm_Cijk", ("Cijk_",))]


def load(path):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        per[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024.0)
    return per


fetch, write = load(sys.argv[1]), load(sys.argv[2])
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha16                      # the tree the counters were collected on (bench.py checks it against its own)
out = {"_doc": __doc__, "csrc_sha16": csrc_sha16(), "kernels": {}}
for fam, keys in FAMILIES:
    f = [v for k, vs in fetch.items() if any(s in k for s in keys) for v in vs]
    w = [v for k, vs in write.items() if any(s in k for s in keys) for v in vs]
    if not f or not w:
        continue
    fr, wr = sum(f) / len(f), sum(w) / len(w)
    out["kernels"][fam] = {"launches_sampled": len(f), "fetch_bytes_raw": fr, "fetch_correction": 2.0, "write_bytes": wr,
                           "hbm_bytes_per_launch": 2.0 * fr + wr}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{k:22s} n={v['launches_sampled']:4d}  fetch(raw) {v['fetch_bytes_raw'] / 1e6:9.1f} MB  write {v['write_bytes'] / 1e6:8.1f} MB  "
          f"hbm/launch {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB")
