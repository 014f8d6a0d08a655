"""The parity contract: every C++-path golden VCF of the reference's tests/run.py that the
scan/combine hot path produces, with the loader/query parameters run.py used for it
(reference tests/run.py:184-870 for the parameters, :935-947 for the vcf attribute order).

Each case: name, callsets json, vid json, query overrides, golden file, mode.
mode "query"  = gt_mpi_gather --produce-Broad-GVCF with vcf_attributes_order
mode "load"   = the loader's in-line combine (produce_combined_vcf): all schema attributes,
                one interval = the column partition.
"""
VCF_ATTRIBUTES_ORDER = ["END", "REF", "ALT", "BaseQRankSum", "ClippingRankSum", "MQRankSum", "ReadPosRankSum",
                        "MQ", "RAW_MQ", "MQ0", "DP", "GT", "GQ", "SB", "AD", "PL", "PGT", "PID", "MIN_DP",
                        "DP_FORMAT", "FILTER"]

FULL = [{"range_list": [{"low": 0, "high": 1000000000}]}]


def _r(lo, hi=1000000000):
    return [{"range_list": [{"low": lo, "high": hi}]}]


HT = "t0_haploid_triploid_1_2_3_triploid_deletion"

CASES = [
    # name, callsets, vid, query overrides, golden, mode
    ("t0_1_2_vcf_at_0", "t0_1_2.json", "vid.json", {"query_column_ranges": FULL}, "t0_1_2_vcf_at_0", "query"),
    ("t0_1_2_vcf_at_multiple_positions", "t0_1_2.json", "vid.json",
     {"query_column_ranges": [[12000, 12142, 12144, 12160, 12290, 12294, 14000, 17384, 18000]]},
     "t0_1_2_vcf_at_multiple_positions", "query"),
    ("t0_1_2_vcf_sites_only_at_0", "t0_1_2.json", "vid.json", {"query_column_ranges": FULL, "sites_only_query": True},
     "t0_1_2_vcf_sites_only_at_0", "query"),
    ("t0_1_2_vcf_at_12150", "t0_1_2.json", "vid.json", {"query_column_ranges": _r(12150)}, "t0_1_2_vcf_at_12150", "query"),
    ("t0_1_2_vcf_at_0_with_FILTER", "t0_1_2.json", "vid.json", {"query_column_ranges": FULL, "produce_FILTER_field": True},
     "t0_1_2_vcf_at_0_with_FILTER", "query"),
    ("t0_1_2_vcf_at_0_phased_GT_vid", "t0_1_2.json", "vid_phased_GT.json", {"query_column_ranges": FULL}, "t0_1_2_vcf_at_0", "query"),
    ("t0_1_2_loading", "t0_1_2.json", "vid.json", {}, "t0_1_2_loading", "load"),
    ("t0_1_2_as_array_loading", "t0_1_2_as_array.json", "vid_as_array.json", {}, "t0_1_2_loading", "load"),
    ("t0_overlapping_loading", "t0_overlapping.json", "vid.json", {}, "t0_overlapping", "load"),
    ("t0_overlapping_at_12202", "t0_overlapping.json", "vid.json", {"query_column_ranges": _r(12202)}, "t0_overlapping_at_12202", "query"),
    ("t0_overlapping_at_12202_partition_loading", "t0_overlapping.json", "vid.json", {"partition_begin": 12202},
     "t0_overlapping_at_12202", "load"),
    ("t6_7_8_vcf_at_0", "t6_7_8.json", "vid.json", {"query_column_ranges": FULL}, "t6_7_8_vcf_at_0", "query"),
    ("t6_7_8_loading", "t6_7_8.json", "vid.json", {}, "t6_7_8_loading", "load"),
    ("t6_7_8_vcf_sites_only_at_0", "t6_7_8.json", "vid.json", {"query_column_ranges": FULL, "sites_only_query": True},
     "t6_7_8_vcf_sites_only_at_0", "query"),
    ("t6_7_8_vcf_at_8029500", "t6_7_8.json", "vid.json", {"query_column_ranges": _r(8029500)}, "t6_7_8_vcf_at_8029500", "query"),
    ("t6_7_8_vcf_at_8029500-8029500", "t6_7_8.json", "vid.json", {"query_column_ranges": _r(8029500, 8029500)},
     "t6_7_8_vcf_at_8029500-8029500", "query"),
    ("t6_7_8_vcf_at_0_phased_GT_vid", "t6_7_8.json", "vid_phased_GT.json", {"query_column_ranges": FULL}, "t6_7_8_vcf_at_0", "query"),
    ("t6_7_8_new_field_gatk", "t6_7_8.json", "vid_MLEAC_MLEAF.json", {}, "t6_7_8_new_field_gatk.vcf", "load"),
    ("info_ops0", "info_ops.json", "vid_info_ops0.json", {}, "info_ops0.vcf", "load"),
    ("info_ops1", "info_ops.json", "vid_info_ops1.json", {}, "info_ops1.vcf", "load"),
    ("t0_1_2_DS_ID_vcf_at_0", "t0_1_2.json", "vid_DS_ID.json", {}, "t0_1_2_DS_ID_vcf_at_0", "load"),
    ("t0_with_missing_PL_SB_fields_t1", "t0_with_missing_PL_SB_fields_t1.json", "vid.json", {},
     "t0_with_missing_PL_SB_fields_t1.vcf", "load"),
    ("t0_1_2_all_asa_loading", "t0_1_2_all_asa.json", "vid_all_asa.json", {}, "t0_1_2_all_asa_loading", "load"),
    ("t0_1_2_combined", "t0_1_2_combined.json", "vid.json", {"query_column_ranges": FULL}, "t0_1_2_combined", "query"),
    ("t0_1_2_combined_loading", "t0_1_2_combined.json", "vid.json", {}, "t0_1_2_combined", "load"),
    (HT + "_loading", HT + ".json", "vid_DS_ID_phased_GT.json", {}, HT + "_loading", "load"),
    (HT + "_vcf", HT + ".json", "vid_DS_ID_phased_GT.json", {"query_column_ranges": FULL}, HT + "_vcf", "query"),
    (HT + "_vcf_produce_GT", HT + ".json", "vid_DS_ID_phased_GT.json", {"query_column_ranges": FULL, "produce_GT_field": True},
     HT + "_vcf_produce_GT", "query"),
    (HT + "_vcf_produce_GT_for_min_value_PL", HT + ".json", "vid_DS_ID_phased_GT.json",
     {"query_column_ranges": FULL, "produce_GT_field": True, "produce_GT_with_min_PL_value_for_spanning_deletions": True},
     HT + "_vcf_produce_GT_for_min_value_PL", "query"),
    (HT + "_vcf_sites_only", HT + ".json", "vid_DS_ID_phased_GT.json", {"query_column_ranges": FULL, "sites_only_query": True},
     HT + "_vcf_sites_only", "query"),
    ("min_PL_spanning_deletion_load_stdout", "min_PL_spanning_deletion.json", "vid_phased_GT.json", {},
     "min_PL_spanning_deletion_load_stdout", "load"),
    ("min_PL_spanning_deletion_vcf_no_min_PL", "min_PL_spanning_deletion.json", "vid_phased_GT.json",
     {"query_column_ranges": FULL, "produce_GT_field": True}, "min_PL_spanning_deletion_vcf_no_min_PL", "query"),
    ("min_PL_spanning_deletion_vcf", "min_PL_spanning_deletion.json", "vid_phased_GT.json",
     {"query_column_ranges": FULL, "produce_GT_field": True, "produce_GT_with_min_PL_value_for_spanning_deletions": True},
     "min_PL_spanning_deletion_vcf", "query"),
]

# Goldens of the hot path that are NOT covered yet, with the reason (kept visible on purpose).
UNCOVERED = {
}


# ---- gt_mpi_gather --print-calls (the reference's "calls" goldens, tests/run.py:184-720) ----------------------------------------
CALLS_ATTRIBUTES = ["REF", "ALT", "BaseQRankSum", "MQ", "RAW_MQ", "MQ0", "ClippingRankSum", "MQRankSum", "ReadPosRankSum", "DP", "GT", "GQ",
                    "SB", "AD", "PL", "DP_FORMAT", "MIN_DP", "PID", "PGT"]   # run.py:50 (the query template's "attributes")
CALLS_ATTRIBUTES_DS_ID = CALLS_ATTRIBUTES + ["DS", "ID"]


def _rl(*pairs):
    return [{"range_list": [{"low": lo, "high": hi} for lo, hi in pairs]}]


CALLS_CASES = [
    # name (= golden file), callsets, vid, query_column_ranges, attributes
    ("t0_1_2_calls_at_0", "t0_1_2.json", "vid.json", FULL, CALLS_ATTRIBUTES),
    ("t0_1_2_calls_at_multiple_positions", "t0_1_2.json", "vid.json", [[12000, 12142, 12144, 12160, 12290, 12294, 14000, 17384, 18000]], CALLS_ATTRIBUTES),
    ("t0_1_2_calls_at_12100", "t0_1_2.json", "vid.json", _rl((12100, 12100)), CALLS_ATTRIBUTES),
    ("t0_1_2_calls_at_12100_12141", "t0_1_2.json", "vid.json", _rl((12100, 12100), (12141, 12141)), CALLS_ATTRIBUTES),
    ("t0_1_2_calls_at_12100_12141_12150", "t0_1_2.json", "vid.json", _rl((12100, 12100), (12141, 12141), (12150, 12150)), CALLS_ATTRIBUTES),
    ("t0_1_2_calls_at_12100_12141_to_12150", "t0_1_2.json", "vid.json", _rl((12100, 12100), (12141, 12150)), CALLS_ATTRIBUTES),
    ("t0_1_2_calls_at_12100_12141_to_12150_12300_17384", "t0_1_2.json", "vid.json", _rl((12100, 12100), (12141, 12150), (12300, 12300), (17384, 17384)),
     CALLS_ATTRIBUTES),
    ("t0_1_2_calls_at_0_with_PL_only", "t0_1_2.json", "vid.json", FULL, ["PL"]),
    ("t0_1_2_calls_at_12150", "t0_1_2.json", "vid.json", _r(12150), CALLS_ATTRIBUTES),
    ("t6_7_8_calls_at_0", "t6_7_8.json", "vid.json", FULL, CALLS_ATTRIBUTES),
    ("t6_7_8_calls_at_8029500", "t6_7_8.json", "vid.json", _r(8029500), CALLS_ATTRIBUTES),
    ("t0_1_2_calls_at_0_phased_GT", "t0_1_2.json", "vid_phased_GT.json", FULL, CALLS_ATTRIBUTES),
    ("t0_1_2_calls_at_12150_phased_GT", "t0_1_2.json", "vid_phased_GT.json", _r(12150), CALLS_ATTRIBUTES),
    ("t6_7_8_calls_at_0_phased_GT", "t6_7_8.json", "vid_phased_GT.json", FULL, CALLS_ATTRIBUTES),
    ("t6_7_8_calls_at_8029500_phased_GT", "t6_7_8.json", "vid_phased_GT.json", _r(8029500), CALLS_ATTRIBUTES),
    ("test_new_fields_MLEAC_only.json", "t6_7_8.json", "vid_MLEAC_MLEAF.json", FULL, ["MLEAC"]),
    ("t0_1_2_DS_ID_calls_at_0", "t0_1_2.json", "vid_DS_ID.json", FULL, CALLS_ATTRIBUTES_DS_ID),
    ("t0_1_2_DS_ID_calls_at_0_phased_GT", "t0_1_2.json", "vid_DS_ID_phased_GT.json", FULL, CALLS_ATTRIBUTES_DS_ID),
    ("t0_with_missing_PL_SB_fields_t1_calls.json", "t0_with_missing_PL_SB_fields_t1.json", "vid.json", FULL, CALLS_ATTRIBUTES),
]
