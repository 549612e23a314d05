#!/usr/bin/env python
"""Copy the rocprofv3 summaries a `gpurun -- bash tools/profile_round.sh <tag>` call left in gpurun_out/prof_<tag>/ into the
tracked profiles/ directory and derive profiles/pmc_traffic.json (HBM bytes per launch of each workload's dominant
kernel(s)), applying the corrections of MI355X_MICROARCH.md "HBM":  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE counts a wide coalesced streaming read at exactly 1/2 (128-byte requests tallied as 64 B) -> x2;
WRITE_SIZE calibrates 1:1 (k_gen_* writes exactly n*8 bytes and reads 7 812 500 KiB per 1e9 rows in these runs).
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
DOMINANT = {"c2": ["k_filter_aggr_plan"], "c2b": ["k_filter_aggr_plan"], "c5": ["k_filter_aggr_plan"], "x6": ["k_filter_aggr_plan"],  # (the prebuilt k_filter_aggr<...> runs once, at a plan's first occurrence)
            "c3": ["k_plane_", "k_chunk_", "k_scope_sample", "k_part_scope_hist", "k_part_hist", "k_part_scatter", "k_part_aggregate", "k_part_colscan", "k_mark_first", "k_group_emit",
                   "k_slot_gid", "k_bitmap_counts", "k_fill_u64"],
            "c3w": ["k_plane_", "k_chunk_scatter", "k_chunk_offsets", "k_chunk_place", "k_chunk_aggregate", "k_scope_sample", "k_part_scope_hist", "k_chunk_counts", "k_compact_cols", "k_part_hist", "k_part_scatter", "k_part_aggregate", "k_part_colscan", "k_first_translate",
                    "k_mark_first", "k_group_emit", "k_slot_gid", "k_bitmap_counts", "k_fill_u64", "k_sel_bitmap"],
            "q1": ["k_part_scope_hist", "k_filter_aggr_plan", "k_group_few",  # (k_group_dense runs once, at the plan's first occurrence, before the plan kernel exists)
                    "k_rank_emit_small", "k_fill_tables", "k_derive", "k_mark_first", "k_group_emit", "k_slot_gid", "k_bitmap_counts", "k_fill_u64", "k_composite"],
            "q2": ["k_part_scope_hist", "k_group_dense", "k_composite", "k_mark_first", "k_group_emit", "k_slot_gid", "k_bitmap_counts", "k_fill_u64"],
            "k9": ["k_plane_", "k_part_scope_hist", "k_part_hist", "k_part_scatter", "k_part_hash_aggregate", "k_part_colscan", "k_group_hash", "k_distinct_sample", "k_mark_first", "k_group_emit",
                   "k_slot_gid", "k_bitmap_counts", "k_fill_u64"],
            "q7": ["k_row_hash", "k_part_scope_hist", "k_part_hist", "k_part_scatter", "k_part_hash_aggregate", "k_part_colscan", "k_group_hash", "k_join_probe_hash", "k_fill_packed", "k_slot_first", "k_tuple_check", "k_rep_mask", "k_gather_or", "k_gather8", "k_emit_perm", "k_group_emit_by_group", "k_rep_mask", "k_emit_rows", "k_tuple_check", "k_mask_bitmap", "k_emit_ids", "k_chunk_counts", "k_scan", "k_distinct_sample", "k_mark_first",
                   "k_group_emit", "k_slot_gid", "k_bitmap_counts", "k_fill_u64", "k_replace_null"],
            "w2": ["k_where_once", "k_where_sample", "k_sel_bitmap", "k_chunk_counts", "k_emit_ids"], "m2": ["k_cmp_mask"]}


def read_pmc(path):
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        val, n, name = line.rstrip("\n").split("\t")
        out[name] = (float(val), int(n))
    return out


traffic, detail = {}, {}
for w, pats in DOMINANT.items():
    for suffix in ("kernel_stats.csv", "pmc_FETCH_SIZE.txt", "pmc_WRITE_SIZE.txt"):
        f = os.path.join(src, f"{w}_{suffix}")
        if os.path.exists(f):
            shutil.copy(f, os.path.join(dst, f"{tag}_{w}_{suffix}"))
    fe, wr = read_pmc(os.path.join(src, f"{w}_pmc_FETCH_SIZE.txt")), read_pmc(os.path.join(src, f"{w}_pmc_WRITE_SIZE.txt"))
    if not fe:
        continue
    tot, parts = 0.0, {}
    for name, (v, _) in fe.items():
        if any(p in name for p in pats):
            rd = v * 1024 * 2
            wv = wr.get(name, (0.0, 0))[0] * 1024
            parts[name.split("(")[0].replace("void ", "")] = {"read_bytes": rd, "write_bytes": wv}
            tot += rd + wv
    traffic[w] = tot
    detail[w] = parts
tpath = os.path.join(dst, "pmc_traffic.json")
if os.path.exists(tpath):  # a partial round (profile_round.sh <tag> "c3w q7"): the other workloads keep their last figures
    old = json.load(open(tpath))
    old.update(traffic)
    traffic = old
json.dump(traffic, open(tpath, "w"), indent=1)
json.dump({"tag": tag, "corrections": "FETCH_SIZE KiB x1024 x2 (gfx950 streaming-read undercount), WRITE_SIZE KiB x1024", "per_kernel": detail},
          open(os.path.join(dst, f"{tag}_pmc_detail.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
