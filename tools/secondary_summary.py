#!/usr/bin/env python
"""rocprofv3 kernel trace of tools/secondary.py -> profiles/r03_secondary_kernel_stats.csv + a markdown table with each
kernel's algorithmic bytes per launch and its fraction of the 8 TB/s HBM peak.
usage: python tools/secondary_summary.py TRACE_DIR OUT_PREFIX"""
import collections
import csv
import glob
import sys

HBM = 8.0e12
N, HW, M, K = 131072, 64 * 2048, 100000, 20
# algorithmic bytes per launch of the named kernel at the sizes tools/secondary.py uses (formula in the last column)
BYTES = {
    "k_voxel_hash<float>": (12 * N + 8 * N, "12 N xyz in + 8 N hash out"),
    "k_voxel_hash<double>": (24 * N + 8 * N, "24 N xyz in + 8 N hash out"),
    "k_run_heads": (8 * N + 4 * N, "8 N sorted keys in + 4 N flags out"),
    "k_emit_samples<float>": (16 * N + 20 * 6100, "16 N (key, index) in + 20 V out"),
    "k_distort": (12 * N + 8 * N + 24 * N, "12 N xyz + 8 N timestamps in, 24 N f64 out"),
    "k_voxel_stats": (12 * N + 8 * N + 52 * 6100, "12 N xyz + 8 N voxel id in, 52 V out"),
    "k_normal_map": (12 * HW + 12 * HW, "12 HW vertex map in + 12 HW normal map out (5x5 window reuse in cache)"),
    "k_pm_project": (16 * HW + 8 * HW, "16 HW stored vertices in + 8 HW z-buffer atomics"),
    "k_pm_resolve": (8 * HW + 32 * HW + 32 * HW, "8 HW keys + 32 HW (vertex, normal) gathered in, 32 HW model out"),
    "k_pm_project_targets": (16 * N + 8 * N, "16 N targets in + 8 N z-buffer atomics"),
    "k_reduce_p2p": (16 * N + 4 * N + 16 * N, "16 N targets + 4 N neighbour ids + 16 N matched map points"),
    "k_pm_iterate": (8 * HW + 16 * HW + K * 16 * HW + 16 * HW,
                     "8 HW keys + 16 HW targets + K x 16 HW model vertices + 16 HW winner normals (K = 20 maps; 84 MB of "
                     "model: Infinity-Cache resident, so this is cache bandwidth, not HBM)"),
    "k_pm_store": (24 * HW + 32 * HW, "24 HW (vertex, normal) maps in, 32 HW slots out"),
    "k_pm_assoc": (8 * HW + 12 * HW + K * 32 * HW + 36 * HW, "keys + points + K x 32 HW model + 36 HW rows out"),
    "k_reduce_p2p": (16 * N + 4 * N + 16 * N, "16 N targets + 4 N neighbour ids + 16 N matched map points"),
    "k_search_rows": (16 * N + 16 * N + 4 * N, "16 N targets + 16 N matched map points + 4 N ids out (compulsory)"),
    "k_procrustes_sums": (24 * N, "24 N (two clouds)"),
    "k_procrustes_cov": (24 * N, "24 N (two clouds)"),
    "k_flag_not_nan": (12 * HW + 4 * HW, "12 HW rows in + 4 HW flags out"),
    "k_compact_scatter": (12 * HW + 8 * HW + 12 * 8192, "rows + flags/offsets in, compacted rows out"),
    "k_project": (12 * N + 8 * N, "12 N xyz in + 8 N z-buffer atomics"),
    "k_project_resolve": (8 * HW + 12 * HW + 12 * HW, "8 HW keys + 12 HW gathered points in, 12 HW vertex map out"),
}


def main():
    trace, prefix = sys.argv[1], sys.argv[2]
    f = glob.glob(trace + "/**/*kernel_trace.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("icp::", "")
        acc[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = sorted(((sum(v), k, v) for k, v in acc.items()), reverse=True)
    with open(prefix + "_kernel_stats.csv", "w") as out:
        out.write("kernel,calls,total_us,avg_us,min_us,max_us,algorithmic_bytes_per_launch,GBps,frac_of_8TBps\n")
        md = ["| kernel | calls | avg µs | algorithmic bytes / launch | GB/s | fraction of 8 TB/s | bytes counted |", "|---|---|---|---|---|---|---|"]
        for tot, k, v in rows:
            avg = tot / len(v)
            key = next((b for b in BYTES if k.startswith(b)), None)
            if key:
                by, why = BYTES[key]
                gbps = by / (avg * 1e-6) / 1e9
                out.write(f"\"{k}\",{len(v)},{tot:.1f},{avg:.2f},{min(v):.2f},{max(v):.2f},{by},{gbps:.1f},{gbps * 1e9 / HBM:.4f}\n")
                md.append(f"| `{k[:60]}` | {len(v)} | {avg:.1f} | {by / 1e6:.2f} MB | {gbps:.0f} | {100 * gbps * 1e9 / HBM:.1f} % | {why} |")
            else:
                out.write(f"\"{k}\",{len(v)},{tot:.1f},{avg:.2f},{min(v):.2f},{max(v):.2f},,,\n")
                if tot > 200:
                    md.append(f"| `{k[:60]}` | {len(v)} | {avg:.1f} | | | | (library / copy kernel) |")
    open(prefix + "_summary.md", "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
