"""Builds profiles/<tag>_large_summary.md from the three rocprofv3 passes of tools/collect_profiles.sh over the BPR step at
the large configuration's table shape (developer tool).

  python tools/large_profile_summary.py r3 gpurun_out/r3_large_stats gpurun_out/r3_large_fetch gpurun_out/r3_large_write
"""
import collections
import csv
import os
import sys

from make_profile_summary import find, short, GATHER_KERNELS

B, D = 4194304, 128
ALG = 24 * D + 28


def main():
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(find(stats_dir, "kernel_trace.csv"))):
        k = short(r["Kernel_Name"])
        if k:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)

    def per_kernel(d, counter):
        a = collections.defaultdict(list)
        for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
            if r["Counter_Name"] == counter:
                k = short(r["Kernel_Name"])
                if k:
                    a[k].append(float(r["Counter_Value"]))
        return a
    f, w = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    steps = max(len(dur.get("seg_tile_kernel<BprUserPol>", [])), 1)
    psteps = max(len(f.get("seg_tile_kernel<BprUserPol>", [])), 1)
    out = ["# %s: BPR step at the large configuration's table shape (MI355X, 1 GPU)" % tag, "",
           "`tools/perf_probe.py --U 10000000 --I 2000000 --d 128 --batches 4194304 --versioned --plain-only` (un-pipelined",
           "`tfrec_bpr_sgd_step_v`, 10M users x 2M items, d=128, 4M triples per step, positives Zipf, %d steps traced) under" % steps,
           "`rocprofv3 --kernel-trace --stats`, `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, one pass each (`tools/collect_profiles.sh`).",
           "FETCH_SIZE counts 64-byte requests: bytes = 2 x FETCH_SIZE x 1024 for streaming kernels (the guide's gfx950 correction);",
           "the row kernels gather 512-byte rows here, whole 128-byte lines, so the same factor is used for them (at d=64, where a row",
           "is two lines, 1.333 was calibrated: both figures are given).  WRITE_SIZE is exact.", "",
           "| kernel | launches/step | us/step | fetched MB/step (x2) | fetched MB/step (x1.333) | written MB/step | TB/s (x2 + written) | of 6.29 TB/s |",
           "|---|---|---|---|---|---|---|---|"]
    tot_us = tot_f = tot_w = 0.0
    for k in sorted(set(dur) | set(f) | set(w)):
        us = sum(dur.get(k, [0])) / steps
        fk = sum(f.get(k, [0])) * 1024 / 1e6 / psteps
        wk = sum(w.get(k, [0])) * 1024 / 1e6 / psteps
        tbs = (2 * fk + wk) / 1e6 / (us * 1e-6) if us else 0.0
        tot_us += us; tot_f += 2 * fk; tot_w += wk
        out.append("| `%s` | %.1f | %.0f | %.0f | %.0f | %.0f | %.2f | %.2f |" % (k, len(dur.get(k, [])) / steps, us, 2 * fk,
                                                                              1.3333 * fk if k.startswith(GATHER_KERNELS) else 2 * fk, wk, tbs, tbs / 6.29))
    alg = B * ALG / 1e6
    out += ["", "Per step: kernels %.0f us, fetched %.0f MB + written %.0f MB = %.0f MB moved; algorithmic %d x %d = %.0f MB: the step moves %.2fx"
            % (tot_us, tot_f, tot_w, tot_f + tot_w, B, ALG, alg, (tot_f + tot_w) / alg),
            "its algorithmic bytes (each triple's user row is read again, twice, by the item half: 1024 bytes per triple that no staging or",
            "versioning removes, because the two occurrences of a triple lie in different item segments).",
            "Moved bytes / kernel time = %.2f TB/s = %.2f of the 6.29 TB/s a streaming copy reaches; at that copy rate the moved bytes alone"
            % ((tot_f + tot_w) / 1e6 / (tot_us * 1e-6), (tot_f + tot_w) / 1e6 / (tot_us * 1e-6) / 6.29),
            "would take %.2f ms, i.e. roofline.frac %.2f of 8 TB/s is the ceiling of this algorithm at this shape." % (
                (tot_f + tot_w) / 1e6 / 6.29 * 1e3, alg / 1e6 / ((tot_f + tot_w) / 1e6 / 6.29) / 8.0), ""]
    import json
    json.dump({"workload_batch": B, "factors": D, "users": 10_000_000, "items": 2_000_000, "bytes_per_step": (tot_f + tot_w) * 1e6,
               "fetched_bytes_per_step": tot_f * 1e6, "written_bytes_per_step": tot_w * 1e6, "kernel_us_per_step": tot_us,
               "source": "profiles/%s_large_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, bytes = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024)" % tag},
              open(os.path.join(root, "profiles", "pmc_traffic_large.json"), "w"), indent=1)
    open(os.path.join(root, "profiles", tag + "_large_summary.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
