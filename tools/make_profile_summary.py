"""Builds profiles/<tag>_summary.md from rocprofv3 outputs merged into gpurun_out/ (developer tool).

  python tools/make_profile_summary.py r1 gpurun_out/prof_r1b gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/bench2.json
"""
import collections
import csv
import json
import os
import shutil
import sys


def main():
    tag, stats_dir, fetch_dir, write_dir, bench_json = sys.argv[1:6]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    os.makedirs(prof, exist_ok=True)
    stats = [f for f in os.listdir(stats_dir) if f.endswith("kernel_stats.csv")][0]
    shutil.copy(os.path.join(stats_dir, stats), os.path.join(prof, tag + "_bench_kernel_stats.csv"))
    shutil.copy(bench_json, os.path.join(prof, tag + "_bench_line.json"))
    line = json.loads(open(bench_json).read().strip().splitlines()[-1])
    rows = list(csv.DictReader(open(os.path.join(stats_dir, stats))))
    out = ["# %s profile (MI355X, 1 GPU)" % tag, "",
           "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline`;",
           "raw table: `profiles/%s_bench_kernel_stats.csv`; bench line of the un-profiled run: `profiles/%s_bench_line.json`." % (tag, tag), "",
           "Un-profiled bench: **%.3f G triples/s**, %.3f ms/step, roofline.frac %.3f; phases (HIP events, ms): %s" % (
               line["value"] / 1e9, line["ms_per_step"], line["roofline"]["frac"],
               ", ".join("%s %.3f" % kv for kv in line["roofline"]["phases_ms"].items())), "",
           "| kernel | calls | avg us | % of GPU time |", "|---|---|---|---|"]
    for r in rows[:16]:
        out.append("| `%s` | %s | %.1f | %s |" % (r["Name"][:100].replace("|", "/"), r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    out.append("")

    def agg(d, counter):
        a = collections.defaultdict(list)
        f = [x for x in os.listdir(d) if x.endswith("counter_collection.csv")][0]
        for r in csv.DictReader(open(os.path.join(d, f))):
            if r["Counter_Name"] == counter:
                a[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
        return a
    if os.path.isdir(fetch_dir) and os.path.isdir(write_dir):
        f, w = agg(fetch_dir, "FETCH_SIZE"), agg(write_dir, "WRITE_SIZE")
        out += ["## HBM traffic per launch", "",
                "Separate passes `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` over `python3 bench.py --steps 5 --warmup 2 --no-eval --no-cpu-baseline`.",
                "Both counters are in KiB.  On gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so fetched bytes = 2 x FETCH_SIZE x 1024;",
                "WRITE_SIZE is exact.  Values below are the median over the full-size launches.", "",
                "| kernel | fetched MB | written MB | total MB |", "|---|---|---|---|"]
        tot = 0.0
        for k in f:
            if "seg_" in k or "build_keys" in k or "rocprim" in k:
                big = [x for x in f[k] if x > max(f[k]) / 4] or [0]
                wl = w.get(k, [0]); wbig = [x for x in wl if x > max(wl) / 4] or [0]
                fm = 2 * sorted(big)[len(big) // 2] * 1024 / 1e6
                wm = sorted(wbig)[len(wbig) // 2] * 1024 / 1e6
                n = 1
                if "rocprim" in k:
                    continue
                tot += fm + wm
                out.append("| `%s` | %.0f | %.0f | %.0f |" % (k.replace("|", "/"), fm, wm, fm + wm))
        B = line["config"]["batch_per_gpu"]
        alg = B * line["roofline"]["algorithmic_bytes_per_triple"] / 1e6
        out += ["", "Algorithmic bytes per step: %d x %d = %.0f MB; measured traffic of the row and key kernels: %.0f MB (%.2fx)." % (
            B, line["roofline"]["algorithmic_bytes_per_triple"], alg, tot, tot / alg), ""]
    open(os.path.join(prof, tag + "_summary.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
