"""Builds profiles/<tag>_summary.md and profiles/pmc_traffic.json from rocprofv3 outputs merged into
gpurun_out/ (developer tool).

  python tools/make_profile_summary.py r1 gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/bench.json

  prof_stats: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline
  pmc_fetch / pmc_write: rocprofv3 --pmc FETCH_SIZE (WRITE_SIZE) --output-format csv -- python3 bench.py --steps 5 --warmup 2
                         --no-eval --no-cpu-baseline --no-other   (one pass each)
  bench.json: the JSON line of an un-profiled default run
  optional, after those: the three counter passes of tools/collect_profiles.sh over the scoring kernel
  (tools/eval_probe.py at the 2M-item, d=128, k=100 shape): eval1 (SQ busy / wave cycles), eval2 (GRBM_GUI_ACTIVE),
  eval3 (instruction and LDS counters)
"""
import collections
import csv
import json
import os
import shutil
import sys

STEP_KERNELS = ("seg_tile_kernel", "seg_fixup_kernel", "utable_flip_kernel", "sort_hist_kernel", "sort_scan_kernel", "sort_scatter_kernel",
                "loss_finalize")
# FETCH_SIZE (KiB) counts memory-side read requests at 64 bytes each.  Streaming reads issue one 128-byte
# request per line (bytes = 2 x FETCH_SIZE x 1024, the microarchitecture guide's correction); gathers of
# 256-byte rows were calibrated with tools/pmc_calibrate.py at 1.5 requests per 128 bytes
# (bytes = 1.333 x FETCH_SIZE x 1024).  WRITE_SIZE (KiB) is exact.
GATHER_KERNELS = ("seg_tile_kernel", "seg_fixup_kernel")


def find(d, suffix):
    for r, _, fs in os.walk(d):
        for f in fs:
            if f.endswith(suffix):
                return os.path.join(r, f)
    raise SystemExit("no *%s under %s" % (suffix, d))


def short(name):
    for k in STEP_KERNELS:
        if k in name:
            pol = ""
            for p in ("BprUserPol", "BprItemPol", "RowsApplyPol"):
                if p in name:
                    pol = "<" + p + ">"
            return k + pol
    return None


def eval_section(d1, d2, d3):
    """Matrix-pipe utilisation of score_topk_kernel from the SQ counters (sums over all dispatches of the passes)."""
    names = set()

    def sums(d):
        a, dur, seen = collections.defaultdict(float), 0, set()
        for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
            if "score_topk" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]:
                names.add(r["Kernel_Name"].replace("void ", "").split("(")[0])
                a[r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    dur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        return a, dur / 1e9, len(seen)
    a1, t1, n1 = sums(d1)
    a2, t2, n2 = sums(d2)
    a3, t3, n3 = sums(d3)
    n_simd, n_cu, n_xcd = 1024, 256, 8
    cyc = a2["GRBM_GUI_ACTIVE"] / n_xcd                      # shader cycles over the pass's dispatches (the counter sums the XCDs)
    clock = cyc / t2 / 1e9
    busy = a1["SQ_VALU_MFMA_BUSY_CYCLES"] / (n_simd * cyc * t1 / t2)
    users, items, d = 262144, 2_000_000, 128
    flops = 2.0 * users * items * d * n1
    out = ["## Scoring kernel: matrix-pipe utilisation (PMC)", "",
           "`tools/eval_probe.py --users 262144 --I 2000000 --d 128 --k 100 --model random --mode bf16` (independent factors) under",
           "`rocprofv3 --pmc` in three passes (`tools/collect_profiles.sh`); sums over the %d dispatches of the bf16 d=128 scoring kernel (`%s`) in a pass." % (n1, ", ".join(sorted(names))), "",
           "| quantity | value |", "|---|---|",
           "| kernel time per dispatch | %.1f ms |" % (t1 / n1 * 1e3),
           "| clock held (GRBM_GUI_ACTIVE / 8 XCDs / time) | %.2f GHz |" % clock,
           "| SQ_INSTS_MFMA per dispatch | %.4g (expected users/32 x items/32 x d/16 = %.4g) |" % (a3["SQ_INSTS_MFMA"] / n3, users / 32 * items / 32 * d / 16),
           "| SQ_VALU_MFMA_BUSY_CYCLES per dispatch | %.4g (= 32 cycles per `v_mfma_f32_32x32x16_bf16`) |" % (a1["SQ_VALU_MFMA_BUSY_CYCLES"] / n1),
           "| **matrix-pipe busy = MFMA busy cycles / (1024 SIMDs x elapsed shader cycles)** | **%.1f %%** |" % (100 * busy),
           "| achieved | %.0f TFLOP/s = %.1f %% of the 2.5 PFLOP/s nominal peak (2.4 GHz), %.1f %% of the peak at the clock held |" % (
               flops / t1 / 1e12, 100 * flops / t1 / 2.5e15, 100 * flops / t1 / (2.5e15 * clock / 2.4)),
           "| CU busy (SQ_BUSY_CU_CYCLES / 256 CUs / cycles) | %.0f %% |" % (100 * a1["SQ_BUSY_CU_CYCLES"] / (n_cu * cyc * t1 / t2)),
           "| LDS array active (SQ_LDS_IDX_ACTIVE / 256 CUs / cycles) | %.0f %%, bank conflicts %.3g |" % (
               100 * a3["SQ_LDS_IDX_ACTIVE"] / (n_cu * cyc * t3 / t2), a3["SQ_LDS_BANK_CONFLICT"]),
           "| wave cycles: waiting (SQ_WAIT_ANY) / issue-stalled (SQ_WAIT_INST_ANY) | %.0f %% / %.0f %% of SQ_WAVE_CYCLES |" % (
               100 * a1["SQ_WAIT_ANY"] / a1["SQ_WAVE_CYCLES"], 100 * a1["SQ_WAIT_INST_ANY"] / a1["SQ_WAVE_CYCLES"]),
           "| vector instructions per matrix instruction | %.1f |" % (a1["SQ_INSTS_VALU"] / a3["SQ_INSTS_MFMA"] * n3 / n1), ""]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    json.dump({"kernel": ", ".join(sorted(names)), "workload": "262144 users x 2M items, d=128, k=100, independent factors",
               "mfma_busy_frac": busy, "clock_ghz": clock, "tflops": flops / t1 / 1e12,
               "source": "rocprofv3 --pmc passes of tools/collect_profiles.sh"},
              open(os.path.join(root, "profiles", "pmc_mfma.json"), "w"), indent=1)
    return out


def main():
    tag, stats_dir, fetch_dir, write_dir, bench_json = sys.argv[1:6]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    os.makedirs(prof, exist_ok=True)
    stats = find(stats_dir, "kernel_stats.csv")
    shutil.copy(stats, os.path.join(prof, tag + "_bench_kernel_stats.csv"))
    line = json.loads(open(bench_json).read().strip().splitlines()[-1])
    json.dump(line, open(os.path.join(prof, tag + "_bench_line.json"), "w"), indent=1)
    rows = list(csv.DictReader(open(stats)))
    rf = line["roofline"]
    out = ["# %s profile (MI355X, 1 GPU)" % tag, "",
           "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline`;",
           "raw table: `profiles/%s_bench_kernel_stats.csv`; bench line of the un-profiled run: `profiles/%s_bench_line.json`." % (tag, tag), "",
           "Un-profiled bench: **%.3f G triples/s**, %.3f ms/step, roofline.frac %.3f; phases of an un-pipelined step (HIP events, ms): %s" % (
               line["value"] / 1e9, line["ms_per_step"], rf["frac"], ", ".join("%s %.3f" % kv for kv in rf["phases_ms"].items())), "",
           "| kernel | calls | avg us | % of GPU time |", "|---|---|---|---|"]
    for r in rows[:18]:
        out.append("| `%s` | %s | %.1f | %s |" % (r["Name"][:110].replace("|", "/"), r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    out.append("")

    def per_kernel(d, counter):
        a = collections.defaultdict(list)
        for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
            if r["Counter_Name"] == counter:
                k = short(r["Kernel_Name"])
                if k:
                    a[k].append(float(r["Counter_Value"]))
        return a
    f, w = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    steps = max(len(f.get("seg_tile_kernel<BprUserPol>", [])), 1)
    out += ["## HBM traffic per step (PMC)", "",
            "Separate passes `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` over `python3 bench.py --steps 5 --warmup 2 --no-eval",
            "--no-cpu-baseline --no-other` (%d steps in the pass).  Both counters are KiB.  FETCH_SIZE tallies memory-side read requests at" % steps,
            "64 bytes: streaming kernels issue one request per 128-byte line (bytes = 2 x FETCH_SIZE x 1024, the guide's gfx950 correction);",
            "gathers of 256-byte rows issue 1.5 requests per 128 bytes (calibrated with `tools/pmc_calibrate.py`: 2^21 random 256-byte rows",
            "= 537 MB read, FETCH_SIZE 393,119 KiB), so the row kernels use bytes = 1.333 x FETCH_SIZE x 1024.  WRITE_SIZE is exact.",
            "Per-step figure = sum over all launches in the pass / steps.", "",
            "| kernel | launches/step | fetched MB/step (guide x2) | fetched MB/step (calibrated) | written MB/step |", "|---|---|---|---|---|"]
    tot_guide = tot_cal = tot_w = 0.0
    moved = {}            # calibrated MB per step, by kernel
    for k in sorted(set(f) | set(w)):
        fk = sum(f.get(k, [0])) * 1024 / 1e6 / steps
        wk = sum(w.get(k, [0])) * 1024 / 1e6 / steps
        cal = fk * (1.3333 if k.startswith(GATHER_KERNELS) else 2.0)
        tot_guide += 2 * fk; tot_cal += cal; tot_w += wk
        moved[k] = cal + wk
        out.append("| `%s` | %.1f | %.0f | %.0f | %.0f |" % (k, len(f.get(k, [])) / steps, 2 * fk, cal, wk))
    B = line["config"]["batch_per_gpu"]
    alg = B * rf["algorithmic_bytes_per_triple"] / 1e6
    out += ["", "Per step: fetched %.0f MB (guide x2) / %.0f MB (calibrated), written %.0f MB; total %.0f MB (calibrated)." % (
                tot_guide, tot_cal, tot_w, tot_cal + tot_w),
            "Algorithmic bytes per step: %d x %d = %.0f MB, so measured traffic is %.2fx algorithmic (calibrated; %.2fx with the guide's x2)." % (
                B, rf["algorithmic_bytes_per_triple"], alg, (tot_cal + tot_w) / alg, (tot_guide + tot_w) / alg), ""]
    # measured traffic of a phase over its duration in the un-pipelined step, against the streaming-copy rate of the chip
    ph = rf["phases_ms"]
    groups = {"index": ("sort_",), "user_tile": ("seg_tile_kernel<BprUserPol>",), "user_fixup": ("seg_fixup_kernel<BprUserPol>",),
              "item_tile": ("seg_tile_kernel<BprItemPol>",), "item_fixup": ("seg_fixup_kernel<BprItemPol>",)}
    out += ["Rate of each phase = its measured traffic / its duration in an un-pipelined step (HIP events of the bench line), against the",
            "6.29 TB/s a streaming copy reaches on this chip:", "", "| phase | MB/step | ms | TB/s | of 6.29 TB/s |", "|---|---|---|---|---|"]
    for name, keys in groups.items():
        mb = sum(v for k, v in moved.items() if k.startswith(keys))
        if name in ph and ph[name] > 0:
            tbs = mb / 1e6 / (ph[name] * 1e-3)
            out.append("| %s | %.0f | %.3f | %.2f | %.2f |" % (name, mb, ph[name], tbs, tbs / 6.29))
    out.append("")
    if len(sys.argv) >= 9:
        out += eval_section(sys.argv[6], sys.argv[7], sys.argv[8])
    json.dump({"workload_batch": B, "factors": line["config"]["factors"],
               "bytes_per_step": (tot_cal + tot_w) * 1e6, "bytes_per_step_guide_x2": (tot_guide + tot_w) * 1e6,
               "fetched_bytes_per_step": tot_cal * 1e6, "written_bytes_per_step": tot_w * 1e6,
               "source": "profiles/%s_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)" % tag},
              open(os.path.join(prof, "pmc_traffic.json"), "w"), indent=1)
    open(os.path.join(prof, tag + "_summary.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
