"""Builds profiles/<tag>_eval_experiments.md from the logs of tools/ab_eval.sh (developer tool): the scoring kernel's
variants measured on one box each, and the per-wave cycle profile of the shipped kernel.

  python tools/eval_ab_table.py r3 gpurun_out/r3_ab_eval.log gpurun_out/r3_ab_eval2.log
"""
import collections
import os
import re
import sys

WHAT = {
    "evbase": "shipped kernel: two tiles of 192 items, one workgroup barrier per tile",
    "ring": "ring of 6 tiles of 64 items guarded by per-slot counters in LDS, a tile staged 3 ahead (no workgroup barrier; a wave may fall 3 tiles behind) -- `-DTF_EVAL_RING=1`",
    "r128": "ring of 3 tiles of 128 items, staged 1 ahead -- `-DTF_EVAL_RING=1 -DTF_EVAL_RING_TI=128 -DTF_EVAL_RING_SLOTS=3 -DTF_EVAL_RING_AHEAD=1`",
    "r64s4": "ring of 4 tiles of 64 items, staged 2 ahead",
    "r64pf2": "ring of 6 tiles of 64 items, staged 2 ahead",
    "af": "all eight A fragments of a block of 32 items requested before the first matrix instruction (one LDS round trip per block instead of four; 24 more registers: the staging registers spill once per tile) -- `-DTF_EVAL_AFRAG_AHEAD=1`",
    "afring": "both of the above",
}


def main():
    tag, logs = sys.argv[1], sys.argv[2:]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = ["# %s: scoring-kernel experiments (MI355X, one GPU)" % tag, "",
           "`tools/ab_eval.sh name=library ...` runs `tools/eval_probe.py` for every build, twice round the list; builds come from",
           "`tools/build_eval_variant.sh NAME -D...`.  bf16, 262,144 users; C5 = 2M items, d=128, k=100; C2 = 200k items, d=64, k=50;",
           "*random* = independent factors, b = 0; *biased* = item biases dominate.  Milliseconds per call of `tfrec_score_topk`",
           "(mean over the runs of a log; boxes differ by several per cent, so compare within a log only).", ""]
    for log in logs:
        t = collections.defaultdict(list)
        prof = {}
        for line in open(log):
            m = re.match(r"\[round \d\] (\S+) (c5|c2) (\S+): (.*)", line)
            if not m:
                continue
            name, shape, model, rest = m.groups()
            for ms in re.findall(r": ([0-9.]+) ms ", rest):
                t[(name, shape, model)].append(float(ms))
            pm = re.search(r"per wave: (.*?) eval", rest)
            if pm and name.endswith("prof"):
                prof[(name, model)] = pm.group(1)
        names = []
        for k in t:
            if k[0] not in names and not k[0].endswith("prof"):
                names.append(k[0])
        out += ["## %s" % os.path.basename(log), "", "| build | what | C5 random | C5 biased | C2 random |", "|---|---|---|---|---|"]
        for n in names:
            cell = lambda sh, mo: ("%.1f" % (sum(t[(n, sh, mo)]) / len(t[(n, sh, mo)]))) if t[(n, sh, mo)] else "-"     # noqa: E731
            out.append("| %s | %s | %s | %s | %s |" % (n, WHAT.get(n, ""), cell("c5", "random"), cell("c5", "biased"), cell("c2", "random")))
        out.append("")
        for (n, mo), p in prof.items():
            out += ["Cycle counts per wave of `%s` (`-DTF_EVAL_PROFILE`: shader-clock reads around the parts of the scan; the reads"
                    " themselves slow the kernel by about a tenth), C5 %s:" % (n, mo), "", "    " + p, ""]
    base = os.path.join(root, "gpurun_out", tag + "_evalprof.log")
    if os.path.exists(base):
        lines = [ln.strip() for ln in open(base) if "per wave" in ln or ln.startswith("eval")]
        out += ["## Shipped kernel, `-DTF_EVAL_PROFILE` build (`%s`)" % os.path.basename(base), "", "C5 random, two calls:", ""] + ["    " + ln for ln in lines] + [""]
    out += ["Reading: none of the variants is kept.  Letting waves drift apart (ring) does not shorten the time a wave spends between",
            "the end of a tile and the start of the next (`store+barrier`, 28 % of a wave's cycles in the shipped kernel, 21-26 % in the",
            "ring builds), so that time is not one wave waiting for a slower one; requesting the A fragments earlier makes the kernel",
            "slower.  What the numbers are consistent with: a wave tile of 32 users x 32 items reads one kilobyte of A fragment from LDS",
            "per matrix instruction, which for four SIMDs is exactly the 128 bytes per clock the LDS delivers, so the LDS, not the matrix",
            "pipe, paces the scan; the way out is a larger wave tile (two user sets per wave), which at d=128 needs more than 128",
            "registers per lane and therefore half the resident waves (measured in round 2: +4 %)."]
    open(os.path.join(root, "profiles", tag + "_eval_experiments.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
