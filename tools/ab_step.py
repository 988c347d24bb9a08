"""Developer tool: the BPR step of several builds of the library on ONE box, one after the other (box-to-box differences
are of the order of the differences between kernel variants, so variants are only compared within a run).

    python tools/ab_step.py [--shape c2|large|both] name=path/to/libtfrec_amd.so ...

Each build runs in a child process (TFREC_AMD_LIB names the library); prints the un-pipelined and pipelined step times
and the phase times."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {"c2": ["--U", "1000000", "--I", "200000", "--d", "64"], "large": ["--U", "10000000", "--I", "2000000", "--d", "128"]}


def main():
    args = sys.argv[1:]
    which = "both"
    if args and args[0] == "--shape":
        which, args = args[1], args[2:]
    shapes = ["c2", "large"] if which == "both" else [which]
    for rep in range(2):                  # twice round the list: drift over the run shows as a difference between the rounds
        for spec in args:
            name, path = spec.split("=", 1)
            for sh in shapes:
                env = dict(os.environ, TFREC_AMD_LIB=os.path.abspath(path))
                cmd = [sys.executable, os.path.join(ROOT, "tools", "perf_probe.py"), "--batches", "4194304", "--versioned", "--steps", "20"] + SHAPES[sh]
                out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
                if out.returncode != 0:
                    print("%s %s FAILED: %s" % (name, sh, out.stderr[-500:]), flush=True)
                    return 1
                lines = [ln for ln in out.stdout.splitlines() if "K=1 " in ln or "phases" in ln]
                print("[round %d] %-8s %-5s %s" % (rep, name, sh, " | ".join(ln.strip() for ln in lines)), flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
