#!/usr/bin/env python3
"""One parametrised sweep over bench.py (replaces the one-off shell loops of earlier rounds).

Runs `bench.py` once per point of a cartesian product of flag values and environment values, each in a process of
its own under a time limit, stops at the first failure, and prints one line per point with the fields asked for.

    python scripts/sweep.py --only angular --vary log2-ops=22,24 --vary variant=0,23,27 --vary wgs-per-cu=0,4,99
    python scripts/sweep.py --only box --vary layout=tiled,plain --vary log2-ops=18,20,22,24,25,26 --steps 200 --warmup 100
    python scripts/sweep.py --only stream --vary stream-mode=1,2,3 --fields value,ms_per_step,roofline.frac
    python scripts/sweep.py --only euclid --env RT_DP_VEC_PASSES=4,8 --vary log2-ops=18,20,22

Named presets reproduce the tables kept under profiles/ (`--preset size_table`, `beat_shapes`, `beat_caps`, `pack`).
Run it on the GPU box from the repository root; write the output under gpurun_out/ and copy what is to be kept to profiles/.
"""
import argparse
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRESETS = {
    "size_table": [["--only", w, "--vary", "layout=" + l, "--vary", "log2-ops=16,18,20,22,24,25,26", "--steps", "200", "--warmup", "100"]
                   for w, l in (("box", "tiled"), ("box", "plain"), ("tri", "tiled"), ("tri", "plain"), ("euclid", "plain"), ("angular", "plain"))],
    "beat_shapes": [["--only", w, "--vary", "variant=0,23,27,28", "--vary", "log2-ops=18,20,22,24,26", "--steps", "200", "--warmup", "50"] for w in ("euclid", "angular")],
    "beat_caps": [["--only", w, "--vary", "variant=23,27", "--vary", "wgs-per-cu=3,4,5,6,7,99", "--steps", "100", "--warmup", "30"] for w in ("euclid", "angular")],
    "pack": [["--only", "pack", "--vary", "log2-ops=20,22,24,26", "--steps", "20", "--warmup", "3"]],
}


def get(d, path):
    for k in path.split("."):
        d = d.get(k) if isinstance(d, dict) else None
    return d


def run_sweep(argv):
    p = argparse.ArgumentParser()
    p.add_argument("--only", required=True)
    p.add_argument("--vary", action="append", default=[], help="flag=v1,v2,... (bench.py flag without the leading dashes)")
    p.add_argument("--env", action="append", default=[], help="NAME=v1,v2,... (environment variable of the library or of bench.py)")
    p.add_argument("--steps", default="100")
    p.add_argument("--warmup", default="30")
    p.add_argument("--fields", default="roofline.frac,ms_per_step")
    p.add_argument("--limit", type=int, default=120, help="seconds per point")
    p.add_argument("extra", nargs="*", help="further bench.py arguments, after --")
    a = p.parse_args(argv)
    axes = [("flag", kv.split("=", 1)[0], kv.split("=", 1)[1].split(",")) for kv in a.vary] + \
           [("env", kv.split("=", 1)[0], kv.split("=", 1)[1].split(",")) for kv in a.env]
    for point in itertools.product(*[vals for _, _, vals in axes]):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--only", a.only, "--steps", a.steps, "--warmup", a.warmup, "--no-cpu-baseline"] + a.extra
        env = dict(os.environ)
        label = []
        for (kind, name, _), v in zip(axes, point):
            if kind == "flag":
                cmd += ["--" + name, v]
            else:
                env[name] = v
            label.append("%s=%s" % (name, v))
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=a.limit, cwd=ROOT)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            print("%s %s FAILED rc=%d\n%s" % (a.only, " ".join(label), r.returncode, r.stderr[-1500:]), flush=True)
            return 1          # a failing GPU step ends the sweep: nothing further is started
        d = json.loads(lines[-1])
        print("%s %s %s" % (a.only, " ".join(label), " ".join("%s=%s" % (f, get(d, f)) for f in a.fields.split(","))), flush=True)
    return 0


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--preset":
        for argv in PRESETS[sys.argv[2]]:
            rc = run_sweep(argv + sys.argv[3:])
            if rc:
                return rc
        return 0
    return run_sweep(sys.argv[1:])


if __name__ == "__main__":
    sys.exit(main())
