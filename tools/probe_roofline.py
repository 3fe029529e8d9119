#!/usr/bin/env python3
"""Stand-alone run of the host system-call probe behind bench.py's ``roofline`` (loadgen --probe-line).

    python tools/probe_roofline.py [--reps 3] [--out profiles/probe_rNN_<host>.json]

For the two fan-out shapes of the BASELINE configurations -- 10 clients / `say` (9 recipients, 67-byte lines) and
1000 clients / `.shout` (999 recipients, 69-byte lines) -- it runs the three legs ``--reps`` times each:

  write_only_closed   K+1 write(2) per round, closed loop                       (round 1's ceiling)
  full_closed         1 select(FD_SETSIZE) + 1 read + K+1 writes, closed loop   -> peak from CPU time
  full_open           the same, open loop (select never sleeps)                 -> peak DEMONSTRATED on the wall clock

and prints the medians.  As in bench.py, ``peak`` is the highest wall-clock rate any repetition of a full leg demonstrated;
the closed loop's CPU-time extrapolation is ``peak_extrapolated`` (round 2 quoted the lower of the two medians).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from nuts333_amd import workloads  # noqa: E402
from nuts333_amd.baseline import host_info  # noqa: E402

SHAPES = {"config2 (10 clients, say)": (67, 9), "config4 (1000 clients, .shout)": (69, 999)}
LEGS = {"write_only_closed": (0, 0), "full_closed": (1, 0), "full_open": (1, 1)}
KEYS = ("cpu_ns_per_line", "wall_ns_per_line", "cpu_ns_select_read_per_line", "cpu_ns_per_write",
        "written_lines_per_s_cpu", "written_lines_per_s_wall")


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    workloads.build_loadgen()
    cpus = workloads.host_cpus()          # quiet-core placement, as in bench.py (nuts333_amd/placement.py)
    readers = max(1, min(4, len(cpus) - 1))
    place = [str(cpus[0]), ",".join(map(str, cpus[1:1 + readers]))] if len(cpus) >= 2 else []
    from nuts333_amd import placement
    doc = {"host": host_info(), "reps": args.reps, "readers": readers, "placement": place, "placement_policy": placement.describe(), "shapes": {}}
    for shape, (size, k) in SHAPES.items():
        rounds = max(100, 300_000 // (k + 1))
        legs = {}
        for leg, (selread, open_loop) in LEGS.items():
            runs = []
            for _ in range(args.reps):
                cmd = [str(workloads.LOADGEN_BIN), "--probe-line", str(size), str(k), str(rounds), str(selread), str(open_loop),
                       str(readers)] + place
                runs.append(json.loads(subprocess.run(cmd, check=True, stdout=subprocess.PIPE, timeout=120).stdout))
            assert all(r["bytes_ok"] for r in runs)
            legs[leg] = {**{key: round(statistics.median(r[key] for r in runs), 1) for key in KEYS},
                         "written_lines_per_s_wall_all": [r["written_lines_per_s_wall"] for r in runs],
                         "written_lines_per_s_cpu_all": [r["written_lines_per_s_cpu"] for r in runs]}
        peak_cpu, peak_demo = legs["full_closed"]["written_lines_per_s_cpu"], legs["full_open"]["written_lines_per_s_wall"]
        peak = max(legs["full_open"]["written_lines_per_s_wall_all"] + legs["full_closed"]["written_lines_per_s_wall_all"])
        doc["shapes"][shape] = {"bytes": size, "recipients": k, "writes_per_input_line": k + 1, "rounds": rounds, "legs": legs,
                                "peak": peak, "peak_extrapolated": peak_cpu, "peak_demonstrated_median_open_loop": peak_demo,
                                "peak_closed_loop_cpu_time": peak_cpu, "peak_open_loop_wall_demonstrated": peak_demo,
                                "peak_write_only": legs["write_only_closed"]["written_lines_per_s_cpu"]}
        print(f"{shape}: peak {peak:,.0f} lines/s demonstrated (open-loop median {peak_demo:,.0f}; CPU-time extrapolation {peak_cpu:,.0f}; "
              f"write-only {legs['write_only_closed']['written_lines_per_s_cpu']:,.0f}); select+read "
              f"{legs['full_closed']['cpu_ns_select_read_per_line'] / 1e3:.2f} us per input line, "
              f"{legs['full_closed']['cpu_ns_per_write'] / 1e3:.3f} us per write")
    if args.out:
        Path(args.out).write_text(json.dumps(doc, indent=1) + "\n")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
