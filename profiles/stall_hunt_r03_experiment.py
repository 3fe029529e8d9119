#!/usr/bin/env python3
"""Box experiment (round 3): try to catch round 2's stalled restatement leg again, this time with counters.

Round 2's driver line had ONE leg at equal CPU per line and 5.3x the wall clock (the restatement on config #4, run
straight after the three probe legs, talker pinned to CPU 0, receivers to CPUs 1-4, host load average 33) and nothing in
the record to say why.  This repeats exactly that sequence -- three probe legs, then the restatement on 2000 `.shout`
lines to 999 recipients -- REPS times, alternating the old placement (``first``: CPU 0 + 1-4) and the round-3
placement (``quiet``: the quietest L3 group), and prints for every run where the talker's wall clock went.

    python profiles/stall_hunt_r03_experiment.py [REPS] > gpurun_out/stall_hunt.log
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from nuts333_amd import placement, workloads  # noqa: E402
from nuts333_amd.talker import PORT_BINARY  # noqa: E402


def main() -> int:
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    workloads.build_loadgen()
    # interleaved, so that both placements meet the same minutes of the (shared, drifting) host
    for k in range(reps):
        for policy in ("first", "quiet"):
            os.environ["NUTS_BENCH_CPUS"] = policy
            cpus = placement.ordered_cpus(refresh=True)
            place = [str(cpus[0]), ",".join(map(str, cpus[1:5]))]
            for selread, open_loop in ((0, 0), (1, 0), (1, 1)):
                subprocess.run([str(workloads.LOADGEN_BIN), "--probe-line", "69", "999", "300", str(selread), str(open_loop), "4"] + place,
                               check=True, stdout=subprocess.DEVNULL, timeout=120)
            la = Path("/proc/loadavg").read_text().split()[0]
            th0 = workloads.cgroup_throttled()
            res = workloads.config4(lines=2000, warmup=500, binary=PORT_BINARY)
            th1 = workloads.cgroup_throttled()
            d = workloads.leg_diagnostics(res)
            why = workloads.attribute_stall(res, throttled_periods=(th1[0] - th0[0]) if th0 and th1 else None)
            print(json.dumps({"policy": policy, "run": k, "loadavg": float(la), "delivered_lines_per_s": round(res["delivered_lines_per_s"]),
                              "exact": res["exact"], **d, "attribution": why}), flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
