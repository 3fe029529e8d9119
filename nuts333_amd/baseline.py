"""Formal CPU baseline: the five BASELINE.json configurations, three repetitions each, median.

    python -m nuts333_amd.baseline [--binary auto|reference|reference_O0|port] [--reps 3]
                                   [--out profiles/baseline_rNN_<host>.json] [--quick]

Reported per configuration: N; input lines/s; delivered lines/s;
delivered == expected (per client); server user/sys CPU per written line; bytes per line; for
the netlink configuration the MSG..EMSG frame counts.  The talker is single-threaded, so
every number is "one core" for the server by construction; the synthetic clients run on the
remaining cores.
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import statistics
import sys
import time
from pathlib import Path

from . import workloads
from .talker import REF_BINARY_O0


def host_info() -> dict:
    model = ""
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"hostname": platform.node(), "kernel": platform.release(), "cpu": model,
            "cpus_available": len(os.sched_getaffinity(0))}


def median_of(runs: list[dict]) -> dict:
    med = lambda f: statistics.median(f(r) for r in runs)
    srv = lambda r: r["servers"][0]
    return {
        "workload": runs[0]["workload"], "reps": len(runs), "clients": runs[0]["clients"],
        "input_lines": runs[0]["input_lines"], "expected_deliveries": runs[0]["expected_deliveries"],
        "all_exact": all(r["exact"] for r in runs),
        "delivered_lines_per_s": round(med(lambda r: r["delivered_lines_per_s"]), 1),
        "delivered_lines_per_s_all": [round(r["delivered_lines_per_s"], 1) for r in runs],
        "input_lines_per_s": round(med(lambda r: r["input_lines_per_s"]), 1),
        "bytes_per_line": round(runs[0]["bytes_per_line"], 3),
        "server_cpu_us_per_written_line": round(med(lambda r: srv(r)["cpu_us_per_written_line"]), 3),
        "server_user_frac": round(med(lambda r: srv(r)["user_frac"] or 0.0), 3),
        "server_busy_frac": round(med(lambda r: srv(r)["busy_frac"]), 3),
        "ack_latency_us_p50": round(med(lambda r: r["ack_latency_us"]["p50"]), 1),
        "ack_latency_us_p99": round(med(lambda r: r["ack_latency_us"]["p99"]), 1),
        "login_s": round(med(lambda r: r["login_s"]), 3),
        "server_rss_peak_kb": max(r.get("server_rss_peak_kb", 0) for r in runs),
        "read_syscalls_per_input_line": round(med(lambda r: srv(r)["read_syscalls_per_input_line"]), 4),
        "write_syscalls_per_written_line": round(med(lambda r: srv(r)["write_syscalls_per_line"]), 4),
        # configuration #5: link frames MEASURED in every repetition (workloads.config5), not modelled
        **({"netlink_writes_t1_to_t2": [r["netlink"]["writes_t1_to_t2"] for r in runs],
            "netlink_writes_t2_to_t1": [r["netlink"]["writes_t2_to_t1"] for r in runs],
            "netlink_expected": {k: runs[0]["netlink"][k] for k in ("expected_act_frames", "expected_msg_frames", "expected_prm_frames")},
            "netlink_exact": all(r["netlink"]["exact"] for r in runs)} if "netlink" in runs[0] else {}),
    }


def plan(quick: bool) -> list[tuple[str, callable]]:
    q = 10 if quick else 1
    return [
        ("config1", lambda b: workloads.config1(lines=10_000 // q, warmup=500, binary=b)),
        ("config2", lambda b: workloads.config2(lines=20_000 // q, warmup=1000, binary=b)),
        ("config2_colour_on", lambda b: workloads.config2(lines=20_000 // q, warmup=1000, colour=1, binary=b)),
        ("config2_all_send", lambda b: workloads.config2(lines=20_000 // q, warmup=1000, all_send=True, binary=b)),
        ("config3", lambda b: workloads.config3(per_client=200 // q, warmup=2, binary=b)),
        ("config3_six_rooms", lambda b: workloads.config3(per_client=200 // q, warmup=2, six_rooms=True, binary=b)),
        ("config4", lambda b: workloads.config4(lines=1000 // q, n=1000 // (4 if quick else 1), warmup=20, binary=b)),
        ("config4_colour_on", lambda b: workloads.config4(lines=1000 // q, n=1000 // (4 if quick else 1), warmup=20, colour=1, binary=b)),
        ("config5", lambda b: workloads.config5(lines=1000 // q, binary=b)),
    ]


def to_markdown(doc: dict) -> str:
    h = doc["host"]
    out = [f"Host: {h['cpu']} ({h['cpus_available']} CPUs available), Linux {h['kernel']}; implementation: "
           f"**{doc['implementation']}** (`{doc['binary']}`); {doc['reps']} repetitions, median.", "",
           "| Config | N | Input lines | Delivered = expected | Delivered lines/s | Input lines/s | Server CPU µs/line (user %) | Busy | B/line | ack p50 / p99 µs |",
           "|---|---|---|---|---|---|---|---|---|---|"]
    for name, m in doc["results"].items():
        out.append(f"| {name} | {m['clients']} | {m['input_lines']} | {m['expected_deliveries']} "
                   f"{'✓' if m['all_exact'] else '✗'} | {m['delivered_lines_per_s']:,.0f} | {m['input_lines_per_s']:,.0f} | "
                   f"{m['server_cpu_us_per_written_line']:.2f} ({100 * m['server_user_frac']:.0f} %) | {m['server_busy_frac']:.2f} | "
                   f"{m['bytes_per_line']:.1f} | {m['ack_latency_us_p50']:.0f} / {m['ack_latency_us_p99']:.0f} |")
    return "\n".join(out) + "\n"


def main(argv: list[str] | None = None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--binary", default="auto", choices=["auto", "reference", "reference_O0", "port", "port_fast"])
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default="")
    ap.add_argument("--quick", action="store_true", help="one tenth of the lines, 250 instead of 1000 clients")
    ap.add_argument("--only", default="", help="comma-separated config names")
    args = ap.parse_args(argv)
    if args.binary == "reference_O0":
        binary, kind = REF_BINARY_O0, "reference (-O0, as the reference's own build script compiles it)"
        if not binary.exists():
            raise SystemExit("oracle/_ref/nuts333_O0 is not built")
    else:
        binary, kind = workloads.pick_binary(args.binary)
    doc = {"host": host_info(), "implementation": kind, "binary": str(binary), "reps": args.reps,
           "quick": args.quick, "started": time.strftime("%Y-%m-%d %H:%M:%S"), "results": {}}
    only = set(filter(None, args.only.split(",")))
    for name, fn in plan(args.quick):
        if only and name not in only:
            continue
        runs = []
        for rep in range(args.reps):
            t = time.time()
            r = fn(binary)
            runs.append(r)
            print(f"[baseline] {name} rep {rep + 1}/{args.reps}: {r['delivered_lines_per_s']:,.0f} delivered/s, "
                  f"exact={r['exact']}, {time.time() - t:.1f}s", file=sys.stderr, flush=True)
        doc["results"][name] = median_of(runs)
    doc["probe_write_67B"] = workloads.probe_write(67, 200_000)
    text = json.dumps(doc, indent=1)
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(text + "\n")
        Path(args.out).with_suffix(".md").write_text(to_markdown(doc))
    print(to_markdown(doc))
    return 0 if all(m["all_exact"] and m.get("netlink_exact", True) for m in doc["results"].values()) else 1


if __name__ == "__main__":
    raise SystemExit(main())
