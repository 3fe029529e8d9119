#!/usr/bin/env python3
"""Print the BASELINE.md rows for configurations #1 / #5 run on the reference's shipped files (container only).

    python tools/shipped_rows.py [--reps 3]

Uses tests/shipped_tree.py: a temporary directory is populated from /root/reference at run time, nothing is kept.
"""
from __future__ import annotations

import argparse
import statistics
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
import shipped_tree as st                      # noqa: E402
from nuts333_amd import workloads              # noqa: E402
from nuts333_amd.talker import REF_BINARY      # noqa: E402


def row(label: str, runs: list[dict]) -> str:
    med = lambda f: statistics.median(f(r) for r in runs)
    return (f"| {label} | {runs[0]['clients']} | {runs[0]['input_lines']} | {all(r['exact'] for r in runs)} | "
            f"{med(lambda r: r['input_lines_per_s']):,.0f} | {med(lambda r: r['delivered_lines_per_s']):,.0f} | "
            f"{med(lambda r: r['servers'][0]['cpu_ns'] / 1e3 / r['input_lines']):.2f} | "
            f"{med(lambda r: r['servers'][0]['cpu_us_per_written_line']):.2f} | {runs[0]['bytes_per_line']:.1f} |")


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    if not st.available():
        raise SystemExit("needs /root/reference and oracle/_ref/nuts333")
    print("| Run | N | Input lines | exact | Input lines/s | Delivered lines/s | Server CPU µs / input line | µs / written line | B/line |")
    print("|---|---|---|---|---|---|---|---|---|")
    a, b, c = [], [], []
    for _ in range(args.reps):                 # interleaved: a shared VM drifts by 10 % over a minute
        a.append(st.config1_shipped())
        b.append(workloads.config1(lines=10_000, warmup=500, prompt=1, binary=REF_BINARY))
        c.append(workloads.config1(lines=10_000, warmup=500, binary=REF_BINARY))
    print(row("#1 shipped `datafiles/config`, `fred`/`test` (prompt on)", a))
    print(row("#1 generated tree, account flagged like `Fred.D` (prompt on)", b))
    print(row("#1 generated tree, prompt off (the formal baseline's #1)", c))
    print(row("#5 shipped `config` + `config2` (line 11 fixed), Fred shouts across the link",
              [st.config5_shipped(lines=200) for _ in range(args.reps)]))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
