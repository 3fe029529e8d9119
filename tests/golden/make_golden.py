#!/usr/bin/env python3
"""Generate tests/golden/*.json by running tests/scenarios.py against the REFERENCE build.

    make -C oracle ref                       # needs /root/reference (never on the GPU box)
    python tests/golden/make_golden.py       # rewrites every fixture
    python tests/golden/make_golden.py markup filters

A fixture is data only: the provisioning (account flags, config switches), the input lines
and, per step, the bytes each client received (latin-1 text; the two nondeterministic
fields masked by nuts333_amd.transcript.normalise).  No reference source text is stored.
Run twice, with the -O2 and the -O0 reference builds, the output must be identical; the
script checks that before writing.
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))

from nuts333_amd.talker import REF_BINARY, REF_BINARY_O0  # noqa: E402
import scenarios                                            # noqa: E402
from scenario_runner import run_scenario                    # noqa: E402


def main(argv: list[str]) -> int:
    names = argv or list(scenarios.SCENARIOS) + list(scenarios.REFERENCE_ONLY)
    if not REF_BINARY.exists():
        print("oracle/_ref/nuts333 is missing: run `make -C oracle ref` first", file=sys.stderr)
        return 2
    out_dir = Path(__file__).resolve().parent
    for name in names:
        a = run_scenario(name, REF_BINARY)
        b = run_scenario(name, REF_BINARY_O0) if REF_BINARY_O0.exists() else a
        if a != b:
            print(f"{name}: -O2 and -O0 reference builds disagree", file=sys.stderr)
            return 1
        c = run_scenario(name, REF_BINARY)
        if a != c:
            print(f"{name}: two runs of the same build disagree (nondeterministic capture)", file=sys.stderr)
            return 1
        path = out_dir / ("reference_only" if name in scenarios.REFERENCE_ONLY else "") / f"{name}.json"
        path.parent.mkdir(exist_ok=True)
        path.write_text(json.dumps(a, indent=1, ensure_ascii=True) + "\n")
        nbytes = sum(len(v) for s in a["steps"] for v in s["recv"].values())
        print(f"{name}: {len(a['steps'])} steps, {nbytes} received bytes -> {path.relative_to(REPO)}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main(sys.argv[1:]))
