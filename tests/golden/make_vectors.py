#!/usr/bin/env python3
"""Generate tests/golden/vectors/transducer.json from the REFERENCE build.

Seeded random input lines, dense in the characters the colour-markup transducer reacts to
(``~``, ``/``, the 21 two-letter codes), are said by one client; what a colour-on and a
colour-off listener receive is recorded.  The unit tests then call the restated transducer
(oracle/nuts_path.c: np_transduce) directly on the same strings -- no sockets -- and require
the same bytes.  Also records, for a few very long lines, the stream for strings that cross
the reference's 1000-byte staging buffer (nuts333.c:1317,1338,1359).
"""
from __future__ import annotations

import json
import random
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))

from nuts333_amd import provision as pv                                  # noqa: E402
from nuts333_amd.talker import REF_BINARY, Talker, free_ports           # noqa: E402
from nuts333_amd.transcript import Session                              # noqa: E402

SEED = 333
CODES = "RS OL UL LI RV FK FR FG FY FB FM FT FW BK BR BG BY BB BM BT BW".split()


def random_line(rng: random.Random, maxlen: int) -> str:
    parts = []
    n = rng.randint(1, maxlen)
    while sum(len(p) for p in parts) < n:
        x = rng.random()
        if x < 0.25:
            parts.append("~" + rng.choice(CODES))
        elif x < 0.35:
            parts.append("/~")
        elif x < 0.45:
            parts.append(rng.choice(["~", "/", "~~", "//", "~F", "~B", "~fr", "~Zz", "/ ~"]))
        elif x < 0.55:
            parts.append(" ")
        else:
            parts.append("".join(rng.choice("abcdefghijklmnopqrstuvwxyzRSFBOLUIVKGYMTW0123456789.,") for _ in range(rng.randint(1, 8))))
    line = "".join(parts)[:maxlen].strip()
    # must reach say(): no leading command character, not a bare dot; keep the verb "say"
    while line and line[0] in ".;!<>-#":
        line = line[1:]
    line = line.rstrip("?!").strip()
    return line or "x"


def main() -> int:
    if not REF_BINARY.exists():
        print("oracle/_ref/nuts333 is missing: run `make -C oracle ref` first", file=sys.stderr)
        return 2
    rng = random.Random(SEED)
    lines = [random_line(rng, 60) for _ in range(250)] + [random_line(rng, 400) for _ in range(40)]
    # long lines around the 1000-byte staging buffer: the say prefix "Bobby says: " is 12 bytes
    for pad in (970, 975, 978, 979, 980, 981, 982, 983, 984, 985, 986, 987):
        lines.append("a" * pad + "~FR" + "b" * 4)
        lines.append("a" * pad + "/~x")
    lines = [l[:985] for l in lines]
    accounts = [pv.Account("Alice", colour=1), pv.Account("Bobby"), pv.Account("Carol")]
    vectors = []
    with tempfile.TemporaryDirectory(prefix="vectors_") as tmp:
        ports = free_ports(3)
        pv.write_tree(tmp, pv.TalkerConfig(mainport=ports[0], wizport=ports[1], linkport=ports[2], max_users=20), accounts)
        with Talker(REF_BINARY, tmp):
            s = Session(ports[0])
            try:
                s.connect("a"); s.login("a", "Alice", colour=True)
                s.connect("b"); s.login("b", "Bobby")
                s.connect("c"); s.login("c", "Carol")
                for line in lines:
                    s.line("b", line)
                    recv = s.steps[-1]["recv"]
                    vectors.append({"line": line, "colour_on": recv["a"], "colour_off": recv["c"], "self": recv["b"]})
            finally:
                s.shutdown()
    out = Path(__file__).resolve().parent / "vectors"
    out.mkdir(exist_ok=True)
    (out / "transducer.json").write_text(json.dumps(
        {"seed": SEED, "speaker": "Bobby", "format": "listener receives transduce('Bobby says: ' + line + '\\n')",
         "vectors": vectors}, indent=0, ensure_ascii=True) + "\n")
    print(f"{len(vectors)} vectors -> {out / 'transducer.json'}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
