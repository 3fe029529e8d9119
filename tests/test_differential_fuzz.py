"""Differential fuzzing: seeded random sessions run against the reference and the restatement.

The hand-written scenarios in tests/scenarios.py pin what we thought of; this pins what we did
not.  For each seed a random but reproducible script is generated from the commands on (and next
to) the hot path -- speech in all its forms with markup-dense text, toggles, movement, review
buffers, clones, level-scoped shouts -- and executed, one step at a time, against both talkers.
Every byte every client receives at every step must agree -- and so must the sequence of write(2)
calls each talker issued to produce them (sizes and order, logged by an LD_PRELOAD shim).  Needs the
reference build; runs wherever oracle/_ref/ is present.
"""
from __future__ import annotations

import random

import pytest

from nuts333_amd import provision as pv
import scenarios
from scenario_runner import run_scenario

NAMES = ["Alice", "Bobby", "Carol", "Dave"]          # USER, USER, WIZ, ARCH
KEYS = ["a", "b", "c", "d"]
LEVELS = [1, 1, 2, 3]
ROOMS = ["drive", "hallway", "corridor", "lounge", "wizroom", "nowhere", "ha", "dr"]
CODES = "RS OL UL LI RV FK FR FG FY FB FM FT FW BK BR BG BY BB BM BT BW".split()
WORDS = ["hello", "there", "nuts", "talker", "lines", "x", "yy", "zzz", "shit", "Scunthorpe", "42", "ok"]


def random_text(rng: random.Random) -> str:
    parts = []
    for _ in range(rng.randint(1, 8)):
        x = rng.random()
        if x < 0.18:
            parts.append("~" + rng.choice(CODES))
        elif x < 0.24:
            parts.append("/~" + rng.choice(["", "FR", "x"]))
        elif x < 0.30:
            parts.append(rng.choice(["~", "~~", "~F", "/", "~zz"]))
        else:
            parts.append(rng.choice(WORDS))
    x = rng.random()
    if x < 0.04:
        parts.insert(rng.randrange(len(parts) + 1), "w" * rng.choice([39, 40, 41, 85]))    # wordfind's 39-char slots
    elif x < 0.08:
        parts += [f"w{j}" for j in range(rng.choice([9, 10, 11, 14]))]                     # ... and its ten of them
    elif x < 0.10:
        parts.append("long" * rng.choice([50, 120, 230]))                                  # review ring cut, write chunking
    text = " ".join(parts)
    text += rng.choice(["", "", "", "?", "!"])
    return text


def make_script(seed: int):
    rng = random.Random(seed)
    colour0 = [rng.random() < 0.5 for _ in KEYS]
    accounts = [pv.Account(n, level=l, colour=int(c), desc=f"is {n.lower()}") for n, l, c in zip(NAMES, LEVELS, colour0)]
    steps = []
    colour = list(colour0)
    for _ in range(90):
        i = rng.randrange(4)
        k = KEYS[i]
        r = rng.random()
        flags = {}
        target = rng.choice(NAMES + ["Nobody", "al", "ob"]).lower()
        if r < 0.22:
            text = random_text(rng)
            line = text if text[0] not in ".;!<>-#" else "x" + text
        elif r < 0.32:
            line = rng.choice([".shout ", "! ", ".sh "]) + random_text(rng)
        elif r < 0.42:
            line = rng.choice([".tell ", "> "]) + target + " " + random_text(rng)
        elif r < 0.50:
            line = rng.choice([";", ".emote ", "#", ".semote "]) + random_text(rng)
        elif r < 0.55:
            line = rng.choice(["< ", ".pemote "]) + target + " " + random_text(rng)
        elif r < 0.59:
            line = "- " + random_text(rng)
        elif r < 0.69:
            line = ".go " + rng.choice(ROOMS)
        elif r < 0.73:
            line = rng.choice([".look", ".review", ".revtell", ".review " + rng.choice(ROOMS)])
        elif r < 0.80:
            line = rng.choice([".ignall", ".ignshout", ".igntell"])
        elif r < 0.84:
            line = ".colour"
            colour[i] = not colour[i]
            flags = {"colour": colour[i]}
        elif r < 0.88:
            line = rng.choice([".wizshout ", ".wizshout wiz ", ".wizshout arch ", ".bcast "]) + random_text(rng)
        elif r < 0.91:
            line = rng.choice([".vis", ".invis"])
        elif r < 0.97:
            line = rng.choice([".clone " + rng.choice(ROOMS[:4]), ".destroy " + rng.choice(ROOMS[:4]), ".myclones",
                               ".csay " + rng.choice(ROOMS[:4]) + " " + random_text(rng),
                               ".chear " + rng.choice(ROOMS[:4]) + " " + rng.choice(["all", "swears", "nothing"])])
        else:
            line = rng.choice([".", ".bogus", ".say", ".tell", ".shout", ".cls"])
            if line == ".":
                continue        # the repeat buffer is overwritten by the sync command; covered in framing.json
        steps.append((k, line[:950], flags))

    def script(s):
        for k, n, c in zip(KEYS, NAMES, colour0):
            s.connect(k)
            s.login(k, n, colour=c)
        for k, line, flags in steps:
            s.line(k, line, **flags)

    cfg = {"ban_swearing": rng.random() < 0.5, "max_clones": 2}
    return cfg, accounts, script


@pytest.fixture(scope="module")
def writelog_shim(tmp_path_factory):
    from scenario_runner import build_writelog_shim
    return build_writelog_shim(tmp_path_factory.mktemp("shim"))


@pytest.mark.reference
@pytest.mark.parametrize("seed", [333, 1996, 7, 20261004, 42, 31337] + list(range(100, 118)))
def test_random_sessions_agree(seed, ref_binary, port_binary, monkeypatch, writelog_shim):
    name = f"__fuzz_{seed}"
    monkeypatch.setitem(scenarios.SCENARIOS, name, lambda: make_script(seed))
    ref_run = run_scenario(name, ref_binary, writelog_shim)
    port_run = run_scenario(name, port_binary, writelog_shim)
    ref, port = ref_run["steps"], port_run["steps"]
    assert len(ref) == len(port)
    for i, (a, b) in enumerate(zip(ref, port)):
        assert a == b, f"seed {seed}, step {i}: {a.get('actor')} sent {a.get('send')!r}\n reference: {a['recv']}\n port     : {b['recv']}"
    # ... and the same write(2) calls behind those bytes: sizes and order (tests/preload_writelog.c); the random text
    # includes lines long enough to cross write_user's 1000-byte staging buffer (nuts333.c:1359-1363)
    a, b = ref_run["write_sizes"][0], port_run["write_sizes"][0]
    if a != b:
        i = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
        raise AssertionError(f"seed {seed}: write #{i} of {len(a)}/{len(b)}: reference {a[max(0, i - 3):i + 4]} restatement {b[max(0, i - 3):i + 4]}")
    monkeypatch.setenv("NUTS_PORT_FAST", "1")
    fast = run_scenario(name, port_binary)["steps"]
    assert fast == ref, f"seed {seed}: fast mode diverges"


# ---------------------------------------------------------------- two talkers, mixed implementations
def make_netlink_script(seed: int):
    """Alice (talker 0) travels to talker 1 and stays; everybody talks at random."""
    rng = random.Random(seed)
    base = scenarios.netlink()
    colour = {"a": False, "b": False, "c": True, "d": False}
    steps = []
    for _ in range(45):
        k = rng.choice("aaabbcd")
        r = rng.random()
        flags = {}
        if r < 0.35:
            text = random_text(rng)
            line = text if text[0] not in ".;!<>-#" else "x" + text
        elif r < 0.55:
            line = ".shout " + random_text(rng)
        elif r < 0.75:
            line = ".tell " + rng.choice(["alice", "bobby", "carol", "dave", "nobody"]) + " " + random_text(rng)
        elif r < 0.85:
            line = rng.choice([";", "#"]) + random_text(rng)
        elif r < 0.92:
            line = rng.choice([".look", ".review", ".ignshout", ".igntell"])
        else:
            line = ".colour"
            colour[k] = not colour[k]
            flags = {"colour": colour[k]}
        steps.append((k, line[:300], flags))

    def script(s):
        s.connect("a", talker=0); s.login("a", "Alice")
        s.connect("d", talker=0); s.login("d", "Dave")
        s.connect("b", talker=1); s.login("b", "Bobby")
        s.connect("c", talker=1); s.login("c", "Carol", colour=True)
        for hop in pv.WALKS["lounge"]:
            s.line("b", f".go {hop}")
            s.line("c", f".go {hop}")
        s.line("a", ".go talker2", expect=b"has been set yet.\n\r")
        for k, line, flags in steps:
            s.line(k, line, **flags)

    return {**base, "script": script}


@pytest.mark.reference
@pytest.mark.parametrize("seed", [333, 1996])
def test_random_netlink_sessions_agree_across_implementations(seed, ref_binary, port_binary, monkeypatch):
    name = f"__nlfuzz_{seed}"
    monkeypatch.setitem(scenarios.SCENARIOS, name, lambda: make_netlink_script(seed))
    ref = run_scenario(name, [ref_binary, ref_binary])["steps"]
    for bins in ([port_binary, port_binary], [port_binary, ref_binary], [ref_binary, port_binary]):
        got = run_scenario(name, bins)["steps"]
        for i, (a, b) in enumerate(zip(ref, got)):
            assert a == b, (f"seed {seed}, talkers {[x.name for x in bins]}, step {i}: {a.get('actor')} sent {a.get('send')!r}\n"
                            f" reference: {a['recv']}\n got      : {b['recv']}")
        assert len(got) == len(ref)
