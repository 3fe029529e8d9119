"""Parity: the restated path reproduces, byte for byte, what the reference put on the wire.

tests/golden/*.json were captured from the unmodified reference build by
tests/golden/make_golden.py.  Here the same scripted clients are replayed against
oracle/_build/talker_port; every byte every client receives at every step must match.
Where the reference build is present (this container, not the GPU box) it is replayed too,
which keeps the fixtures honest; and the netlink scenario is additionally run with one side
the restatement and the other the real reference, in both orders -- the strongest
conformance check of the wire protocol available.
"""
from __future__ import annotations

BOTH_TIERS = True       # tests/conftest.py: every test here also runs in the gpu tier, on the MI355X box's host

import json
from pathlib import Path

import pytest

import scenarios
from scenario_runner import run_scenario

GOLDEN = Path(__file__).resolve().parent / "golden"
NAMES = list(scenarios.SCENARIOS)


def _load(name):
    return json.loads((GOLDEN / f"{name}.json").read_text())


def _diff(gold, got):
    for i, (a, b) in enumerate(zip(gold, got)):
        if a != b:
            lines = [f"first difference at step {i}: op={a.get('op')} actor={a.get('actor')} send={a.get('send')!r}"]
            for k in sorted(set(a["recv"]) | set(b["recv"])):
                if a["recv"].get(k) != b["recv"].get(k):
                    lines += [f"  client {k} golden: {a['recv'].get(k)!r}", f"  client {k} got   : {b['recv'].get(k)!r}"]
            return "\n".join(lines)
    return f"step count differs: golden {len(gold)} got {len(got)}"


def test_every_scenario_has_a_fixture():
    assert sorted(p.stem for p in GOLDEN.glob("*.json")) == sorted(NAMES)


@pytest.mark.parametrize("name", NAMES)
def test_port_matches_golden(name, port_binary):
    gold = _load(name)
    got = run_scenario(name, port_binary)
    assert got["steps"] == gold["steps"], _diff(gold["steps"], got["steps"])
    # user records written to disk at logout (save_user_details, nuts333.c:1645-1673), clock fields masked
    assert got.get("files") == gold.get("files")


@pytest.mark.host_only
@pytest.mark.parametrize("name", NAMES)
def test_port_fast_mode_puts_the_same_bytes_on_the_wire(name, port_binary, monkeypatch):
    """NUTS_PORT_FAST=1 (transduce once per colour variant, reset merged into the same write, TCP_NODELAY on
    netlinks -- INTEGRATION.md section 3) must be invisible to every client and to a netlink peer."""
    monkeypatch.setenv("NUTS_PORT_FAST", "1")
    gold = _load(name)
    got = run_scenario(name, port_binary)
    assert got["steps"] == gold["steps"], _diff(gold["steps"], got["steps"])


@pytest.mark.reference
@pytest.mark.parametrize("name", NAMES)
def test_reference_still_matches_golden(name, ref_binary):
    gold = _load(name)
    got = run_scenario(name, ref_binary)
    assert got["steps"] == gold["steps"], _diff(gold["steps"], got["steps"])
    assert got.get("files") == gold.get("files")


@pytest.fixture(scope="module")
def writelog_shim(tmp_path_factory):
    from scenario_runner import build_writelog_shim
    return build_writelog_shim(tmp_path_factory.mktemp("shim"))


@pytest.mark.reference
@pytest.mark.parametrize("name", NAMES)
def test_same_write_calls_in_the_same_order_as_the_reference(name, ref_binary, port_binary, writelog_shim):
    """Stronger than equal bytes: the restatement issues the same write(2) calls -- same sizes, same order, on client
    sockets and netlinks -- as the reference over the whole session (login, look, every command, every relay).  TCP
    hides write boundaries from a client; an LD_PRELOAD shim (tests/preload_writelog.c) logs them inside each talker.
    This is what makes the restatement's system-call cost representative when it stands in as cpu_baseline "port"."""
    ref = run_scenario(name, ref_binary, writelog_shim)["write_sizes"]
    port = run_scenario(name, port_binary, writelog_shim)["write_sizes"]
    assert [len(x) for x in ref] == [len(x) for x in port], "different number of write(2) calls"
    if len(ref) > 1:
        # two live talkers: whether a frame from the other talker or a local client's line is served first within one
        # select() wake-up depends on arrival time, so the ORDER of a talker's writes is not a function of the script
        # (seen: two adjacent writes swapped in 1 run of 3, in the reference as much as in the restatement).  Same calls,
        # same sizes, any order.
        ref, port = [sorted(x) for x in ref], [sorted(x) for x in port]
    for t, (a, b) in enumerate(zip(ref, port)):
        if a != b:
            i = next(i for i, (x, y) in enumerate(zip(a, b)) if x != y)
            raise AssertionError(f"talker {t}: write #{i} of {len(a)}: reference {a[max(0, i - 3):i + 4]} restatement {b[max(0, i - 3):i + 4]}")
    assert all(len(x) > 20 for x in ref)


@pytest.mark.reference
@pytest.mark.parametrize("order", ["port_dials_reference", "reference_dials_port"])
def test_netlink_interop_with_reference(order, port_binary, ref_binary):
    """talker 0 dials talker 1; mix the implementations across the link."""
    bins = [port_binary, ref_binary] if order == "port_dials_reference" else [ref_binary, port_binary]
    gold = _load("netlink")["steps"]
    got = run_scenario("netlink", bins)["steps"]
    assert got == gold, _diff(gold, got)


@pytest.mark.reference
@pytest.mark.parametrize("name", list(scenarios.REFERENCE_ONLY))
def test_reference_only_fixture(name, ref_binary):
    """SURVEY.md 8(f)4 remainder: board ``.B`` and mail ``.M`` files as the reference writes them.  The restatement
    does not implement boards or mail, so this is checked against the reference build only."""
    gold = json.loads((GOLDEN / "reference_only" / f"{name}.json").read_text())
    got = run_scenario(name, ref_binary)
    assert got["steps"] == gold["steps"], _diff(gold["steps"], got["steps"])
    assert got["files"] == gold["files"]


def test_board_and_mail_file_formats_in_the_fixture():
    """The formats themselves, read off the committed fixture (nuts333.c:5020-5040, 2462-2503; DOCS/about_dirs:25-30)."""
    files = json.loads((GOLDEN / "reference_only" / "board_mail_files.json").read_text())["files"]
    board = files["datafiles/drive.B"]
    posts = board.split("PT: T\r")[1:]                       # machine-readable posting time, then CR, then the header
    assert len(posts) == 4 and board.startswith("PT: T\r")
    for post, who in zip(posts, ("Alice", "Bobby", "Alice", "A presence")):
        head, _, body = post.partition("\n")
        assert head == f"~OLFrom: {who}  [ DATE ]"
        assert body.endswith("\n\n") and all(len(l) <= 80 for l in body.split("\n"))     # wrapped at 80 columns
    assert "w" * 80 + "\n" + "w" * 80 + "\n" + "w" * 10 + "\n\n" in posts[2]
    assert "~FRwith a colour command~RS" in posts[1]          # stored as typed; expanded when read
    mail = files["userfiles/Bobby.M"]
    stamp, _, rest = mail.partition("\r")
    assert stamp == "T"                                       # first line: time of the newest mail (has_unread_mail, c:2426-2438)
    assert rest.lstrip("\r") == ("~OLFrom: Alice  [ DATE ]\nhello by mail\n\n"
                                  "~OLFrom: Dave  [ DATE ]\na second mail: the new stamp goes on top, the old mail stays\n\n")


def test_fixtures_contain_no_reference_source():
    """A fixture is data: provisioning, inputs, received bytes -- nothing else."""
    for p in list(GOLDEN.glob("*.json")) + list((GOLDEN / "reference_only").glob("*.json")):
        d = json.loads(p.read_text())
        assert set(d) <= {"scenario", "config", "accounts", "steps", "files"}
        for st in d["steps"]:
            assert set(st) <= {"op", "actor", "name", "send", "note", "recv"}
            assert st["op"] in {"connect", "login", "line", "raw", "close", "dialog", "peer"}
