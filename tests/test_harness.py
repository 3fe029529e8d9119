"""Host logic: provisioner, launcher, load generator and the five BASELINE workloads.

Runs against the restatement (always available); where the reference build is present the
same small workloads are repeated against it.  Every run must deliver *exactly* the expected
number of lines to *each* client -- the harness counts per client, not in aggregate.
"""
from __future__ import annotations

import ctypes
import ctypes.util
import json
import os
import subprocess
import sys
import tempfile
from pathlib import Path

import pytest

from nuts333_amd import provision as pv
from nuts333_amd import workloads
from nuts333_amd.talker import Talker, free_ports

REPO = Path(__file__).resolve().parent.parent


# ---------------------------------------------------------------- provisioner
def test_password_hash_is_the_documented_known_answer():
    """DOCS/userdata_format:19 / motd1:4-5: crypt("test","NU") == NUKyNCCLvgLH."""
    path = ctypes.util.find_library("crypt")
    if not path:
        pytest.skip("libcrypt not found")
    lib = ctypes.CDLL(path)
    lib.crypt.restype = ctypes.c_char_p
    assert lib.crypt(b"test", b"NU") == pv.PASSWORD_HASH.encode()


def test_account_record_layout():
    rec = pv.Account("Uaaa", level=2, colour=1).render().splitlines()
    assert rec[0] == pv.PASSWORD_HASH and len(rec) == 6
    f = rec[1].split()
    assert len(f) == 10 and f[4] == "2" and f[9] == "1" and f[5] == "0"     # level, colour, prompt (c:1622)
    assert rec[2:] == ["localhost", "is a bot", "enters", "goes"]


def test_bot_names_are_legal_and_unique():
    names = [pv.bot_name(i) for i in range(1100)]
    assert len(set(names)) == 1100 and names[0] == "Uaaa" and names[27] == "Uabb"
    assert all(n.isalpha() and 3 <= len(n) <= 12 for n in names)
    with pytest.raises(ValueError):
        pv.write_account("/tmp", pv.Account("x1"))


def test_config_render_fits_the_parser_limits():
    text = pv.TalkerConfig(sites=[pv.Site("talker2", "127.0.0.1", 5002, "v")]).render()
    assert all(len(l) < 80 for l in text.splitlines())            # fgets(line,81) at c:466
    assert "INIT:" in text and "ROOMS:" in text and "SITES:" in text
    assert "lg lounge co BOTH ACCEPT" in text and "dr drive ha PUB" in text


def test_talker_ports_come_from_below_the_ephemeral_range():
    """A talker's three ports must not be ports the kernel may hand to a client connection as its source port between our
    probe and the talker's bind() (round 4: one "Can't bind to main port: Address already in use" in the suite)."""
    from nuts333_amd.talker import port_window
    e_lo, e_hi = (int(x) for x in Path("/proc/sys/net/ipv4/ip_local_port_range").read_text().split()[:2])
    lo, hi = port_window()
    got = [p for _ in range(50) for p in free_ports(3)]
    assert len(set(got)) == 150 and all(lo <= p < hi for p in got)
    # the exemption is port_window's own: when the window it returns lies outside the ephemeral range no drawn
    # port may be inside it; when the host's range leaves no 1000 ports on either side -- (1024, 65535), but also e.g.
    # (10000, 65000) -- the documented fallback 12000-20000 intersects the range and there is nothing to assert
    if hi <= e_lo or lo > e_hi:
        assert all(not e_lo <= p <= e_hi for p in got)
    else:
        assert (lo, hi) == (12000, 20000) and min(20000, e_lo) - 12000 < 1000 and 65535 - e_hi < 1000, (lo, hi, e_lo, e_hi)


def test_port_window_on_hosts_with_a_wide_ephemeral_range():
    """with ip_local_port_range tuned to start at or below 12000 the window below it was empty or negative
    (ZeroDivisionError / 'no free ports' at every talker boot).  The window then moves above the range, or stays put."""
    from nuts333_amd.talker import port_window
    assert port_window((32768, 60999)) == (12000, 20000)          # the stock range
    assert port_window((15000, 60999)) == (12000, 15000)          # below it, as long as 1000 ports fit
    assert port_window((12500, 60999)) == (61000, 65536)          # too little room below: above the top end
    assert port_window((1024, 60999)) == (61000, 65536)
    assert port_window((1024, 65535)) == (12000, 20000)           # no room anywhere outside: the plain window, never span <= 0
    for e in ((32768, 60999), (15000, 60999), (12500, 60999), (1024, 60999), (1024, 65535), (10000, 65000)):
        lo, hi = port_window(e)
        assert hi - lo >= 500 and 1024 < lo < hi <= 65536, (e, lo, hi)


def test_both_talkers_accept_the_generated_tree(tmp_path, port_binary):
    bins = [port_binary]
    from nuts333_amd.talker import REF_BINARY
    if REF_BINARY.exists():
        bins.append(REF_BINARY)
    for i, b in enumerate(bins):
        ports = free_ports(3)
        root = pv.write_tree(tmp_path / f"t{i}", pv.TalkerConfig(mainport=ports[0], wizport=ports[1], linkport=ports[2]),
                             [pv.Account("Fred", level=4)])
        with Talker(b, root) as t:
            assert t.alive() and t.pid > 0
        assert not t.alive()


def test_close_does_not_wait_for_a_broadcast_nobody_will_be_sent(tmp_path, port_binary):
    """the content-aware wait in Session.close() looks for 'SIGN OFF:' for up to 5 s.  With the only listener
    ignoring everything (write_room skips it, nuts333.c:1413) it used to sit out the whole 5 s; a listener that CAN hear it
    still ends the wait at once, with the broadcast in the step's capture."""
    import time
    from nuts333_amd.transcript import Session
    ports = free_ports(3)
    root = pv.write_tree(tmp_path / "t", pv.TalkerConfig(mainport=ports[0], wizport=ports[1], linkport=ports[2]),
                         [pv.Account("Alice"), pv.Account("Bobby"), pv.Account("Carol")])
    with Talker(port_binary, root):
        s = Session(ports[0])
        try:
            for k, n in (("a", "Alice"), ("b", "Bobby")):
                s.connect(k); s.login(k, n)
            # the toggle's words as PAYLOAD flip nothing -- not in the speaker's own "You say:" echo, not in the
            # listener's copy; only the talker's reply to the user who toggled, as a whole line, does
            s.line("a", "You are now ignoring everyone.")
            assert "says: You are now ignoring everyone." in s.steps[-1]["recv"]["b"]
            assert s.clients["a"].hears_broadcasts is True and s.clients["b"].hears_broadcasts is True
            s.line("b", ".ignall")
            assert s.clients["b"].hears_broadcasts is False
            s.line("b", "You will now hear everyone again.")
            assert "You say: You will now hear everyone again." in s.steps[-1]["recv"]["b"]
            assert s.clients["b"].hears_broadcasts is False and s.clients["a"].hears_broadcasts is True
            t = time.monotonic()
            s.close("a")
            assert time.monotonic() - t < 2.5 and "SIGN OFF" not in s.steps[-1]["recv"].get("b", "")
            s.line("b", ".ignall")
            assert s.clients["b"].hears_broadcasts is True
            s.connect("c"); s.login("c", "Carol")
            t = time.monotonic()
            s.close("c")
            assert time.monotonic() - t < 2.5 and "SIGN OFF: Carol" in s.steps[-1]["recv"]["b"]
        finally:
            s.shutdown()


def test_restatement_survives_the_fd_setsize_cliff(tmp_path, port_binary):
    """select() with FD_SETSIZE: the first descriptor >= 1024 cannot go into the mask.  The reference aborts there
    (-O2, fortified FD_SET) or corrupts memory (-O0) -- INTEGRATION.md section 4; the restatement turns the client
    away.  The daemon is started with 1000 descriptors already in use so that 40 connections reach the cliff."""
    import resource
    import socket
    soft, _ = resource.getrlimit(resource.RLIMIT_NOFILE)
    if soft < 1200:
        pytest.skip("RLIMIT_NOFILE too low")
    ports = free_ports(3)
    root = pv.write_tree(tmp_path / "t", pv.TalkerConfig(mainport=ports[0], wizport=ports[1], linkport=ports[2], max_users=2000))
    socks = []
    with Talker(port_binary, root, burn_fds=1000) as t:
        try:
            for _ in range(40):
                socks.append(socket.create_connection(("127.0.0.1", ports[0]), timeout=5))
            socks[-1].settimeout(5)
            data = b""
            while b"talker is full" not in data:
                chunk = socks[-1].recv(4096)
                if not chunk:
                    break
                data += chunk
            assert b"Sorry, the talker is full at the moment." in data
            assert t.alive()
        finally:
            for s in socks:
                s.close()


@pytest.mark.reference
def test_reference_dies_at_the_fd_setsize_cliff(tmp_path, ref_binary):
    """Documents the defect (nuts333.c:94,251-258,274): the fortified -O2 build aborts on FD_SET(1024)."""
    import resource
    import socket
    import time
    if resource.getrlimit(resource.RLIMIT_NOFILE)[0] < 1200:
        pytest.skip("RLIMIT_NOFILE too low")
    ports = free_ports(3)
    root = pv.write_tree(tmp_path / "t", pv.TalkerConfig(mainport=ports[0], wizport=ports[1], linkport=ports[2], max_users=2000))
    socks = []
    t = Talker(ref_binary, root, burn_fds=1000)
    t.start()
    try:
        for _ in range(40):
            try:
                socks.append(socket.create_connection(("127.0.0.1", ports[0]), timeout=2))
            except OSError:
                break
        time.sleep(0.3)
        assert not t.alive()
        assert b"buffer overflow detected" in (root / "boot.log").read_bytes()
    finally:
        for s in socks:
            s.close()
        t.stop()


def test_boot_failure_is_reported(tmp_path, port_binary):
    root = pv.write_tree(tmp_path / "bad", pv.TalkerConfig())
    (root / "datafiles" / "config").write_text("INIT:\nlogging YES\n")      # the shipped config2's bad option (c:599-621)
    with pytest.raises(RuntimeError):
        Talker(port_binary, root).start(timeout=3)


# ---------------------------------------------------------------- workloads, exact delivery
def _check(res, kind=""):
    assert res["ok"] and res["exact"] and res["per_client_exact"], (kind, res)
    assert res["deliveries"] == res["expected_deliveries"]
    assert res["acks"] == res["input_lines"] == res["planned_input_lines"]
    assert all(res.get("servers_alive_after", [res.get("server_alive_after", True)]))


def test_config1_single_client(port_binary):
    res = workloads.config1(lines=200, warmup=20, binary=port_binary)
    _check(res)
    assert res["deliveries"] == 0 and res["bytes_per_line"] == 65.0          # "You say: " + 54 + "\n\r"


def test_config2_ten_clients_one_room(port_binary):
    res = workloads.config2(lines=300, warmup=30, binary=port_binary)
    _check(res)
    assert res["deliveries"] == 300 * 9
    # 9 x "Uaaa says: "+54+2 = 67 B and 1 x "You say: "+54+2 = 65 B per input line
    assert abs(res["bytes_per_line"] - (9 * 67 + 65) / 10) < 1e-9


def test_config2_colour_on_costs_two_writes_and_more_bytes(port_binary):
    res = workloads.config2(lines=100, colour=1, binary=port_binary)
    _check(res)
    assert abs(res["bytes_per_line"] - ((9 * 67 + 65) / 10 + 8)) < 1e-9      # ESC[0m before \n\r and after


def test_config3_mixed_schedule_is_seeded_and_exact(port_binary):
    a = workloads.config3(per_client=10, n=25, binary=port_binary)
    b = workloads.config3(per_client=10, n=25, binary=port_binary)
    _check(a); _check(b)
    assert a["expected_deliveries"] == b["expected_deliveries"] and a["lines_total"] == b["lines_total"]
    c = workloads.config3(per_client=10, n=25, seed=334, binary=port_binary)
    assert c["expected_deliveries"] != a["expected_deliveries"]


def test_config3_six_room_variant(port_binary):
    res = workloads.config3(per_client=5, n=24, six_rooms=True, binary=port_binary)
    _check(res)
    assert "6 rooms" in res["workload"]


def test_config4_shout_fan_out(port_binary):
    res = workloads.config4(lines=20, n=120, warmup=2, binary=port_binary)
    _check(res)
    assert res["deliveries"] == 20 * 119
    assert abs(res["bytes_per_line"] - (119 * 69 + 67) / 120) < 1e-9          # "Uaaa shouts: " / "You shout: "


def test_config5_two_talkers_over_a_netlink(port_binary):
    res = workloads.config5(lines=30, warmup=3, binary=port_binary)
    _check(res)
    assert res["deliveries"] == 2 * 30 * 14 and len(res["servers"]) == 2
    _check_link_frames(res, lines=30, travellers=5)


def _check_link_frames(res, lines, travellers):
    """MEASURED link traffic (write(2) counts of each talker that did not go to one of its own clients) against
    what the protocol prescribes: one ACT frame per relayed command (nuts333.c:3801), one MSG..EMSG frame per
    line that reaches a travelling user (c:1302-1305), one PRM frame per relayed command (c:2181)."""
    nl = res["netlink"]
    assert nl["writes_t1_to_t2"] == lines
    assert nl["writes_t2_to_t1"] == (2 * lines * (travellers - 1) + 2 * lines) + lines
    assert nl["exact"]


@pytest.mark.parametrize("impl", ["port", "reference"])
def test_config5_link_frames_counted_on_the_wire(impl, port_binary, request):
    """The same count taken a second, independent way: a relay on the link parses both byte streams and counts
    frames by verb; the timed frames are told apart by the payload phrase.  Both methods and the formula agree."""
    binary = port_binary if impl == "port" else request.getfixturevalue("ref_binary")
    lines, travellers = 25, 4
    res = workloads.config5(lines=lines, travellers=travellers, tap=True, binary=binary)
    _check(res)
    _check_link_frames(res, lines, travellers)
    tap = res["netlink"]["tap"]
    assert tap["marked_dial_to_accept"] == {"ACT": lines}
    assert tap["marked_accept_to_dial"] == {"MSG": 2 * lines * (travellers - 1) + 2 * lines}
    # whole session: every MSG is closed by an EMSG; one PRM per relayed command -- the timed shouts plus each
    # traveller's arrival `look` (nuts333.c:3218-3224); one TRANS and one GRANTED per traveller plus GRANTED CONNECT
    assert tap["accept_to_dial"]["MSG"] == tap["accept_to_dial"]["EMSG"]
    assert tap["accept_to_dial"]["PRM"] == lines + travellers == tap["dial_to_accept"]["ACT"]
    assert tap["dial_to_accept"]["TRANS"] == travellers and tap["accept_to_dial"]["GRANTED"] == travellers + 1


def _syscalls(res):
    s = res["servers"][0]
    return s["read_syscalls"], s["write_syscalls"], s["bytes_written"]


def test_syscall_cost_model_of_the_restatement(port_binary, monkeypatch):
    """Exact counts from /proc/<pid>/io: one read(2) per input line, one write(2) per written line, two
    with colour on (nuts333.c:136, 1363, 1365); the fast mode folds the reset into the same write."""
    monkeypatch.delenv("NUTS_PORT_FAST", raising=False)
    off = workloads.config2(lines=400, binary=port_binary)
    on = workloads.config2(lines=400, colour=1, binary=port_binary)
    assert _syscalls(off) == (400, 4000, off["bytes_total"]) and _syscalls(on) == (400, 8000, on["bytes_total"])
    monkeypatch.setenv("NUTS_PORT_FAST", "1")
    fast = workloads.config2(lines=400, colour=1, binary=port_binary)
    assert _syscalls(fast) == (400, 4000, on["bytes_total"]) and fast["exact"]


@pytest.mark.reference
def test_restatement_issues_the_same_system_calls_as_the_reference(ref_binary, port_binary, monkeypatch):
    """The restatement stands in as cpu_baseline kind "port": it must cost what the reference costs.
    Same workload, same number of read(2)/write(2) calls, same bytes -- colour off and on, N = 10 and 200."""
    monkeypatch.delenv("NUTS_PORT_FAST", raising=False)
    for kw in ({"lines": 400}, {"lines": 400, "colour": 1}):
        assert _syscalls(workloads.config2(binary=ref_binary, **kw)) == _syscalls(workloads.config2(binary=port_binary, **kw))
    a = workloads.config4(lines=30, n=200, binary=ref_binary)
    b = workloads.config4(lines=30, n=200, binary=port_binary)
    assert _syscalls(a) == _syscalls(b) == (30, 30 * 200, a["bytes_total"])
    c = workloads.config3(per_client=10, n=25, binary=ref_binary)
    d = workloads.config3(per_client=10, n=25, binary=port_binary)
    assert _syscalls(c) == _syscalls(d)


@pytest.mark.reference
def test_small_workloads_against_the_reference(ref_binary):
    _check(workloads.config2(lines=300, warmup=30, binary=ref_binary), "config2")
    _check(workloads.config3(per_client=10, n=25, binary=ref_binary), "config3")
    _check(workloads.config4(lines=20, n=120, binary=ref_binary), "config4")
    r5 = workloads.config5(lines=30, binary=ref_binary)
    _check(r5, "config5")
    _check_link_frames(r5, lines=30, travellers=5)


def _login_write_sizes(binary, tmp_path, tag, name, colour, shim):
    """Sizes of the write(2) calls a talker issues on the client socket for: accept + motd1, then name / password +
    motd2 + look, logged by the LD_PRELOAD shim tests/preload_writelog.c."""
    import scenarios
    import socket
    import time
    ports = free_ports(3)
    root = pv.write_tree(tmp_path / tag, pv.TalkerConfig(mainport=ports[0], wizport=ports[1], linkport=ports[2]),
                         [pv.Account(name, colour=colour)])
    (root / "motd1").write_text(scenarios.LONG_MOTD1)
    (root / "motd2").write_text(scenarios.LONG_MOTD2)
    log = tmp_path / f"{tag}.writes"

    def read_until(sock, needle):
        buf = b""
        while needle not in buf:
            chunk = sock.recv(65536)
            assert chunk, buf[-200:]
            buf += chunk
        return buf

    t = Talker(binary, root, extra_env={"LD_PRELOAD": str(shim), "WRITELOG": str(log)})
    t.start()
    try:
        s = socket.create_connection(("127.0.0.1", ports[0]), timeout=10)
        got = read_until(s, b"Give me a name: ")
        s.sendall(name.encode() + b"\n"); got += read_until(s, b"Give me a password: ")
        s.sendall(b"test\n"); got += read_until(s, b"has been set yet.")
        time.sleep(0.1)
        s.close()
    finally:
        t.stop()
    return [int(l.split()[2]) for l in log.read_text().splitlines()], got


@pytest.mark.reference
@pytest.mark.parametrize("colour", [0, 1])
def test_more_flushes_where_the_reference_does_for_long_banners(tmp_path, ref_binary, port_binary, colour):
    """more() stages a whole file through one 1000-byte buffer (nuts333.c:2205-2296); with banners over
    1 KB the write(2) BOUNDARIES -- not just the bytes, and not just the count -- must match.  motd1 (pre-login, never
    coloured) crosses all three flush rules: before a newline with > 994 staged, before a '~' with > 994 staged, and
    at exactly 1000; motd2 is read with the account's colour flag."""
    from scenario_runner import build_writelog_shim
    shim = build_writelog_shim(tmp_path)
    ref, ref_bytes = _login_write_sizes(ref_binary, tmp_path, "ref", "Alice", colour, shim)
    port, port_bytes = _login_write_sizes(port_binary, tmp_path, "port", "Alice", colour, shim)
    assert ref_bytes == port_bytes
    assert ref == port, (ref, port)
    # motd1 as crafted in tests/scenarios.py: 995 | "\n\r" + 993 | "x\n\r" + 997 = 1000 | the rest
    i = ref.index(995)
    assert ref[i:i + 3] == [995, 995, 1000] and sum(ref) == len(ref_bytes)


def test_loadgen_reports_a_rejected_login(tmp_path, port_binary):
    """Unprovisioned account -> the talker asks to confirm a new password -> hard failure, not a hang."""
    ports = free_ports(3)
    root = pv.write_tree(tmp_path / "t", pv.TalkerConfig(mainport=ports[0], wizport=ports[1], linkport=ports[2]))
    with Talker(port_binary, root) as t:
        spec = workloads.Spec()
        c = spec.add_client("Nobody", ports[0])
        spec.add_line(c, "hello", [])
        with pytest.raises(RuntimeError):
            workloads.run_spec(spec, [t], timeout_s=10)


def test_write_probes_produce_numbers(built):
    p = workloads.probe_write(67, 20000)
    assert p["cpu_ns_per_write"] > 0
    out = subprocess.run([str(workloads.LOADGEN_BIN), "--probe-fanout", "67", "9", "500"], check=True, stdout=subprocess.PIPE).stdout
    j = json.loads(out)
    assert j["written_lines_per_s_cpu"] > 0 and j["writes_per_line"] == 9 and j["select_read"] == 0 and j["bytes_ok"]


@pytest.mark.parametrize("open_loop", [0, 1])
def test_line_probe_does_the_per_input_line_system_calls(built, open_loop):
    """--probe-line: 1 select(FD_SETSIZE) + 1 read + (recipients+1) writes per round; every byte
    written must arrive; the select+read share is reported on its own."""
    out = subprocess.run([str(workloads.LOADGEN_BIN), "--probe-line", "69", "49", "400", "1", str(open_loop), "2"],
                         check=True, stdout=subprocess.PIPE).stdout
    j = json.loads(out)
    assert j["mode"] == ("open" if open_loop else "closed") and j["select_read"] == 1 and j["select_nfds"] == 1024
    assert j["writes_per_line"] == 50 and j["recipients"] == 49 and j["bytes_ok"] is True
    assert 0 < j["cpu_ns_select_read_per_line"] < j["cpu_ns_per_line"]
    # the two shares add up (values are printed to 0.1 ns, so allow 50 x 0.05 of rounding)
    assert abs(j["cpu_ns_per_line"] - (j["cpu_ns_select_read_per_line"] + 50 * j["cpu_ns_per_write"])) < 5.0


# ---------------------------------------------------------------- bench.py contract
CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _bench(*args, env=None, full=False):
    """One bench.py run -> its stdout line (the compact record); with ``full`` also the full record it wrote beside it."""
    with tempfile.TemporaryDirectory() as tmp:
        record = Path(tmp) / "full.json"
        out = subprocess.run([sys.executable, str(REPO / "bench.py"), *args], check=True, stdout=subprocess.PIPE,
                             env={**os.environ, "NUTS_BENCH_FULL_RECORD": str(record), **(env or {})},
                             timeout=900).stdout.decode().strip().splitlines()
        assert len(out) == 1, out
        j = json.loads(out[0])
        # the driver's record keeps ~8 KB of stdout+stderr tail: the line must fit whole (round 3's 12 KB line did not)
        assert len(out[0]) <= 6000, len(out[0])
        if not full:
            return j
        assert j["full_record"] == str(record)
        return j, json.loads(record.read_text())


def _assert_compact_free_text(line: dict, full: dict) -> None:
    """What compact_line PROMISES of the free-text fields: counts exact, the kept entries cut as documented --
    never `line == full`, which fails on exactly the busy hosts the warnings exist for."""
    sys.path.insert(0, str(REPO))
    import bench
    assert line["warnings_count"] == len(full["warnings"])
    assert line["warnings"] in [bench.compact_warnings(full["warnings"], t) for t in bench.TIGHT_LEVELS]
    assert len(line["warnings"]) == min(bench.WARNINGS_KEPT, len(full["warnings"]))
    if "extras_errors" in full:
        assert line["extras_errors_count"] == len(full["extras_errors"])
        assert line["extras_errors"] in [bench.compact_errors(full["extras_errors"], t) for t in bench.TIGHT_LEVELS]


def test_bench_single_replica_contract():
    line, j = _bench("--steps", "2", "--warmup", "1", "--lines-per-step", "200", "--binary", "port", "--workload", "config2", full=True)
    assert CONTRACT_KEYS <= set(line) and CONTRACT_KEYS <= set(j)
    # the compact line copies, never recomputes: every figure it carries is the full record's
    for k in ("value", "ms_per_step", "delivered", "expected_delivered", "cpu_baseline", "configs_all_exact"):
        assert line[k] == j[k], k
    _assert_compact_free_text(line, j)
    assert {k: v for k, v in line["roofline"].items() if k not in ("note", "unit", "probe_legs")} == \
           {k: v for k, v in j["roofline"].items() if k not in ("note", "unit", "probe")}
    # each full probe leg explains itself in the line -- rates, CPU/wall of the median repetition, load average
    for short, leg in (("open", "full_open"), ("closed", "full_closed")):
        got, src = line["roofline"]["probe_legs"][short], j["roofline"]["probe"][leg]
        assert got["wall_all"] == src["written_lines_per_s_wall_all"] and len(src["cpu_over_wall_all"]) == 3
        assert got["cpu_over_wall"] in src["cpu_over_wall_all"] and 0 < got["cpu_over_wall"] <= 1.05
        assert got["loadavg_before"] == src["loadavg_before"][0]
    assert "cpu_baseline_O0" not in j          # --binary port: the as-shipped-flags leg belongs to a reference headline
    assert [(c["name"], c["delivered_lines_per_s"], c["exact"], c["rate_all_reps"]) for c in line["configs"]] == \
           [(c["name"], c["delivered_lines_per_s"], c["exact"], c["rate_all_reps"]) for c in j["configs"]]
    assert line["host"]["loadavg_before_run"] == j["host"]["loadavg_before_run"]
    assert "note" in line["configs"][3] or len(json.dumps(line)) > 5400          # (notes go first when warnings crowd the line)
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["gpu_used"] is False
    assert j["delivered"] == j["expected_delivered"] == 2 * 200 * 9
    assert j["vs_baseline"] is None and j["scaling"] == "weak" and j["higher_is_better"] is True
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] == 1
    r = j["roofline"]
    assert r["bound"] == "host-syscall" and 0 < r["frac"] < 2 and r["traffic"] is None
    # the ceiling counts what SURVEY.md 8(d) says one input line costs: 1 select + 1 read + (9 + 1) writes
    assert r["per_input_line"] == {**r["per_input_line"], "select": 1, "read": 1, "write": 10, "select_nfds": 1024}
    assert set(r["probe"]) == {"write_only_closed", "full_closed", "full_open"}
    # the ceiling is a DEMONSTRATED rate: the best wall-clock rate any repetition of a full leg reached;
    # frac follows from the probe fields by one division; the CPU-time extrapolation stands beside it
    demonstrated = r["probe"]["full_open"]["written_lines_per_s_wall_all"] + r["probe"]["full_closed"]["written_lines_per_s_wall_all"]
    assert len(demonstrated) == 6 and r["peak"] == max(demonstrated)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["peak_write_only"] > 0
    assert abs(r["frac_extrapolated"] - r["achieved"] / r["peak_extrapolated"]) < 1e-3
    assert all(leg["bytes_ok"] and leg["reps"] == 3 for leg in r["probe"].values())
    assert isinstance(j["warnings"], list) and j["diagnostics"]["placement"]["talker_cpus"]
    for k in ("wall_s", "server_busy_frac", "server_run_delay_frac", "ack_latency_us", "progress_gaps", "sender_receiver_busy_frac"):
        assert k in j["diagnostics"], k
    assert "model" not in j["config"] and "workload" in j["config"]
    # every BASELINE configuration is in the same line, exact, with measured link frames for #5
    names = [c["name"] for c in j["configs"]]
    assert names == ["config1", "config2", "config3", "config3_six_rooms", "config4", "config5"]
    assert j["configs_all_exact"] and j["extras_errors"] == []
    assert [c["n"] for c in j["configs"]] == [1, 10, 100, 100, 1000, 20] and [c["reps"] for c in j["configs"]] == [3, 3, 3, 3, 3, 1]
    # BASELINE.json's "all 6 rooms" and the shipped five, side by side at the formal size, each saying which it is
    five, six = j["configs"][2], j["configs"][3]
    assert "over 5 rooms" in five["workload"] and "over 6 rooms" in six["workload"] and "datafiles/config:34-39" in five["note"]
    assert six["input_lines"] == five["input_lines"] == 20000 and six["delivered"] != five["delivered"]
    assert j["configs"][5]["netlink"]["writes_t2_to_t1"] == 11000 and j["configs"][5]["netlink"]["writes_t1_to_t2"] == 1000


def test_bench_extras_cannot_cost_the_result_line(monkeypatch):
    """A configuration that fails inside `configs` is reported in the line, the others still run (aggregation logic
    only: the workloads are stubbed)."""
    sys.path.insert(0, str(REPO))
    import bench

    def fake(rate, n, deliveries=10):
        return {"clients": n, "workload": "stub", "delivered_lines_per_s": rate, "input_lines_per_s": rate / 10, "input_lines": 1,
                "deliveries": deliveries, "expected_deliveries": deliveries, "wall_s": 1.0, "exact": True,
                "ack_latency_us": {"p50": 1.0}, "servers": [{"cpu_us_per_written_line": 1.0, "busy_frac": 1.0}]}

    rates = iter([300.0, 100.0, 200.0])
    monkeypatch.setattr(bench.workloads, "config1", lambda **kw: fake(next(rates), 1))
    monkeypatch.setattr(bench.workloads, "config2", lambda **kw: (_ for _ in ()).throw(RuntimeError("talker did not boot")))
    monkeypatch.setattr(bench.workloads, "config3", lambda **kw: fake(5.0, 100))
    monkeypatch.setattr(bench.workloads, "config4", lambda **kw: fake(7.0, 1000))
    monkeypatch.setattr(bench.workloads, "config5", lambda **kw: {**fake(9.0, 20), "netlink": {"exact": True}})
    errors = []

    def attempt(what, fn):
        try:
            return fn()
        except Exception as e:
            errors.append(f"{what}: {e}")
            return None

    out = bench.all_configs(Path("x"), True, "config4", fake(8.0, 1000), attempt, headline_size=(1000, 20))
    assert [e["name"] for e in out] == ["config1", "config2", "config3", "config3_six_rooms", "config4", "config5"]
    assert out[0]["delivered_lines_per_s"] == 200.0 and out[0]["rate_all_reps"] == [300.0, 100.0, 200.0]     # the median run
    assert out[1]["exact"] is False and "error" in out[1] and len(errors) == 3
    assert out[4]["includes_headline_run"] and out[4]["reps"] == 3 and out[4]["rate_all_reps"] == [8.0, 7.0, 7.0]
    assert out[5]["exact"] and out[5]["netlink"] == {"exact": True}
    # a timed run of another size (--steps 5, --lines-per-step ...) is NOT one of the formal-size repetitions
    errors.clear()
    for size in ((500, 100), None):
        other = bench.all_configs(Path("x"), True, "config4", fake(8.0, 1000), attempt, headline_size=size)
        assert other[4]["includes_headline_run"] is False and other[4]["reps"] == 3 and other[4]["rate_all_reps"] == [7.0, 7.0, 7.0]


def test_bench_exit_code_is_nonzero_when_a_configuration_is_inexact(monkeypatch, capsys, port_binary, tmp_path):
    """the line still prints (fault-tolerant extras), but a driver that reads only the exit code
    learns that the record is not exact."""
    sys.path.insert(0, str(REPO))
    import bench
    monkeypatch.setenv("NUTS_BENCH_FULL_RECORD", str(tmp_path / "full.json"))
    real = workloads.config2(lines=60, warmup=5, binary=port_binary)
    assert real["exact"]
    monkeypatch.setattr(bench, "run_workload", lambda *a, **kw: dict(real))
    monkeypatch.setattr(bench, "syscall_roofline", lambda *a, **kw: None)
    monkeypatch.setattr(bench, "device_floor_in_child", lambda: None)
    verdicts = iter([[{"name": "config1", "exact": True}, {"name": "config2", "exact": False, "error": "stub"}],
                     [{"name": "config1", "exact": True}]])
    monkeypatch.setattr(bench, "all_configs", lambda *a, **kw: next(verdicts))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "1", "--warmup", "0", "--workload", "config2", "--binary", "port"])
    assert bench.main() == 1
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["configs_all_exact"] is False and line["value"] > 0
    assert bench.main() == 0
    assert json.loads(capsys.readouterr().out.strip().splitlines()[-1])["configs_all_exact"] is True


def test_stall_attribution_from_a_runs_own_counters():
    """a leg at equal CPU per line and several times the wall clock must say where the time went."""
    def run(busy, run_delay, sender_busy, gaps_s=0.0):
        wall = 10.0
        return {"wall_s": wall, "ack_latency_us": {"p50": 1300.0, "p99": 1400.0, "max": 250000.0},
                "slow_acks": {"threshold_us": 13000.0, "count": 40, "total_s": gaps_s},
                "progress_gaps": {"threshold_ms": 5.0, "count": 40, "total_s": gaps_s, "max_ms": 200.0},
                "placement": {"talker_cpus": [0], "receiver_cpus": [1, 2, 3, 4]},
                "servers": [{"busy_frac": busy, "run_delay_frac": run_delay, "sleep_frac": max(0.0, 1 - busy - run_delay),
                             "involuntary_switches": 999, "voluntary_switches": 5, "cpu_us_per_written_line": 1.43}],
                "workers": [{"senders": 1, "busy_frac": sender_busy, "run_delay_s": (1 - sender_busy) * wall, "cpu_s": sender_busy * wall},
                            {"senders": 0, "busy_frac": 1.0, "run_delay_s": 0.0, "cpu_s": wall}]}
    assert workloads.attribute_stall(run(0.97, 0.01, 1.0)) is None
    assert workloads.attribute_stall(run(0.19, 0.0, 1.0), saturating=False) is None          # config #1 / #5 never saturate
    assert "RUNNABLE but off its core" in workloads.attribute_stall(run(0.19, 0.80, 1.0))
    msg = workloads.attribute_stall(run(0.19, 0.0, 0.15, gaps_s=8.0))
    assert "ASLEEP in select()" in msg and "0.15 of its core" in msg and "8.00 s in 40 gaps" in msg
    assert "throttled this container in 7 periods" in workloads.attribute_stall(run(0.19, 0.80, 1.0), throttled_periods=7)
    assert "unattributed" in workloads.attribute_stall(run(0.19, 0.0, 1.0))


def test_throttling_is_blamed_first_only_when_it_can_explain_the_window():
    """the cgroup counters are taken around the whole leg (boot + 1000 logins + warm-up + window); one throttled
    period during login must not label a window whose talker shows no run-queue wait."""
    def run(run_delay, sender_busy):
        return {"wall_s": 10.0, "spin": 1, "ack_latency_us": {"p50": 1300.0, "p99": 1400.0, "max": 250000.0}, "slow_acks": None,
                "progress_gaps": {"threshold_ms": 5.0, "count": 4, "total_s": 1.0, "max_ms": 200.0},
                "placement": {"talker_cpus": [0], "receiver_cpus": [1]},
                "servers": [{"busy_frac": 0.2, "run_delay_frac": run_delay, "sleep_frac": 0.8 - run_delay, "involuntary_switches": 3,
                             "voluntary_switches": 5, "cpu_us_per_written_line": 1.4}],
                "workers": [{"senders": 1, "busy_frac": sender_busy, "run_delay_s": 8.0, "cpu_s": 1.5}]}
    msg = workloads.attribute_stall(run(0.0, 0.15), throttled_periods=1, throttled_ms=12.0)
    assert "ASLEEP in select()" in msg and "context: cgroup CPU quota throttled this container in 1 periods (12 ms)" in msg
    assert " -- cgroup CPU quota throttled" in workloads.attribute_stall(run(0.0, 0.15), throttled_periods=60, throttled_ms=5000.0)
    assert " -- cgroup CPU quota throttled" in workloads.attribute_stall(run(0.3, 1.0), throttled_periods=2, throttled_ms=10.0)
    # workers that do not busy-poll (spin 0) are never blamed for being idle
    quiet = {**run(0.0, 0.15), "spin": 0}
    assert "unattributed" in workloads.attribute_stall(quiet)


def test_client_bound_replicas_say_so_in_the_line():
    """8 replicas under the 1-GPU box's 16-core quota get int(16/8 - 1.5) = 0 -> 1 receiver thread each,
    which DESIGN.md section 5's sweep calls client-bound: the line must carry that, in the words the review asked for."""
    sys.path.insert(0, str(REPO))
    import bench
    assert max(1, min(4, int(16.0 / 8 - 1.5))) == 1 and max(1, min(4, int(16.0 / 4 - 1.5))) == 2      # bench.main's arithmetic
    w = bench.client_bound_warning("config4", 8, 16.0, 1, 256)
    assert "8 replica(s) under a 16-core quota leave 1 receiver thread each: client-bound" in w and "measures the quota" in w
    assert bench.client_bound_warning("config4", 4, 16.0, 2, 256) is None
    assert bench.client_bound_warning("config5", 8, 16.0, 1, 256) is None          # never saturates the talker anyway
    assert "on 2 schedulable CPUs" in bench.client_bound_warning("config2", 1, None, 1, 2)


def test_quota_bound_replica_run_carries_the_warning(monkeypatch, capsys, port_binary, tmp_path):
    """The same through bench.main(): a monkeypatched 3-core quota leaves one receiver thread, and the line warns."""
    sys.path.insert(0, str(REPO))
    import bench
    monkeypatch.setenv("NUTS_BENCH_FULL_RECORD", str(tmp_path / "full.json"))
    monkeypatch.setattr(bench.workloads, "cgroup_cpu_quota", lambda: 3.0)
    monkeypatch.setattr(bench.workloads, "MAX_CLIENT_THREADS", 4)          # restored after the test: main() lowers it
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "1", "--warmup", "0", "--lines-per-step", "100", "--workload", "config2",
                                      "--binary", "port", "--no-extras"])
    assert bench.main() == 0
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["host"]["receiver_threads_per_replica"] == 1 and line["host"]["cgroup_cpu_quota_cores"] == 3.0
    # (first in the list; a 100-line run with one receiver may ALSO be reported as a harness stall, which is the point)
    assert line["warnings_count"] >= 1 and "under a 3-core quota leave 1 receiver thread each: client-bound" in line["warnings"][0]
    assert line["delivered"] == line["expected_delivered"] == 900


def test_repetitions_far_apart_on_a_busy_talker_are_named_as_a_slow_core():
    """Round 4's confirmation run on the box (load average 39): config #4 read 623 k / 466 k / 538 k lines/s with the talker 1.00 busy
    every time and `warnings: []`.  Nothing stalled -- the core was slower (1.85 us per line against 1.31) -- and the line now says so."""
    sys.path.insert(0, str(REPO))
    import bench

    def run(rate, busy, cost):
        return {"delivered_lines_per_s": rate, "servers": [{"busy_frac": busy, "cpu_us_per_written_line": cost}]}
    w = bench.slow_core_warning("config4", [run(622932.9, 1.0, 1.60), run(466427.7, 0.996, 2.14), run(537852.9, 0.997, 1.85)])
    assert w.startswith("config4: repetitions 622,933 / 466,428 / 537,853") and "x1.34" in w and "1.60 / 2.14 / 1.85 us" in w
    assert bench.slow_core_warning("config4", [run(750e3, 1.0, 1.3), run(760e3, 1.0, 1.3), run(740e3, 1.0, 1.3)]) is None
    assert bench.slow_core_warning("config4", [run(750e3, 1.0, 1.3), run(400e3, 0.5, 1.3)]) is None      # a stall: attribute_stall's business
    assert bench.slow_core_warning("config1", [run(200e3, 0.6, 3.0), run(150e3, 0.6, 3.0)]) is None      # never saturates the talker


def test_roofline_fraction_outside_its_band_is_named_in_the_line():
    """VERDICT r3 item 4 / ADVICE r3: below 0.85 or above 1.02 the line carries a 'roofline:' warning; the gpu tier asserts
    that invariant rather than the band itself."""
    sys.path.insert(0, str(REPO))
    import bench
    rf = {"frac": 0.95, "frac_extrapolated": 0.96, "probe": {"full_open": {"written_lines_per_s_wall_all": [1, 2, 3]}}}
    assert bench.roofline_warnings(rf, [20.0, 0, 0]) == [] and bench.roofline_warnings(None, None) == []
    low = bench.roofline_warnings({**rf, "frac": 0.831, "frac_extrapolated": 1.051}, [64.0, 0, 0])
    assert len(low) == 1 and low[0].startswith("roofline:") and "0.831" in low[0] and "1.051" in low[0] and "read 64" in low[0]
    high = bench.roofline_warnings({**rf, "frac": 1.04}, None)
    assert len(high) == 1 and high[0].startswith("roofline:") and "cannot exceed" in high[0]


def test_compact_line_fits_the_drivers_tail_even_with_long_warnings():
    """Round 3's line was 12 KB and the driver's record kept its last ~8 KB: load average and the restatement leg were cut
    off.  Built here from a committed full record of that very shape, with six 1 KB warnings on top."""
    sys.path.insert(0, str(REPO))
    import bench
    full = json.loads((REPO / "profiles" / "bench_r03_driverargs_run4_final_mi355xhost.json").read_text())
    line = bench.compact_line(full, "gpurun_out/bench_full_n1.json")
    assert len(json.dumps(line)) <= bench.LINE_BUDGET
    assert line["host"]["loadavg_before_run"] == full["host"]["loadavg_before_run"]
    assert line["cpu_baseline_port"]["ratio_to_timed_run"] == full["cpu_baseline_port"]["ratio_to_timed_run"]
    assert line["cpu_baseline_port"]["busy_all_reps"] == [d["server_busy_frac"] for d in full["cpu_baseline_port"]["diagnostics_all_reps"]]
    noisy = {**full, "warnings": [f"config4 repetition {k}: " + "x" * 1000 for k in range(9)]}
    line = bench.compact_line(noisy, "gpurun_out/bench_full_n1.json", tight=1)
    assert len(json.dumps(line)) <= bench.LINE_BUDGET and line["warnings_count"] == 9 and len(line["warnings"]) == 6
    assert line["value"] == full["value"] and line["roofline"]["frac"] == full["roofline"]["frac"]


def test_line_with_round5_fields_and_a_full_set_of_long_warnings_still_fits():
    """`probe_legs` + `cpu_baseline_O0` (about 430 bytes) used up the budget's headroom -- on round 5's own full
    record, five or six 500-character warnings of distinct kinds gave 6,158 / 6,322 bytes in the tightest form there was,
    and the whole-warning stderr echo added 3 KB to the same ~8 KB tail.  Replayed here for every count 0..WARNINGS_KEPT and
    for a worst case with more warnings than slots and eight 1 KB errors: the printed line fits, the figures are untouched,
    the counts are whole, and what is cut is only prose."""
    sys.path.insert(0, str(REPO))
    import bench
    full = json.loads((REPO / "profiles" / "bench_r05_driverargs_mi355xhost_full.json").read_text())
    assert "cpu_baseline_O0" in full and "full_open" in full["roofline"]["probe"]            # a record of the round-5 shape
    kinds = ["roofline: ", "probe: ", "restatement/reference delivered-rate ratio ", "8 replica(s) client-bound ",
             "config4: the core was slower ", "config2 repetition 1: harness stall: "]
    assert len({bench._warning_kind(k) for k in kinds}) == len(kinds) == bench.WARNINGS_KEPT
    plain = json.loads(bench.render_line({**full, "warnings": []}, "gpurun_out/bench_full_n1.json"))
    for n in range(bench.WARNINGS_KEPT + 1):
        noisy = {**full, "warnings": [k + "x" * (500 - len(k)) for k in kinds[:n]]}
        text = bench.render_line(noisy, "gpurun_out/bench_full_n1.json")
        assert len(text) <= bench.LINE_BUDGET, (n, len(text))
        line = json.loads(text)
        _assert_compact_free_text(line, noisy)
        for k in ("value", "ms_per_step", "cpu_baseline", "cpu_baseline_port", "device_floor", "host", "diagnostics", "configs_all_exact"):
            assert line[k] == plain[k], (n, k)
        for k in ("achieved", "peak", "frac", "frac_extrapolated", "frac_write_only", "probe_legs"):
            assert line["roofline"][k] == plain["roofline"][k], (n, k)
        assert {k: v for k, v in line["cpu_baseline_O0"].items() if k != "flags"} == {k: v for k, v in plain["cpu_baseline_O0"].items() if k != "flags"}
        assert [(c["name"], c.get("delivered_lines_per_s"), c["exact"], c["rate_all_reps"]) for c in line["configs"]] == \
               [(c["name"], c.get("delivered_lines_per_s"), c["exact"], c["rate_all_reps"]) for c in plain["configs"]]
    worst = {**full, "warnings": [k + "x" * 500 for k in kinds] + ["config3 repetition 2: harness stall: " + "y" * 500] * 4,
             "extras_errors": ["device floor: " + "e" * 1000] * 8}
    text = bench.render_line(worst, "gpurun_out/bench_full_n1.json")
    assert len(text) <= bench.LINE_BUDGET, len(text)
    line = json.loads(text)
    assert line["warnings_count"] == 10 and len(line["warnings"]) == 6 and line["extras_errors_count"] == 8 and len(line["extras_errors"]) == 6
    assert [bench._warning_kind(w) for w in line["warnings"]] == [bench._warning_kind(k) for k in kinds]      # still one of each kind
    # the stderr echo shares the tail with the line: what main() prints there is the kept selection in the 160-character form
    echo = sum(len(f"[bench] WARNING: {w}\n") for w in bench.compact_warnings(worst["warnings"], 1))
    assert echo <= bench.WARNINGS_KEPT * 180 and len(text) + echo + 600 <= 8000


def test_as_shipped_flags_leg_cannot_go_wrong_in_silence(monkeypatch):
    """the -O0 leg was the only one whose ratio or exactness could be off without a warning or an exit code.
    Stubbed talker runs: in the band nothing is said; outside it the line names the ratio (same kind as the restatement
    leg's sentence, so it shares that slot); an inexact repetition is named and `exact` is false (main() exits 1 on it)."""
    sys.path.insert(0, str(REPO))
    import bench

    def run(rate, exact=True, cpu=1.41):
        return {"delivered_lines_per_s": rate, "exact": exact, "input_lines": 2000, "deliveries": 1998000,
                "servers": [{"cpu_us_per_written_line": cpu, "busy_frac": 1.0}]}
    timed = run(750000.0, cpu=1.32)
    monkeypatch.setattr(bench, "leg_report", lambda *a, **k: {})
    monkeypatch.setattr(bench, "measured", lambda fn: (fn(), {}))

    def leg(results):
        it, warnings = iter(results), []
        monkeypatch.setattr(bench, "run_workload", lambda *a, **k: next(it))
        return bench.as_shipped_flags_leg("config4", 2500, 500, False, timed, lambda what, fn: fn(), warnings), warnings
    o0, w = leg([run(705000.0), run(712000.0), run(690000.0)])
    assert o0["ratio_to_timed_run"] == 0.94 and o0["exact"] and w == []
    o0, w = leg([run(150000.0), run(140000.0), run(705000.0)])                    # two stalled repetitions: median 150 k
    assert o0["ratio_to_timed_run"] == 0.2 and len(w) == 1 and w[0].startswith("-O0 build/reference delivered-rate ratio 0.20 outside [0.85, 1.05]")
    assert "150,000 / 140,000 / 705,000" in w[0] and "stalls above" in w[0] and bench._warning_kind(w[0]) == "ratio"
    o0, w = leg([run(705000.0), run(712000.0, exact=False), run(690000.0)])
    assert o0["exact"] is False and len(w) == 1 and "cpu_baseline_O0.exact is false" in w[0] and bench._warning_kind(w[0]) == "ratio"


def test_line_keeps_one_warning_of_each_kind_before_a_second_of_any():
    """every stalled repetition adds its own ~500-character warning; six of those used to fill the line's six
    slots and push the roofline / ratio / client-bound ones out.  Order is kept, lengths are cut, the count is whole."""
    sys.path.insert(0, str(REPO))
    import bench
    stall = [f"config2 repetition {k}: harness stall: " + "y" * 600 for k in range(7)]
    ws = stall[:5] + ["roofline: the talker ran at only 0.800 ...", "probe: open-loop leg median ...",
                      "restatement/reference delivered-rate ratio 0.80 outside [0.9, 1.1] ...",
                      "8 replica(s) under a 16-core quota leave 1 receiver thread each: client-bound by ...",
                      "config4: repetitions 1 / 2 lines/s spread x2.00 with the talker >= 1.00 busy in each: the core was slower, ..."] + stall[5:]
    kept = bench.compact_warnings(ws)
    assert len(kept) == 6 and [bench._warning_kind(w) for w in kept] == ["stall", "roofline", "probe", "ratio", "client-bound", "slow-core"]
    assert kept[0] == stall[0][:397] + "..." and all(len(w) <= 400 for w in kept)
    assert all(len(w) <= 160 for w in bench.compact_warnings(ws, tight=1))
    assert bench.compact_warnings(stall) == [w[:397] + "..." for w in stall[:6]]          # one kind only: the first six, in order
    assert bench.compact_warnings([]) == []
    full = json.loads((REPO / "profiles" / "bench_r03_driverargs_run4_final_mi355xhost.json").read_text())
    line = bench.compact_line({**full, "warnings": ws}, None)
    assert line["warnings"] == kept and line["warnings_count"] == len(ws) == 12


def _probe_leg(walls, ratios, la=9.8):
    reps = [{"written_lines_per_s_wall": w, "cpu_ns_per_line": 1e12 / w * r, "wall_ns_per_line": 1e12 / w} for w, r in zip(walls, ratios)]
    med = sorted(reps, key=lambda r: r["written_lines_per_s_wall"])[len(reps) // 2]
    return {**med, "written_lines_per_s_wall_all": walls, "cpu_over_wall_all": ratios, "loadavg_before": [la, 9.0, 8.0]}


def test_probe_legs_explain_themselves_in_the_line():
    """VERDICT r4 item 2, on the driver's own r04 numbers: open loop 536,647 / 535,873 / 547,691 against closed loop 780,537 /
    781,169 / 676,750 with `warnings: []` and nothing in the line to tell a waiting probe thread from a slow core.  Both
    readings of that record, stubbed: the warning names the leg and which reading it was."""
    sys.path.insert(0, str(REPO))
    import bench
    opened, closed = [536647, 535873, 547691], [780537, 781169, 676750]

    def roofline(open_ratios, closed_ratios):
        return {"probe": {"full_open": _probe_leg(opened, open_ratios), "full_closed": _probe_leg(closed, closed_ratios)}}
    # (1) the probing thread of the open loop was on its core 0.69 of the time: it waited or was descheduled
    w = bench.probe_warnings(roofline([0.69, 0.69, 0.70], [0.99, 0.99, 0.86]))
    assert len(w) == 2 and all(x.startswith("probe:") for x in w)
    assert "open-loop leg median 536,647" in w[0] and "0.69 x the closed-loop median 780,537" in w[0]
    assert "only 0.69 of the wall clock" in w[0] and "waited" in w[0] and "descheduled" in w[0]
    assert "closed-loop leg repetitions 780,537 / 781,169 / 676,750" in w[1] and "spread x1.15" in w[1] and "[0.99, 0.99, 0.86]" in w[1]
    # (2) same rates, thread busy throughout: the core itself ran slower
    w = bench.probe_warnings(roofline([0.99, 0.99, 0.99], [0.99, 0.99, 0.99]))
    assert len(w) == 2 and "busy 0.99 of the wall clock: the core itself ran slower" in w[0] and "open-loop" in w[0]
    # a clean record (round 4's builder runs: the two legs within 3 %) says nothing
    clean = {"probe": {"full_open": _probe_leg([782436, 767391, 789645], [0.99] * 3), "full_closed": _probe_leg([771281, 769337, 771759], [0.99] * 3)}}
    assert bench.probe_warnings(clean) == [] and bench.probe_warnings(None) == []
    # and the line carries the three fields per leg, inside the budget, on a committed full record of the real shape
    full = json.loads((REPO / "profiles" / "bench_r04_driverargs_run3_final_mi355xhost_full.json").read_text())
    rf = roofline([0.69, 0.69, 0.70], [0.99, 0.99, 0.86])
    full["roofline"]["probe"]["full_open"].update(rf["probe"]["full_open"])
    full["roofline"]["probe"]["full_closed"].update(rf["probe"]["full_closed"])
    full["warnings"] = bench.probe_warnings(full["roofline"])
    full["cpu_baseline_O0"] = {"value": 722017.0, "unit": "lines/s", "cores": 1, "kind": "reference", "binary": "oracle/_ref/nuts333_O0",
                               "flags": "gcc, no -O flag", "sample": "x", "reps": 3, "exact": True, "rate_all_reps": [722017.0, 725000.1, 719000.2],
                               "ratio_to_timed_run": 0.96, "server_cpu_us_per_written_line": 1.37, "busy_all_reps": [1.0, 1.0, 1.0]}
    line = bench.compact_line(full, "gpurun_out/bench_full_n1.json")
    assert line["roofline"]["probe_legs"]["open"] == {"wall_all": opened, "cpu_over_wall": 0.69, "loadavg_before": 9.8}
    assert line["roofline"]["probe_legs"]["closed"] == {"wall_all": closed, "cpu_over_wall": 0.99, "loadavg_before": 9.8}
    assert line["cpu_baseline_O0"]["rate_all_reps"] == [722017.0, 725000.1, 719000.2] and line["cpu_baseline_O0"]["ratio_to_timed_run"] == 0.96
    assert line["warnings_count"] == 2 and line["warnings"][0].startswith("probe: open-loop")
    text = bench.render_line(full, "gpurun_out/bench_full_n1.json")          # what main() prints: tight form if need be
    assert len(text) <= bench.LINE_BUDGET, len(text)
    printed = json.loads(text)
    assert printed["roofline"]["probe_legs"] == line["roofline"]["probe_legs"] and printed["warnings_count"] == 2
    assert "open-loop leg median 536,647" in printed["warnings"][0] and "0.69 x" in printed["warnings"][0]
    assert len(bench.render_line({**full, "warnings": []}, "gpurun_out/bench_full_n1.json")) <= bench.LINE_BUDGET - 150


def test_a_reference_build_that_went_missing_fails_instead_of_skipping(monkeypatch, tmp_path):
    """build() leaves oracle/_build/ref_built.json when it compiled oracle/_ref/.  Marker present and a
    binary absent (or changed) = an error everywhere the reference would be used: pick_binary() raises instead of
    headlining the restatement, and the `ref_binary` fixture / test_gpu_box._binary fail instead of skipping."""
    import hashlib
    from nuts333_amd import talker
    monkeypatch.setattr(talker, "REPO", tmp_path)
    monkeypatch.setattr(talker, "REF_BINARY", tmp_path / "oracle" / "_ref" / "nuts333")
    monkeypatch.setattr(workloads, "REF_BINARY", tmp_path / "oracle" / "_ref" / "nuts333")
    (tmp_path / "oracle" / "_build").mkdir(parents=True)
    (tmp_path / "oracle" / "_ref").mkdir()
    monkeypatch.delenv("NUTS_REQUIRE_REFERENCE", raising=False)
    monkeypatch.setattr(talker, "KFD_NODE", tmp_path / "no-such-device-node")
    assert talker.reference_expected_but_missing() is None          # no marker: a machine without /root/reference, skips are honest
    # marker AND binaries lost together (a snapshot that dropped every ignored file).  The expectation then comes
    # from outside the artefacts: the box scripts' env flag, or the machine being a GPU box (device node, no reference tree).
    monkeypatch.setenv("NUTS_REQUIRE_REFERENCE", "1")
    assert "lost its prebuilt artefacts" in talker.reference_expected_but_missing() and "NUTS_REQUIRE_REFERENCE=1" in talker.reference_expected_but_missing()
    with pytest.raises(FileNotFoundError, match="lost its prebuilt artefacts"):
        workloads.pick_binary("auto")
    monkeypatch.delenv("NUTS_REQUIRE_REFERENCE")
    (tmp_path / "kfd").write_text("")
    monkeypatch.setattr(talker, "KFD_NODE", tmp_path / "kfd")
    monkeypatch.setattr(talker, "REFERENCE_TREE", tmp_path / "no-reference-tree-here")
    assert "a GPU box" in talker.reference_expected_but_missing()
    monkeypatch.setenv("NUTS_REQUIRE_REFERENCE", "0")                 # waived by name
    assert talker.reference_expected_but_missing() is None and workloads.pick_binary("auto")[1] == "port"
    monkeypatch.delenv("NUTS_REQUIRE_REFERENCE")
    monkeypatch.setattr(talker, "REFERENCE_TREE", tmp_path)          # a GPU machine that has the sources can build: nothing demanded
    assert talker.reference_expected_but_missing() is None
    monkeypatch.setattr(talker, "KFD_NODE", tmp_path / "no-such-device-node")
    blob = b"\x7fELF stand-in bytes"
    talker.ref_marker().write_text(json.dumps({"sha256": {"nuts333": hashlib.sha256(blob).hexdigest()}}))
    assert "nuts333 was built for this snapshot, but it is missing" in talker.reference_expected_but_missing()
    for kind in ("auto", "reference"):
        with pytest.raises(FileNotFoundError, match="missing"):
            workloads.pick_binary(kind)
    assert workloads.pick_binary("port")[1] == "port"          # asking for the restatement by name stays possible
    (tmp_path / "oracle" / "_ref" / "nuts333").write_bytes(blob)
    assert talker.reference_expected_but_missing() is None
    (tmp_path / "oracle" / "_ref" / "nuts333").write_bytes(blob + b"!")
    assert "differs from the build recorded" in talker.reference_expected_but_missing()


def test_every_worker_reports_its_counters_and_its_own_window(port_binary):
    """a receiver thread that got no loop turn between the last delivery and the stop used to report
    cpu_s = 0 (a fabricated 'busy 0.00'); it now samples on its way out, over the window it really covered."""
    res = workloads.config2(lines=300, warmup=10, binary=port_binary)
    assert res["spin"] == 1 and res["workers"]
    for w in res["workers"]:
        assert w["window_s"] >= res["wall_s"] * 0.5 and w["cpu_s"] > 0 and 0 < w["busy_frac"] <= 1.05, w


def test_placement_tops_up_a_set_too_small_to_hold_the_harness(monkeypatch):
    """a 2-core / 4-thread host.  One thread per core would leave talker + 1 receiver (client-bound) and
    nothing for config #5's second talker; the set is topped up with the sibling threads, quietest first -- except the
    talker's own sibling (CPU 3 here, talker on CPU 1), which goes last however quiet it is."""
    from nuts333_amd import placement
    cpus = [0, 1, 2, 3]
    monkeypatch.setattr(placement.os, "sched_getaffinity", lambda _pid: set(cpus))
    monkeypatch.setattr(placement, "topology", lambda allowed: ({c: 0 for c in cpus}, {0: 0, 1: 1, 2: 0, 3: 1}))
    monkeypatch.setattr(placement, "busy_sample", lambda interval=0.25: {0: 0.0, 1: 0.0, 2: 0.5, 3: 0.0})
    monkeypatch.delenv("NUTS_BENCH_CPUS", raising=False)
    got = placement.choose()
    assert got["policy"] == "quiet" and got["sets"] == [[1, 0, 2, 3]] and "topped up with 2" in got["note"]


def test_run_reports_where_the_talkers_wall_clock_went(port_binary):
    res = workloads.config2(lines=400, warmup=20, binary=port_binary)
    d = workloads.leg_diagnostics(res)
    assert 0 < d["server_busy_frac"] <= 1.05 and 0 <= d["server_run_delay_frac"] < 1 and d["wall_s"] == round(res["wall_s"], 4)
    assert d["placement"]["talker_cpus"] and set(d["placement"]["talker_cpus"]).isdisjoint(d["placement"]["receiver_cpus"])
    assert [w["senders"] for w in res["workers"]].count(1) == 1 and sum(w["clients"] for w in res["workers"]) == 10
    assert d["progress_gaps"]["threshold_ms"] == 5.0 and d["slow_acks"]["count"] >= 0


def test_placement_picks_the_quietest_l3_group_and_one_thread_per_core(monkeypatch):
    """The MI355X box in miniature: 2 L3 groups x 4 cores x 2 threads (siblings at +8); another tenant sits on CPU 0 and
    on CPU 9 (the sibling of core 1).  Rounds 1-2 pinned the talker to CPU 0."""
    from nuts333_amd import placement
    cpus = list(range(16))
    monkeypatch.setattr(placement.os, "sched_getaffinity", lambda _pid: set(cpus))
    l3 = {c: (0 if c % 8 < 4 else 4) for c in cpus}
    core = {c: c % 8 for c in cpus}
    monkeypatch.setattr(placement, "topology", lambda allowed: (l3, core))
    monkeypatch.setattr(placement, "busy_sample", lambda interval=0.25: {c: (1.0 if c in (0, 9) else 0.0) for c in cpus})
    monkeypatch.delenv("NUTS_BENCH_CPUS", raising=False)
    got = placement.choose(4)
    assert got["policy"] == "quiet" and got["sets"] == [[4, 5, 6, 7]]
    two = placement.choose(4, groups=2)["sets"]
    assert two[0] == [4, 5, 6, 7] and two[1][:2] == [2, 3] and set(two[1]) == {0, 1, 2, 3}    # busy cores last
    monkeypatch.setenv("NUTS_BENCH_CPUS", "first")
    assert placement.choose(4)["sets"] == [cpus] and placement.choose(4)["policy"] == "first"
    monkeypatch.delenv("NUTS_BENCH_CPUS")
    monkeypatch.setattr(placement, "topology", lambda allowed: None)          # a VM without cache topology in sysfs
    assert placement.choose(4)["sets"] == [cpus]


def test_write_only_probe_leg_cannot_hang():
    """the write-only leg's reader used to answer every ack with an input line nobody read, with a
    blocking send(); past the socket buffers it stopped draining and the talker thread spun forever (reproduced with
    these arguments: 1,000,000 rounds x 60 B)."""
    out = subprocess.run([str(workloads.LOADGEN_BIN), "--probe-line", "67", "0", "1000000", "0", "0", "1"], check=True,
                         stdout=subprocess.PIPE, timeout=120).stdout
    j = json.loads(out)
    assert j["bytes_ok"] and j["rounds"] == 1000000 and j["select_read"] == 0


def test_bench_default_headline_is_the_largest_configuration():
    """Driver default: BASELINE configs[3] -- 1000 clients, .shout.  Headline leg only here."""
    j = _bench("--steps", "2", "--warmup", "1", "--binary", "port", "--no-extras")
    assert j["config"]["baseline_config"] == "config4" and j["config"]["clients"] == 1000
    assert j["config"]["lines_per_step"] == 100 and j["delivered"] == j["expected_delivered"] == 2 * 100 * 999
    assert j["server_syscalls"]["per_input_line"] == {"read": 1.0, "write": 1000.0}
    assert j["roofline"] is None and "configs" not in j


def test_bench_two_replicas_gloo():
    """N > 1 = N independent replicas, launched the way the driver launches them (torch.distributed.run)."""
    port = free_ports(1)[0]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(REPO / "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--lines-per-step", "10", "--binary", "port"],
                         check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600).stdout.decode()
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    j = json.loads(lines[0])
    assert j["config"]["baseline_config"] == "config4"
    assert j["n_gpus"] == 2 and j["delivered"] == j["expected_delivered"] == 2 * (2 * 10 * 999)
    assert j["roofline"] is None            # probes run at N=1 only


def test_bench_self_launch_of_replicas():
    j = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--lines-per-step", "200", "--binary", "port", "--workload", "config2")
    assert j["n_gpus"] == 2 and j["delivered"] == 2 * 200 * 9


def test_bench_failed_replica_ends_the_job_instead_of_hanging(tmp_path):
    """a rank whose run fails must take the others down through the first collective, not leave them in
    gloo until its 30-minute timeout.  Rank 1 is given a load generator that cannot run."""
    port = free_ports(1)[0]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if r == 1:
            env["NUTS_BENCH_INJECT_FAILURE"] = "1"
        procs.append(subprocess.Popen([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                                       "--lines-per-step", "50", "--binary", "port", "--workload", "config2"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=120) for p in procs]
    assert [p.returncode for p in procs] == [1, 1]
    assert b"injected failure" in outs[1][1] and b"another replica failed" in outs[0][1]
    assert not outs[0][0].strip()            # no result line from a job that did not complete
