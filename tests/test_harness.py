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
    res = workloads.config5(lines=30, binary=port_binary)
    _check(res)
    assert res["deliveries"] == 2 * 30 * 14 and len(res["servers"]) == 2
    assert res["netlink_frames_t2_to_t1"] == 300


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
    _check(workloads.config5(lines=30, binary=ref_binary), "config5")


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
    assert json.loads(out)["writes_per_s"] > 0


# ---------------------------------------------------------------- bench.py contract
CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _bench(*args, env=None):
    out = subprocess.run([sys.executable, str(REPO / "bench.py"), *args], check=True, stdout=subprocess.PIPE,
                         env={**os.environ, **(env or {})}, timeout=600).stdout.decode().strip().splitlines()
    assert len(out) == 1, out
    return json.loads(out[0])


def test_bench_single_replica_contract():
    j = _bench("--steps", "2", "--warmup", "1", "--lines-per-step", "200", "--binary", "port")
    assert CONTRACT_KEYS <= set(j)
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["gpu_used"] is False
    assert j["delivered"] == j["expected_delivered"] == 2 * 200 * 9
    assert j["vs_baseline"] is None and j["scaling"] == "weak" and j["higher_is_better"] is True
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] == 1
    assert j["roofline"]["bound"] == "host-syscall" and 0 < j["roofline"]["frac"] < 2 and j["roofline"]["traffic"] is None
    assert "model" not in j["config"] and "workload" in j["config"]


def test_bench_two_replicas_gloo():
    """N > 1 = N independent replicas, launched the way the driver launches them (torch.distributed.run)."""
    port = free_ports(1)[0]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(REPO / "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--lines-per-step", "200", "--binary", "port"],
                         check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600).stdout.decode()
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["delivered"] == j["expected_delivered"] == 2 * (2 * 200 * 9)
    assert j["roofline"] is None            # probes run at N=1 only


def test_bench_self_launch_of_replicas():
    j = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--lines-per-step", "200", "--binary", "port")
    assert j["n_gpus"] == 2 and j["delivered"] == 2 * 200 * 9
