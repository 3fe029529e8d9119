"""BASELINE configurations #1 and #5 *as BASELINE.json words them*: on the reference's own shipped files.

    "Boot talker with datafiles/config, 1 local telnet client, .say in lounge"
    "Two-server netlink (datafiles/config + config2) with cross-link .shout traffic"

Every other run in this repository boots a tree that ``nuts333_amd/provision.py`` generates (same room graph, our
own texts), because no reference file may travel to the GPU box.  This module is the container-only cross-check:
it populates a temporary directory AT TEST TIME from ``/root/reference`` -- ``datafiles/config``, ``config2``, the
room descriptions, ``userfiles/Fred.D``, ``motd1``/``motd2`` -- boots ``oracle/_ref/nuts333`` there and drives it
with the same load generator.  Nothing is copied into the repository, committed or shipped; callers are
``reference``-marked tests and ``tools/shipped_rows.py`` (which prints the BASELINE.md rows).

What the shipped files force on the run (all measured behaviour of the reference):
* fixed ports 7000-7002 and 5000-5002 (``datafiles/config:5-7``, ``config2:8-10``);
* the only account is ``Fred`` (GOD, password ``test``, **prompt on**, ``userfiles/Fred.D:2``): every input line
  costs two ``write_user`` calls, the acknowledgement and the prompt (``nuts333.c:218-219, 2174-2197``);
* ``config2`` does not boot as shipped -- ``logging YES`` is not a 3.3.3 option (``datafiles/config2:11``,
  ``nuts333.c:599-607``): line 11 is rewritten to ``system_logging ON`` in the temporary copy, nowhere else;
* on the second talker the listener is a NEW-level account the run creates through the ordinary new-user dialogue
  (``nuts333.c:1552-1587``): NEW users may listen but not shout (``nuts333.h:206-209``), so the cross-link shouts
  come from Fred, who travels from talker 1 (``.go talker2``) and shouts as a remote user of talker 2.
"""
from __future__ import annotations

import os
import shutil
import socket
import tempfile
import time
from pathlib import Path

from nuts333_amd import workloads
from nuts333_amd.talker import REF_BINARY, Talker

REFERENCE = Path(os.environ.get("REFERENCE", "/root/reference"))
PORTS_T1 = (7000, 7001, 7002)     # datafiles/config:5-7
PORTS_T2 = (5000, 5001, 5002)     # datafiles/config2:8-10


def available() -> bool:
    return (REFERENCE / "datafiles" / "config").exists() and REF_BINARY.exists()


def ports_free(ports) -> bool:
    for p in ports:
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", p))
            except OSError:
                return False
    return True


def populate(root: Path, fix_config2: bool = False) -> Path:
    """A talker working directory made of the reference's shipped files (nuts333.h:3-14)."""
    for d in ("datafiles", "userfiles", "helpfiles", "mailspool"):
        (root / d).mkdir(parents=True, exist_ok=True)
    for f in (REFERENCE / "datafiles").iterdir():
        shutil.copy(f, root / "datafiles" / f.name)
    shutil.copy(REFERENCE / "userfiles" / "Fred.D", root / "userfiles" / "Fred.D")
    for f in ("motd1", "motd2"):
        shutil.copy(REFERENCE / f, root / f)
    if fix_config2:
        path = root / "datafiles" / "config2"
        lines = path.read_text().split("\n")
        assert lines[10].split() == ["logging", "YES"], lines[10]          # datafiles/config2:11
        lines[10] = "system_logging    ON"
        path.write_text("\n".join(lines))
    return root


def _cpu(pin: bool, k: int):
    cpus = workloads.host_cpus()
    return cpus[k] if pin and len(cpus) > k + 1 else None


def config1_shipped(lines: int = 10_000, warmup: int = 500, pin: bool = True) -> dict:
    """Shipped ``datafiles/config``; ``fred``/``test``; ``.go lounge``; closed-loop ``say``."""
    tmp = Path(tempfile.mkdtemp(prefix="nuts333_shipped_"))
    t = None
    try:
        populate(tmp)
        t = Talker(REF_BINARY, tmp, config_name="config", cpu=_cpu(pin, 0))
        t.start()
        spec = workloads.Spec()
        c = spec.add_client("fred", PORTS_T1[0])                # the name is typed in lower case (motd1:4)
        spec.add_pre(c, ".go lounge")
        for i in range(-warmup, lines):
            spec.add_line(c, workloads.payload(i % 1_000_000), [], warm=i < 0, self_lines=2)
        res = workloads.run_spec(spec, [t], timeout_s=300, pin=pin)
        res.pop("per_client_lines", None)
        res["workload"] = f"config1 on the shipped files: datafiles/config, fred/test, .go lounge, {lines} say lines"
        res["server_alive_after"] = t.alive()
        return res
    finally:
        if t is not None:
            t.stop()
        shutil.rmtree(tmp, ignore_errors=True)


def _create_new_user(port: int, name: str, password: str = "test") -> None:
    """The ordinary new-account dialogue (nuts333.c:1552-1587); leaves by closing the socket."""
    s = socket.create_connection(("127.0.0.1", port), timeout=10)
    buf = b""

    def until(needle: bytes) -> None:
        nonlocal buf
        deadline = time.monotonic() + 10
        while needle not in buf:
            if time.monotonic() > deadline:
                raise TimeoutError(f"{needle!r} never came; got {buf[-200:]!r}")
            chunk = s.recv(4096)
            if not chunk:
                raise ConnectionError(f"closed while waiting for {needle!r}; got {buf[-200:]!r}")
            buf += chunk
        buf = b""

    until(b"Give me a name: ")
    s.sendall(name.encode() + b"\n")
    until(b"Give me a password: ")
    s.sendall(password.encode() + b"\n")
    until(b"confirm password: ")
    s.sendall(password.encode() + b"\n")
    until(b"has been set yet.")
    s.close()
    time.sleep(0.2)


def config5_shipped(lines: int = 200, pin: bool = True) -> dict:
    """Shipped ``config`` + ``config2`` (line 11 fixed in the temporary copy); Fred travels and shouts across the link."""
    tmp = Path(tempfile.mkdtemp(prefix="nuts333_shipped_nl_"))
    t1 = t2 = None
    try:
        populate(tmp / "t2", fix_config2=True)
        populate(tmp / "t1")
        t2 = Talker(REF_BINARY, tmp / "t2", config_name="config2", cpu=_cpu(pin, 0))
        t2.start()
        t1 = Talker(REF_BINARY, tmp / "t1", config_name="config", cpu=_cpu(pin, 1))
        t1.start()                                               # auto_connect YES: dials talker2 at boot (config:14,46)
        t1.wait_syslog("Connection to talker2 verified")
        _create_new_user(PORTS_T2[0], "Listener")
        spec = workloads.Spec()
        fred = spec.add_client("fred", PORTS_T1[0])
        listener = spec.add_client("Listener", PORTS_T2[0])
        spec.add_pre(fred, ".go talker2")                         # drive is the CONNECT room (config:35)
        for k in range(lines):
            spec.add_line(fred, ".shout " + workloads.payload(k), [listener], self_lines=2)   # ack + prompt (via PRM)
        res = workloads.run_spec(spec, [t1, t2], timeout_s=300, pin=pin, threads=2)
        per_client = res.pop("per_client_lines")
        s1, s2 = res["servers"]
        res["netlink"] = {"writes_t1_to_t2": s1["write_syscalls"] - per_client[fred],      # ACT frames
                          "writes_t2_to_t1": s2["write_syscalls"] - 2 * per_client[listener],  # colour_def ON: 2 writes/line
                          "expected_act_frames": lines, "expected_msg_frames": lines, "expected_prm_frames": lines}
        res["workload"] = (f"config5 on the shipped files: datafiles/config + config2 (line 11 fixed), Fred travels to "
                           f"talker2 and shouts {lines} lines, one NEW-level listener there")
        res["servers_alive_after"] = [t1.alive(), t2.alive()]
        return res
    finally:
        for t in (t1, t2):
            if t is not None:
                t.stop()
        shutil.rmtree(tmp, ignore_errors=True)
