"""Run one transcript scenario (tests/scenarios.py) against given talker binaries.

Shared by tests/golden/make_golden.py (reference build -> fixtures) and the parity tests
(restatement, and the reference again where it is present).  A scenario is either
single-talker -- ``(config kwargs, accounts, script)`` -- or a dict describing several
talkers joined by netlinks; in that case ``binaries`` may name a different binary per
talker, which is how the restatement is tested against the real reference across the wire.
"""
from __future__ import annotations

import tempfile
from dataclasses import asdict
from pathlib import Path
from typing import Sequence

from nuts333_amd import provision as pv
from nuts333_amd.talker import Talker, free_ports
from nuts333_amd.transcript import Peer, Session

import scenarios


def build_writelog_shim(directory: Path) -> Path:
    """Compile tests/preload_writelog.c (an LD_PRELOAD logger of write(2) sizes on sockets) into ``directory``."""
    import subprocess
    shim = Path(directory) / "writelog.so"
    subprocess.run(["gcc", "-O2", "-fPIC", "-shared", str(Path(__file__).resolve().parent / "preload_writelog.c"),
                    "-o", str(shim), "-ldl"], check=True)
    return shim


def run_scenario(name: str, binaries: Path | Sequence[Path], writelog_shim: Path | None = None) -> dict:
    spec = (scenarios.SCENARIOS.get(name) or scenarios.REFERENCE_ONLY[name])()
    if isinstance(spec, tuple):
        cfg_kw, accounts, script = spec
        spec = {
            "configs": lambda p: [pv.TalkerConfig(mainport=p[0][0], wizport=p[0][1], linkport=p[0][2],
                                                  **{"max_users": 50, **cfg_kw})],
            "accounts": [accounts], "boot_order": [0], "script": script, "config_kw": cfg_kw,
        }
    n = len(spec["accounts"])
    if isinstance(binaries, (str, Path)):
        binaries = [Path(binaries)] * n
    assert len(binaries) == n, f"scenario {name} needs {n} binaries"
    talkers: list[Talker | None] = [None] * n
    with tempfile.TemporaryDirectory(prefix=f"scn_{name}_") as tmp:
        ports = [free_ports(3) for _ in range(n)]
        sess = Session(ports[0][0], talker_ports=[p[0] for p in ports])
        sess.link_ports = [p[2] for p in ports]
        sess.wiz_ports = [p[1] for p in ports]
        # scripted netlink peers listen before any talker boots (a talker with auto_connect dials at boot)
        for key in spec.get("peers", []):
            sess.peers[key] = Peer(key)
        cfgs = spec["configs"](ports, {k: p.port for k, p in sess.peers.items()}) if spec.get("peers") else spec["configs"](ports)
        try:
            wlog = Path(tmp) / "writes.log"
            shim_env = {"LD_PRELOAD": str(writelog_shim), "WRITELOG": str(wlog)} if writelog_shim is not None else None
            for i in spec["boot_order"]:
                root = Path(tmp) / f"t{i}"
                pv.write_tree(root, cfgs[i], spec["accounts"][i])
                for rel, content in spec.get("files", {}).items():
                    (root / rel).write_text(content)
                talkers[i] = Talker(binaries[i], root, extra_env=shim_env)
                talkers[i].start()
            for i, needle in spec.get("wait_syslog", []):
                talkers[i].wait_syslog(needle)
            spec["script"](sess)
            # every client has just completed a .version round trip: all writes the script caused are done.  Read the
            # log NOW -- what the talkers write while the clients are being closed below is a race with their SIGKILL.
            write_sizes = None
            if writelog_shim is not None:
                by_pid: dict[int, list[int]] = {}
                for ln in (wlog.read_text().splitlines() if wlog.exists() else []):
                    pid, _fd, n = ln.split()
                    by_pid.setdefault(int(pid), []).append(int(n))
                write_sizes = [by_pid.get(t.pid, []) for t in talkers]
            alive = [t.alive() for t in talkers]
            sess.shutdown()
            # on-disk side effects the scenario wants pinned (user records written at logout)
            files = {}
            for rel in spec.get("collect_files", []):
                path = Path(tmp) / "t0" / rel
                deadline = __import__("time").monotonic() + 5
                while not path.exists() and __import__("time").monotonic() < deadline:
                    __import__("time").sleep(0.02)
                __import__("time").sleep(0.1)
                files[rel] = scenarios.mask_file(rel, path.read_bytes().decode("latin-1")) if path.exists() else None
        finally:
            sess.shutdown()
            for t in talkers:
                if t is not None:
                    t.stop()
    if not all(alive):
        raise RuntimeError(f"a talker died during scenario {name}: alive={alive}")
    return {
        "scenario": name,
        "config": spec.get("config_kw", {}),
        "accounts": [[asdict(a) for a in accs] for accs in spec["accounts"]],
        "steps": sess.steps,
        **({"files": files} if spec.get("collect_files") else {}),
        **({"write_sizes": write_sizes} if writelog_shim is not None else {}),
    }
