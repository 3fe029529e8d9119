"""Boot / stop a talker process (the reference build or our restatement) in a scratch tree.

Launch rules, each a measured reference behaviour:

* the talker daemonises itself -- the parent forks, sleeps one second and exits
  (``nuts333.c:79-83``) -- so the process we start is not the one that serves; the
  daemon's PID is taken from the boot line the child appends to ``./syslog``
  (``nuts333.c:86-87``, ``1434-1444``);
* stdout/stderr go to a file, never a pipe (the daemon keeps the descriptors open);
* argv[0] is kept short: the reference does ``strcpy(progname,argv[0])`` into a 40-byte
  global (``nuts333.c:62``, ``nuts333.h:283``), so starting it by a long absolute path
  overflows -- the -O2 build aborts with "buffer overflow detected" (FORTIFY), the -O0 build
  silently corrupts ``confile``.  Found on the GPU box, whose checkout path is long;
* it is stopped with SIGKILL once every client socket is closed: the SIGTERM path runs
  ``talker_shutdown`` which walks freed list nodes (``nuts333.c:4044``).
"""
from __future__ import annotations

import os
import re
import signal
import socket
import subprocess
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
REF_BINARY = REPO / "oracle" / "_ref" / "nuts333"
REF_BINARY_O0 = REPO / "oracle" / "_ref" / "nuts333_O0"
PORT_BINARY = REPO / "oracle" / "_build" / "talker_port"

_BOOT_RE = re.compile(r"Booted successfully with PID (\d+)")


_HANDED_OUT: set[int] = set()


def free_ports(n: int = 3) -> list[int]:
    """n distinct currently-free loopback TCP ports.  Under a multi-rank launch (RANK set) each rank
    draws from its own 400-port window, so replicas booted at the same instant cannot pick the same
    port between our probe and the talker's bind()."""
    rank = os.environ.get("RANK")
    if rank is not None and rank.isdigit():
        base = 20000 + (int(rank) % 100) * 400
        ports: list[int] = []
        start = int.from_bytes(os.urandom(2), "little") % 400
        for k in range(400):
            p = base + (start + k) % 400
            with socket.socket() as s:
                try:
                    s.bind(("127.0.0.1", p))
                except OSError:
                    continue
            ports.append(p)
            if len(ports) == n:
                return ports
    # Not bind(port 0): that hands out EPHEMERAL ports, the range every client connection of the suite draws its source
    # port from -- between our probe and the talker's bind() a scripted client or one of a load generator's 1000 sockets can
    # be given the very port (seen once in ~200 boots as "Can't bind to main port: Address already in use"; the talker sets
    # SO_REUSEADDR, nuts333.c:1183, but an auto-bound client socket does not).  Draw from a window below the ephemeral range.
    lo, hi = port_window()
    ports = []
    start = int.from_bytes(os.urandom(4), "little") % (hi - lo)
    for k in range(hi - lo):
        p = lo + (start + k) % (hi - lo)
        if p in _HANDED_OUT:          # promised to an earlier caller of this process that may not have bound it yet
            continue
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", p))
            except OSError:
                continue
        ports.append(p)
        if len(ports) == n:
            if len(_HANDED_OUT) > 2000:
                _HANDED_OUT.clear()
            _HANDED_OUT.update(ports)
            return ports
    raise RuntimeError(f"no {n} free TCP ports in {lo}-{hi}")


def port_window(ephemeral: tuple[int, int] | None = None) -> tuple[int, int]:
    """[lo, hi) to draw talker ports from: outside the kernel's ephemeral range ``ip_local_port_range`` (``ephemeral``
    overrides the /proc reading, for the unit test).  Normally 12000 up to the range's low end (at most 20000); on a host
    whose range starts at or below ~13000 (a common tuning is ``1024 65535``) the window below it is too small or empty
    (a modulus by a non-positive span), so take the room above its top end, and if the range covers
    everything, 12000-20000 inside it: a collision with a client's source port is then possible again, as it was with
    bind(0), but nothing divides by zero and _HANDED_OUT still keeps this process's own draws apart."""
    if ephemeral is None:
        try:
            a, b = Path("/proc/sys/net/ipv4/ip_local_port_range").read_text().split()[:2]
            ephemeral = (int(a), int(b))
        except (OSError, ValueError):
            ephemeral = (32768, 60999)
    e_lo, e_hi = ephemeral
    lo, hi = 12000, min(20000, e_lo)
    if hi - lo >= 1000:
        return lo, hi
    if 65535 - e_hi >= 1000:
        return e_hi + 1, min(65536, e_hi + 1 + 8000)
    return 12000, 20000


def ref_marker() -> Path:
    """Written by ``__graft_entry__.build()`` when it compiled oracle/_ref/ (so: in the container, where /root/reference
    exists); travels to the GPU box beside the binaries (both git-ignored, neither gpurun-ignored).  Holds the sha256 of
    each reference binary built.  Kept outside oracle/_ref/ so that a snapshot that lost that directory still says so."""
    return REPO / "oracle" / "_build" / "ref_built.json"


#: where the reference's sources are when they are anywhere (the build container); never present on the GPU box
REFERENCE_TREE = Path("/root/reference")
#: the kernel's compute device node: present on a GPU box (reading the path touches no GPU), absent in the build container
KFD_NODE = Path("/dev/kfd")


def reference_required() -> str | None:
    """the marker and the binaries it vouches for are both git-ignored artefacts and travel together; a snapshot
    that drops ignored files wholesale loses both, and the absence of the marker then reads as "a machine that never had the
    reference" -- skips with rc 0, the restatement in the headline.  So the expectation also lives where the artefacts do
    not: ``NUTS_REQUIRE_REFERENCE=1`` (set by tools/box_r06.sh) demands the reference binary, ``=0`` waives it, and with
    neither set a machine that HAS a GPU device node and has NO /root/reference is taken to be the GPU box, which can only
    ever run the reference if the snapshot carried it.  Returns why it is required, or None."""
    env = os.environ.get("NUTS_REQUIRE_REFERENCE", "").strip()
    if env == "0":
        return None
    if env:
        return f"NUTS_REQUIRE_REFERENCE={env} is set"
    if KFD_NODE.exists() and not REFERENCE_TREE.exists():
        return (f"this machine has {KFD_NODE} and no {REFERENCE_TREE} (a GPU box: the snapshot is expected to carry the prebuilt "
                f"oracle/_ref/; NUTS_REQUIRE_REFERENCE=0 waives this)")
    return None


def reference_expected_but_missing() -> str | None:
    """the marker says a reference build was made for this snapshot; if a binary it names is absent
    or differs, say so -- callers FAIL on it instead of skipping (tests) or headlining the restatement (bench.py).
    Without a marker the same holds when ``reference_required()`` says the reference must be here."""
    import hashlib
    import json
    m = ref_marker()
    if not m.exists():
        why = reference_required()
        if why and not REF_BINARY.exists():
            return (f"oracle/_ref/{REF_BINARY.name} is missing, and so is the marker oracle/_build/ref_built.json, but {why}: the snapshot "
                    f"lost its prebuilt artefacts -- rebuild with __graft_entry__.build() where /root/reference exists")
        return None
    try:
        want = json.loads(m.read_text())["sha256"]
    except (OSError, ValueError, KeyError) as e:
        return f"{m} unreadable: {e!r}"
    for name, digest in want.items():
        b = REPO / "oracle" / "_ref" / name
        if not b.exists():
            return f"oracle/_build/ref_built.json says oracle/_ref/{name} was built for this snapshot, but it is missing"
        if hashlib.sha256(b.read_bytes()).hexdigest() != digest:
            return f"oracle/_ref/{name} differs from the build recorded in oracle/_build/ref_built.json"
    return None


class Talker:
    def __init__(self, binary: os.PathLike | str, root: os.PathLike | str, config_name: str = "config",
                 cpu: int | None = None, tz: str = "UTC", burn_fds: int = 0, extra_env: dict[str, str] | None = None):
        self.binary = Path(binary)
        self.root = Path(root)
        self.config_name = config_name
        self.cpu = cpu
        self.tz = tz
        self.burn_fds = burn_fds     # tests only: start the daemon with this many descriptors already in use
        # tests only: variables for THIS talker's environment (e.g. the LD_PRELOAD write-size logger).  Merged into the
        # copy handed to Popen; os.environ is never touched, so a profiler's own LD_PRELOAD and every other child of the
        # calling process are left alone.  An LD_PRELOAD given here is put in front of an inherited one.
        self.extra_env = dict(extra_env or {})
        self.pid: int | None = None

    # -- lifecycle -------------------------------------------------------------------
    def start(self, timeout: float = 15.0) -> int:
        if not self.binary.exists():
            raise FileNotFoundError(
                f"{self.binary} is not built; run `make -C oracle` (reference) or `python -c "
                "'import __graft_entry__ as g; g.build()'`")
        syslog = self.root / "syslog"
        if syslog.exists():
            syslog.unlink()
        env = dict(os.environ, TZ=self.tz)
        for k, v in self.extra_env.items():
            env[k] = f"{v}:{env[k]}" if k == "LD_PRELOAD" and env.get(k) else v
        out = open(self.root / "boot.log", "wb")

        # No preexec_fn (unsafe when the parent has threads, e.g. after torch was imported): the new session comes
        # from start_new_session, and the CPU pin is inherited -- the calling thread narrows its own affinity
        # around the fork and restores it.  The daemon the launcher forks (nuts333.c:79) inherits it in turn.
        burn = [os.open(os.devnull, os.O_RDONLY) for _ in range(self.burn_fds)]
        saved_affinity = None
        try:
            if self.cpu is not None:
                try:
                    saved_affinity = os.sched_getaffinity(0)
                    os.sched_setaffinity(0, {self.cpu})
                except OSError:
                    saved_affinity = None
            launcher = subprocess.Popen([self.binary.name[:30], self.config_name], executable=str(self.binary),
                                        cwd=self.root, stdin=subprocess.DEVNULL, pass_fds=burn,
                                        stdout=out, stderr=subprocess.STDOUT, env=env, start_new_session=True)
        finally:
            if saved_affinity is not None:
                os.sched_setaffinity(0, saved_affinity)
            out.close()
            for fd in burn:
                os.close(fd)
        deadline = time.monotonic() + timeout
        rc = None
        while time.monotonic() < deadline:
            if syslog.exists():
                m = _BOOT_RE.search(syslog.read_text(errors="replace"))
                if m:
                    self.pid = int(m.group(1))
                    break
            rc = launcher.poll()
            if rc not in (None, 0):
                break
            time.sleep(0.02)
        if self.pid is None:
            launcher.kill() if launcher.poll() is None else None
            log = (self.root / "boot.log").read_text(errors="replace")
            raise RuntimeError(f"talker did not boot (launcher rc={rc}):\n{log}")
        # reap the launcher (it exits one second after fork; do not wait for it inline)
        self._launcher = launcher
        return self.pid

    def alive(self) -> bool:
        if self.pid is None:
            return False
        try:
            os.kill(self.pid, 0)
        except ProcessLookupError:
            return False
        try:
            stat = Path(f"/proc/{self.pid}/stat").read_text()
            return stat.rsplit(")", 1)[1].split()[0] != "Z"
        except OSError:
            return False

    def stop(self) -> None:
        if self.pid is not None:
            try:
                os.kill(self.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            for _ in range(200):
                if not self.alive():
                    break
                time.sleep(0.01)
            self.pid = None
        launcher = getattr(self, "_launcher", None)
        if launcher is not None:
            try:
                launcher.wait(timeout=3)
            except subprocess.TimeoutExpired:
                launcher.kill()
                launcher.wait()
            self._launcher = None

    def __enter__(self) -> "Talker":
        self.start()
        return self

    def __exit__(self, *exc) -> None:
        self.stop()

    def wait_syslog(self, needle: str, timeout: float = 10.0) -> None:
        """Block until ./syslog contains ``needle`` (e.g. the netlink 'verified' line, nuts333.c:3428)."""
        deadline = time.monotonic() + timeout
        path = self.root / "syslog"
        while time.monotonic() < deadline:
            if path.exists() and needle in path.read_text(errors="replace"):
                return
            time.sleep(0.02)
        raise TimeoutError(f"{needle!r} never appeared in {path}")

    # -- /proc sampling --------------------------------------------------------------
    def cpu_times(self) -> tuple[float, float]:
        """(user_s, sys_s) of the daemon from /proc/<pid>/stat fields 14/15."""
        stat = Path(f"/proc/{self.pid}/stat").read_text()
        f = stat.rsplit(")", 1)[1].split()
        hz = os.sysconf("SC_CLK_TCK")
        return int(f[11]) / hz, int(f[12]) / hz

    def rss_peak_kb(self) -> int:
        for line in Path(f"/proc/{self.pid}/status").read_text().splitlines():
            if line.startswith("VmHWM:"):
                return int(line.split()[1])
        return 0
