"""The five BASELINE.json configurations as concrete, deterministic workloads.

Each ``configN`` function provisions a scratch tree (``provision.py``), boots a talker
(``talker.py``), writes a spec for the compiled load generator (``loadgen/loadgen.c``),
runs it and returns one result dict.  The definitions follow SURVEY.md section 8(d):

1. 1 client ``Fred`` (GOD), ``.go lounge``, closed-loop ``say``             -- plumbing
2. 10 level-1 clients left in ``drive``; client 0 says M lines               -- 9 recipients/line
3. 100 clients, 20 per room over the 5 rooms; seeded (333) mix of
   70 % say / 20 % .shout / 10 % .tell, every client sends                   -- mixed
4. 1000 level-1 clients in ``drive``; client 0 ``.shout``s M lines            -- 999 recipients/line
5. two talkers joined by a netlink; 10 locals each, 5 of talker1's users
   ``.go talker2``; one remote and one local user ``.shout``                  -- MSG/EMSG relay

The metric (BASELINE.json) is *delivered lines per second* = lines written to recipients
other than the sender, divided by the wall time of the timed phase; the sender's own
acknowledgement lines (``You say: ...``) are counted separately as ``acks``.
"""
from __future__ import annotations

import json
import os
import random
import shutil
import subprocess
import tempfile
from pathlib import Path
from typing import Callable, Sequence

from . import placement
from . import provision as pv
from .talker import PORT_BINARY, REF_BINARY, Talker, free_ports, reference_expected_but_missing

HERE = Path(__file__).resolve().parent
LOADGEN_SRC = HERE / "loadgen" / "loadgen.c"
LOADGEN_BIN = HERE / "loadgen" / "loadgen"

LOOK_END = r"has been set yet.*\n\r"     # '*': any bytes (a colour user gets ESC[0m before the newline)

#: 54-byte payload with a 6-digit sequence number, no '~', no leading command
#: character, none of the three filtered words (nuts333.h:275-277).
PAYLOAD_FMT = "synthetic broadcast line {:06d} from the nuts333 bench"
PAYLOAD_LEN = 54


def payload(seq: int) -> str:
    s = PAYLOAD_FMT.format(seq % 1_000_000)
    assert len(s) == PAYLOAD_LEN, len(s)
    return s


def build_loadgen(force: bool = False) -> Path:
    if force or not LOADGEN_BIN.exists() or LOADGEN_BIN.stat().st_mtime < LOADGEN_SRC.stat().st_mtime:
        subprocess.run(["gcc", "-O2", "-Wall", "-Wextra", "-pthread", str(LOADGEN_SRC), "-o", str(LOADGEN_BIN)],
                       check=True)
    return LOADGEN_BIN


def host_cpus() -> list[int]:
    """The cores this process places talkers and receivers on, the talker's first (see placement.py)."""
    return placement.ordered_cpus()


#: upper bound on busy-polling receiver threads per load-generator process; bench.py lowers it when several replicas
#: share one cgroup CPU quota (see cgroup_cpu_quota)
MAX_CLIENT_THREADS = 4


def cgroup_cpu_quota() -> float | None:
    """CPU-time quota of this container in cores (cgroup v2 ``cpu.max``), or None when unlimited / unknown.  The
    one-GPU MI355X box allows 16 cores' worth of CPU time although all 256 CPUs are schedulable: a talker plus four
    spinning receivers per replica fits twice, not four times -- the fourth replica gets throttled, which would be
    a property of the quota, not of the talker."""
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        return None if quota == "max" else int(quota) / int(period)
    except (OSError, ValueError):
        return None


def cgroup_throttled() -> tuple[int, int] | None:
    """(nr_throttled, throttled_usec) of this container so far, or None."""
    try:
        d = dict(l.split() for l in Path("/sys/fs/cgroup/cpu.stat").read_text().splitlines())
        return int(d["nr_throttled"]), int(d["throttled_usec"])
    except (OSError, ValueError, KeyError):
        return None


def pick_binary(kind: str = "auto") -> tuple[Path, str]:
    """('reference' | 'port' | 'auto') -> (binary path, kind actually used)."""
    if kind == "port_fast":
        # the restatement with INTEGRATION.md section 3's CPU-side changes switched on (same bytes on the wire)
        os.environ["NUTS_PORT_FAST"] = "1"
        if not PORT_BINARY.exists():
            raise FileNotFoundError("oracle/_build/talker_port is not built")
        return PORT_BINARY, "port_fast"
    os.environ.pop("NUTS_PORT_FAST", None)
    if kind in ("reference", "auto"):
        lost = reference_expected_but_missing()
        if lost:          # never fall back to the restatement in silence when the reference was built for this snapshot
            raise FileNotFoundError(lost)
    if kind in ("reference", "auto") and REF_BINARY.exists():
        return REF_BINARY, "reference"
    if kind == "reference":
        raise FileNotFoundError(f"{REF_BINARY} missing: build it with `make -C oracle ref` where /root/reference exists")
    if PORT_BINARY.exists():
        return PORT_BINARY, "port"
    raise FileNotFoundError("neither oracle/_ref/nuts333 nor oracle/_build/talker_port is built; run __graft_entry__.build()")


# --------------------------------------------------------------------------- spec / run
class Spec:
    def __init__(self) -> None:
        self.clients: list[tuple[str, str, str, int]] = []
        self.pre: list[tuple[int, str, str]] = []
        self.lines: list[tuple[int, str]] = []
        self.warm: list[tuple[int, str]] = []
        self.expect_warm_lines = 0
        self.expect_lines = 0
        self.expected_deliveries = 0
        self.expected_self_extra = 0          # lines a sender gets besides its ack (its prompt line): not deliveries
        self.expected_per_client: list[int] = []

    def add_client(self, name: str, port: int, host: str = "127.0.0.1", password: str = pv.PASSWORD) -> int:
        self.clients.append((name, password, host, port))
        self.expected_per_client.append(0)
        return len(self.clients) - 1

    def add_pre(self, idx: int, line: str, expect: str = LOOK_END) -> None:
        self.pre.append((idx, expect, line))

    def add_line(self, sender: int, text: str, recipients: Sequence[int], warm: bool = False, self_lines: int = 1) -> None:
        """One input line; ``recipients`` are the client indices that must receive one line each.
        ``warm`` lines run in an untimed phase before the timed one.  ``self_lines`` is what the sender itself
        gets back per input line: 1 = the acknowledgement ("You say: ..."); 2 when the account has its prompt on
        (one more write_user from prompt(), nuts333.c:2174-2197)."""
        assert "\n" not in text and "\t" not in text and len(text) < 900
        if warm:
            self.warm.append((sender, text))
            self.expect_warm_lines += self_lines + len(recipients)
            return
        self.lines.append((sender, text))
        self.expected_per_client[sender] += self_lines   # the acknowledgement (+ the prompt)
        for r in recipients:
            assert r != sender
            self.expected_per_client[r] += 1
        self.expect_lines += self_lines + len(recipients)
        self.expected_deliveries += len(recipients)
        self.expected_self_extra += self_lines - 1

    def render(self, server_pids: Sequence[int], threads: int, cpus: Sequence[int], timeout_s: float,
               login_window: int, spin: bool = True, quickack: bool = True, drain_quiet_ms: int = 40) -> str:
        out = [f"threads {threads}", f"login_window {login_window}", f"timeout_s {timeout_s}",
               f"expect_lines {self.expect_lines}", f"expect_warm_lines {self.expect_warm_lines}",
               f"spin {int(spin)}", f"quickack {int(quickack)}", f"drain_quiet_ms {drain_quiet_ms}"]
        if cpus:
            out.append("cpus " + ",".join(str(c) for c in cpus))
        out += [f"server_pid {p}" for p in server_pids]
        out += [f"client {n} {p} {h} {port}" for n, p, h, port in self.clients]
        out += [f"pre {i} {e}\t{l}" for i, e, l in self.pre]
        out += [f"warm {i} {t}" for i, t in self.warm]
        out += [f"line {i} {t}" for i, t in self.lines]
        return "\n".join(out) + "\n"


def run_spec(spec: Spec, talkers: Sequence[Talker], *, timeout_s: float = 600.0, pin: bool = True,
             threads: int | None = None) -> dict:
    build_loadgen()
    cpus = host_cpus()
    server_cpus = {t.cpu for t in talkers if t.cpu is not None}
    client_cpus = [c for c in cpus if c not in server_cpus] if pin else []
    if threads is None:
        # Few receiver threads, on cores next to the talker's: with loopback TCP the talker's write(2)
        # touches cache lines last used by the receiving core, so spreading receivers over many cores
        # (or a second CCD) inflates the *server's* CPU per line by ~45 % (1.38 -> 2.0+ us on an EPYC
        # 9575F, profiles/sweep_loadgen_threads_r01_mi355xhost.log).  2-7 threads measure the same; 1 is
        # client-bound.  4 is the portable choice.
        threads = max(1, min(len(client_cpus) or len(cpus) - 1 or 1, MAX_CLIENT_THREADS, len(spec.clients)))
    # the talker listens with backlog 10 (nuts333.c:1189): keep concurrent logins below it
    login_window = max(1, 8 // threads)
    # talker<->talker traffic sits behind Nagle + delayed ACK (neither side sets TCP_NODELAY): wait longer
    quiet = 40 if len(talkers) == 1 else 400
    text = spec.render([t.pid for t in talkers], threads, client_cpus, timeout_s, login_window, drain_quiet_ms=quiet)
    proc = subprocess.run([str(LOADGEN_BIN)], input=text.encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=timeout_s + 60)
    err = proc.stderr.decode(errors="replace")[-2000:]
    try:
        res = json.loads(proc.stdout.decode().strip().splitlines()[-1])
    except (IndexError, json.JSONDecodeError):
        raise RuntimeError(f"loadgen produced no result (rc={proc.returncode}): {err}")
    if proc.returncode != 0 or not res.get("ok"):
        raise RuntimeError(f"loadgen failed (rc={proc.returncode}): {err} {res}")
    res["expected_deliveries"] = spec.expected_deliveries
    res["deliveries"] -= spec.expected_self_extra        # the load generator counts "lines that are not acks"
    res["self_extra_lines"] = spec.expected_self_extra
    res["per_client_exact"] = res["per_client_lines"] == spec.expected_per_client
    res["placement"] = {"talker_cpus": sorted(server_cpus), "receiver_cpus": client_cpus[:threads]}
    res["exact"] = bool(res["per_client_exact"] and res["lines_total"] == spec.expect_lines
                        and res["deliveries"] == spec.expected_deliveries)
    return summarise(res)


def summarise(res: dict) -> dict:
    wall = res["wall_s"]
    deliv = res["deliveries"]
    res["delivered_lines_per_s"] = deliv / wall if wall > 0 else 0.0
    res["input_lines_per_s"] = res["input_lines"] / wall if wall > 0 else 0.0
    res["bytes_per_line"] = res["bytes_total"] / res["lines_total"] if res["lines_total"] else 0.0
    for s in res["servers"]:
        written = res["lines_total"]          # every line, acks included, costs the server one write_user
        s["cpu_us_per_written_line"] = s["cpu_ns"] / 1e3 / written if written else 0.0
        tot = s["utime_s"] + s["stime_s"]
        s["user_frac"] = s["utime_s"] / tot if tot > 0 else None
        s["busy_frac"] = s["cpu_ns"] / 1e9 / wall if wall > 0 else 0.0
        s["write_syscalls_per_line"] = s.get("write_syscalls", 0) / written if written else 0.0
        s["read_syscalls_per_input_line"] = s.get("read_syscalls", 0) / res["input_lines"] if res["input_lines"] else 0.0
        # where the talker's wall clock went: on the CPU, runnable but kept off it, or asleep (select / a full socket)
        s["run_delay_frac"] = s.get("run_delay_ns", 0) / 1e9 / wall if wall > 0 else 0.0
        s["sleep_frac"] = max(0.0, 1.0 - s["busy_frac"] - s["run_delay_frac"])
    for w in res.get("workers", []):
        # each worker's own sampling window (it ends a loop turn after the run's, later still for a descheduled one)
        span = w.get("window_s") or wall
        w["busy_frac"] = w["cpu_s"] / span if span > 0 else 0.0
    return res


def leg_diagnostics(res: dict) -> dict:
    """The fields that let one run explain its own wall clock: the talker's time split into
    on-CPU / runnable-but-waiting / asleep, how long no line moved anywhere, how slow the slowest acknowledgements
    were, and whether the (busy-polling) receiver threads were themselves kept off their cores."""
    srv = res["servers"][0]
    workers = res.get("workers", [])
    sender_w = [w for w in workers if w.get("senders")] or workers
    return {"wall_s": round(res["wall_s"], 4),
            "server_busy_frac": round(srv["busy_frac"], 3),
            # busy 1.0 and still slow = the core itself was slower (a neighbour on the sibling thread or in the L3): it
            # shows here, in the cost of a line, not in the scheduler's counters
            "server_cpu_us_per_written_line": round(srv["cpu_us_per_written_line"], 3),
            "server_run_delay_frac": round(srv["run_delay_frac"], 4),
            "server_sleep_frac": round(srv["sleep_frac"], 3),
            "server_involuntary_switches": srv.get("involuntary_switches"),
            "server_voluntary_switches": srv.get("voluntary_switches"),
            "ack_latency_us": {k: res["ack_latency_us"][k] for k in ("p50", "p99", "max")},
            "slow_acks": res.get("slow_acks"), "progress_gaps": res.get("progress_gaps"),
            "receiver_busy_frac_min": round(min((w["busy_frac"] for w in workers), default=0.0), 3),
            "sender_receiver_busy_frac": round(min((w["busy_frac"] for w in sender_w), default=0.0), 3),
            "sender_receiver_run_delay_s": round(max((w["run_delay_s"] for w in sender_w), default=0.0), 4),
            "placement": res.get("placement")}


def attribute_stall(res: dict, *, saturating: bool = True, throttled_periods: int | None = None,
                    throttled_ms: float | None = None) -> str | None:
    """None when the talker was the bottleneck, as a saturating configuration (#2-#4) expects; otherwise one sentence
    saying where its idle time went, from the run's own counters.  The cgroup's throttle counters are taken around the
    WHOLE leg (boot, logins, drain, warm-up, timed window), so they are blamed first only when they can plausibly account
    for the window's idle time -- throttled for at least half of it, or the talker itself shows run-queue wait; otherwise
    they follow the counter-based attribution as context."""
    srv = res["servers"][0]
    idle = 1.0 - srv["busy_frac"]
    if not saturating or idle < 0.10:
        return None
    d = leg_diagnostics(res)
    head = f"harness stall: talker busy only {srv['busy_frac']:.2f} of {res['wall_s']:.2f} s wall"
    gaps = d["progress_gaps"] or {}
    tail = (f"; no line moved anywhere for {gaps.get('total_s', 0):.2f} s in {gaps.get('count', 0)} gaps >= {gaps.get('threshold_ms', 5):.0f} ms "
            f"(longest {gaps.get('max_ms', 0):.0f} ms); ack p50/max {d['ack_latency_us']['p50']:.0f}/{d['ack_latency_us']['max']:.0f} us")
    if throttled_periods:
        quota = f"cgroup CPU quota throttled this container in {throttled_periods} periods" + (
            f" ({throttled_ms:.0f} ms)" if throttled_ms is not None else "")
        if srv["run_delay_frac"] > 0.01 or (throttled_ms is not None and throttled_ms / 1e3 >= idle * res["wall_s"] / 2):
            return f"{head} -- {quota}{tail}"
        tail += f"; context: {quota} somewhere in the leg (boot and logins included), too little to explain the window"
    if srv["run_delay_frac"] > idle / 2:
        return (f"{head} -- it was RUNNABLE but off its core for {srv['run_delay_frac']:.2f} of the wall "
                f"({srv.get('involuntary_switches')} involuntary switches): another tenant on CPU {d['placement']['talker_cpus']}{tail}")
    # (workers only busy-poll under `spin 1`: with spin 0 a low busy fraction is the design, not a symptom)
    if res.get("spin", 1) and d["sender_receiver_busy_frac"] < 0.90:
        return (f"{head} -- it was ASLEEP in select() while the sender's busy-polling receiver thread got only "
                f"{d['sender_receiver_busy_frac']:.2f} of its core (run-queue wait {d['sender_receiver_run_delay_s']:.2f} s): "
                f"the closed loop waited for the client, not the talker{tail}")
    return f"{head} -- asleep for {srv['sleep_frac']:.2f} of the wall with its receivers running: unattributed{tail}"


# --------------------------------------------------------------------------- configs
def _boot(root: Path, cfg: pv.TalkerConfig, accounts, binary: Path, cpu: int | None) -> Talker:
    pv.write_tree(root, cfg, accounts)
    t = Talker(binary, root, cpu=cpu)
    t.start()
    return t


def _server_cpu(pin: bool, k: int = 0) -> int | None:
    if not pin:
        return None
    cpus = host_cpus()
    return cpus[k] if len(cpus) > k + 1 else None


def _run_single(build: Callable[[Spec, int], None], accounts, *, binary: Path, pin: bool, workdir: Path | None,
                timeout_s: float, max_users: int = 1100, colour: int = 0, rooms=pv.DEFAULT_ROOMS) -> dict:
    tmp = Path(tempfile.mkdtemp(prefix="nuts333_", dir=workdir))
    talker = None
    try:
        ports = free_ports(3)
        cfg = pv.TalkerConfig(mainport=ports[0], wizport=ports[1], linkport=ports[2], max_users=max_users, rooms=rooms)
        talker = _boot(tmp, cfg, accounts, binary, _server_cpu(pin))
        spec = Spec()
        build(spec, ports[0])
        res = run_spec(spec, [talker], timeout_s=timeout_s, pin=pin)
        res.pop("per_client_lines", None)
        res["server_rss_peak_kb"] = talker.rss_peak_kb()
        res["server_alive_after"] = talker.alive()
        return res
    finally:
        if talker is not None:
            talker.stop()
        shutil.rmtree(tmp, ignore_errors=True)


def config1(lines: int = 10_000, *, warmup: int = 0, prompt: int = 0, binary: Path, pin: bool = True, workdir=None,
            timeout_s: float = 300.0) -> dict:
    """1 client, ``.go lounge``, closed-loop say; nobody else hears it (plumbing / latency).  ``prompt=1`` gives the
    account the flags of the shipped ``userfiles/Fred.D`` (prompt on, character echo on): two writes per input line."""
    accounts = [pv.Account("Fred", level=4, desc="the GOD account", prompt=prompt, charmode_echo=prompt)]

    def build(spec: Spec, port: int) -> None:
        c = spec.add_client("Fred", port)
        spec.add_pre(c, ".go lounge")          # GOD teleports (nuts333.c:4399-4404)
        for i in range(-warmup, lines):
            spec.add_line(c, payload(i % 1_000_000), [], warm=i < 0, self_lines=1 + prompt)

    res = _run_single(build, accounts, binary=binary, pin=pin, workdir=workdir, timeout_s=timeout_s)
    res["workload"] = f"config1: 1 client, {lines} say lines in lounge"
    return res


def config2(lines: int = 20_000, n: int = 10, *, colour: int = 0, all_send: bool = False, warmup: int = 0, binary: Path,
            pin: bool = True, workdir=None, timeout_s: float = 300.0) -> dict:
    """n clients in ``drive``; client 0 (or everyone, ``all_send``) says ``lines`` lines."""
    accounts = [pv.Account(pv.bot_name(i), level=1, colour=colour) for i in range(n)]

    def build(spec: Spec, port: int) -> None:
        ids = [spec.add_client(pv.bot_name(i), port) for i in range(n)]
        senders = ids if all_send else ids[:1]
        per = lines // len(senders)
        for k in range(-(warmup // len(senders)), per):
            for s in senders:
                spec.add_line(s, payload(k % 1_000_000), [r for r in ids if r != s], warm=k < 0)

    res = _run_single(build, accounts, binary=binary, pin=pin, workdir=workdir, timeout_s=timeout_s, max_users=n + 10)
    res["workload"] = f"config2: {n} clients in one room, {lines} say lines, {'all send' if all_send else 'client 0 sends'}, colour {colour}"
    return res


def config3(per_client: int = 200, n: int = 100, *, seed: int = 333, six_rooms: bool = False, warmup: int = 0, binary: Path,
            pin: bool = True, workdir=None, timeout_s: float = 600.0) -> dict:
    """n clients spread evenly over the rooms (5 as shipped, or 6 with the ``shop``); seeded 70/20/10
    say/.shout/.tell mix from every client."""
    room_set = pv.SIX_ROOMS if six_rooms else pv.DEFAULT_ROOMS
    rooms = [r.name for r in room_set]
    room_of = [rooms[i % len(rooms)] for i in range(n)]
    accounts = [pv.Account(pv.bot_name(i), level=2 if room_of[i] == "wizroom" else 1) for i in range(n)]
    rng = random.Random(seed)

    def build(spec: Spec, port: int) -> None:
        ids = [spec.add_client(pv.bot_name(i), port) for i in range(n)]
        for i in ids:
            for hop in pv.WALKS[room_of[i]]:
                spec.add_pre(i, f".go {hop}")
        members = {r: [i for i in ids if room_of[i] == r] for r in rooms}
        # ``warmup`` untimed rounds first, drawn from their own generator so that the timed schedule (and with it
        # the expected delivery count) is the same with and without a warm-up
        for rnd, gen in [(-1 - k, random.Random(seed + 1 + k)) for k in range(warmup)] + [(k, rng) for k in range(per_client)]:
            for s in ids:
                x = gen.random()
                text = payload((rnd % 10_000) * n + s)
                if x < 0.70:
                    spec.add_line(s, text, [r for r in members[room_of[s]] if r != s], warm=rnd < 0)
                elif x < 0.90:
                    spec.add_line(s, ".shout " + text, [r for r in ids if r != s], warm=rnd < 0)
                else:
                    tgt = gen.choice([r for r in ids if r != s])
                    spec.add_line(s, f".tell {pv.bot_name(tgt)} {text}", [tgt], warm=rnd < 0)

    res = _run_single(build, accounts, binary=binary, pin=pin, workdir=workdir, timeout_s=timeout_s, max_users=n + 10,
                      rooms=room_set)
    res["workload"] = f"config3: {n} clients over {len(rooms)} rooms, {per_client} lines each, 70/20/10 say/shout/tell, seed {seed}"
    return res


def config4(lines: int = 1000, n: int = 1000, *, colour: int = 0, warmup: int = 0, binary: Path, pin: bool = True, workdir=None,
            timeout_s: float = 900.0) -> dict:
    """n clients in ``drive``; client 0 ``.shout``s ``lines`` lines (n-1 recipients each)."""
    accounts = [pv.Account(pv.bot_name(i), level=1, colour=colour) for i in range(n)]

    def build(spec: Spec, port: int) -> None:
        ids = [spec.add_client(pv.bot_name(i), port) for i in range(n)]
        for k in range(-warmup, lines):
            spec.add_line(ids[0], ".shout " + payload(k % 1_000_000), ids[1:], warm=k < 0)

    res = _run_single(build, accounts, binary=binary, pin=pin, workdir=workdir, timeout_s=timeout_s, max_users=n + 10)
    res["workload"] = f"config4: {n} clients, client 0 shouts {lines} lines, colour {colour}"
    return res


def config5(lines: int = 1000, locals_each: int = 10, travellers: int = 5, *, warmup: int = 0, tap: bool = False,
            binary: Path, pin: bool = True, workdir=None, timeout_s: float = 300.0) -> dict:
    """Two talkers joined by a netlink; a remote and a local user of talker2 shout concurrently.

    Frames on the link are MEASURED, two independent ways (nuts333.c:1302-1305, 3801, 2181):
    * always: each talker's write(2) count from /proc/<pid>/io minus the lines its own clients received
      = writes that went to the link socket (one per frame: write_sock, nuts333.c:1281-1286);
    * with ``tap=True``: a counting relay on the link (``linktap.LinkTap``) that parses both byte streams and
      counts frames by verb, the timed ones recognised by the payload phrase.  Not used in timed runs.
    """
    from .linktap import LinkTap
    tmp = Path(tempfile.mkdtemp(prefix="nuts333_nl_", dir=workdir))
    t1 = t2 = link = None
    try:
        p1, p2 = free_ports(3), free_ports(3)
        names1 = [pv.bot_name(i) for i in range(locals_each)]
        names2 = [pv.bot_name(locals_each + i) for i in range(locals_each)]
        if tap:
            link = LinkTap(p2[2], marker=b"synthetic broadcast line")
        # talker2 accepts in its lounge (ACCEPT room); it must know talker1's site string as
        # reverse-resolved by gethostbyaddr (nuts333.c:322, 2908-2909): list both spellings
        cfg2 = pv.TalkerConfig(mainport=p2[0], wizport=p2[1], linkport=p2[2], verification="fred123x",
                               sites=[pv.Site("talker1", "localhost", p1[2], "bloggs456x"),
                                      pv.Site("talker1", "127.0.0.1", p1[2], "bloggs456x")])
        rooms1 = tuple(pv.Room(r.label, r.name, r.links, r.access or ("PUB" if r.name == "drive" else ""),
                               "CONNECT talker2" if r.name == "drive" else ("" if r.netlink == "ACCEPT" else r.netlink),
                               r.description) for r in pv.DEFAULT_ROOMS)
        cfg1 = pv.TalkerConfig(mainport=p1[0], wizport=p1[1], linkport=p1[2], verification="bloggs456x",
                               auto_connect=True, rooms=rooms1,
                               sites=[pv.Site("talker2", "127.0.0.1", link.port if link else p2[2], "fred123x")])
        t2 = _boot(tmp / "t2", cfg2, [pv.Account(n) for n in names2], binary, _server_cpu(pin, 0))
        t1 = _boot(tmp / "t1", cfg1, [pv.Account(n) for n in names1], binary, _server_cpu(pin, 1))
        t1.wait_syslog("Connection to talker2 verified")
        spec = Spec()
        ids1 = [spec.add_client(n, p1[0]) for n in names1]
        ids2 = [spec.add_client(n, p2[0]) for n in names2]
        gone = ids1[:travellers]
        for i in gone:
            spec.add_pre(i, ".go talker2")        # TRANS -> GRANTED -> ACT look -> MSG frames back
        on_t2 = ids2 + gone
        remote_sender, local_sender = gone[0], ids2[0]
        for k in range(-warmup, lines):
            for s in (remote_sender, local_sender):
                spec.add_line(s, ".shout " + payload(k % 1_000_000), [r for r in on_t2 if r != s], warm=k < 0)
        res = run_spec(spec, [t1, t2], timeout_s=timeout_s, pin=pin, threads=2)
        res["workload"] = (f"config5: 2 talkers, {locals_each} locals each, {travellers} of talker1's users on talker2, "
                           f"1 remote + 1 local sender x {lines} shouts")
        # measured: write(2) calls of each talker that did NOT end in one of its own clients' sockets went to the link
        per_client = res.pop("per_client_lines")
        rx1, rx2 = sum(per_client[i] for i in ids1), sum(per_client[i] for i in ids2)
        s1, s2 = res["servers"]
        res["netlink"] = {
            "method": "/proc/<pid>/io write(2) count of each talker minus the lines its own clients received",
            "writes_t1_to_t2": s1["write_syscalls"] - rx1,        # ACT <name> .shout ...        (nuts333.c:3801)
            "writes_t2_to_t1": s2["write_syscalls"] - rx2,        # MSG..EMSG frames + PRM <name>  (c:1302-1305, 2181)
            # what the protocol says those must be: every line that reaches a travelling user crossed the link as ONE
            # MSG..EMSG frame; every relayed command is ONE ACT frame and is answered by ONE PRM frame
            "expected_act_frames": lines,
            "expected_msg_frames": 2 * lines * (len(gone) - 1) + 2 * lines,
            "expected_prm_frames": lines,
        }
        res["netlink"]["exact"] = (res["netlink"]["writes_t1_to_t2"] == res["netlink"]["expected_act_frames"] and
                                   res["netlink"]["writes_t2_to_t1"] == res["netlink"]["expected_msg_frames"]
                                   + res["netlink"]["expected_prm_frames"])
        if link:
            res["netlink"]["tap"] = link.snapshot()
        res["servers_alive_after"] = [t1.alive(), t2.alive()]
        return res
    finally:
        for t in (t1, t2):
            if t is not None:
                t.stop()
        if link:
            link.close()
        shutil.rmtree(tmp, ignore_errors=True)


def probe_write(nbytes: int = 67, count: int = 300_000) -> dict:
    build_loadgen()
    out = subprocess.run([str(LOADGEN_BIN), "--probe-write", str(nbytes), str(count)], check=True,
                         stdout=subprocess.PIPE).stdout
    return json.loads(out)
