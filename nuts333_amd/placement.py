"""Which cores the talker and the synthetic clients run on.

The talker is single-threaded and the measurement is its CPU time against the wall clock, so where it sits matters
more than anything else the harness does (round 1: receivers on another CCD inflate the talker's ``write(2)`` by 45 %;
round 2: a neighbour on the talker's core moves a run by 5-30 %).  Rounds 1-2 pinned the talker to the FIRST CPU
of the affinity set -- CPU 0 on the MI355X box, whose 256 CPUs are schedulable by every tenant of the host and where
every other harness that "pins to the first core" lands too.  The one stalled leg of round 2's driver run (the
restatement at equal CPU per line and 5.3x the wall clock) had nothing in its record to say
whether the talker was runnable and kept off its core or asleep waiting for a descheduled sender.  (Measured afterwards
on the box, 12 + 12 interleaved runs: CPU 0 is not systematically worse -- profiles/stall_hunt_r03_*; this is a precaution.)

So: sample ``/proc/stat`` for half a second, group the allowed CPUs by shared L3 (one CCD: the talker and
its receivers must share it), drop SMT siblings, and take the quietest cores of the quietest group -- talker on the
quietest one.  The choice and the idleness it was based on go into the result line.  ``NUTS_BENCH_CPUS=first``
restores the old rule; no sysfs topology (a VM, a container without /sys) falls back to it as well.
"""
from __future__ import annotations

import os
import time
from pathlib import Path

_SYS = Path("/sys/devices/system/cpu")
#: what one replica can put to use: 1 talker + 4 receiver threads (workloads.MAX_CLIENT_THREADS) + config #5's second talker
NEED_CPUS = 6
_cached: dict | None = None


def _parse_list(text: str) -> list[int]:
    out: list[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def _stat() -> dict[int, tuple[int, int]]:
    """cpu -> (busy jiffies, total jiffies)"""
    out = {}
    for line in Path("/proc/stat").read_text().splitlines():
        if line.startswith("cpu") and line[3].isdigit():
            f = line.split()
            v = [int(x) for x in f[1:9]]              # user nice system idle iowait irq softirq steal
            out[int(f[0][3:])] = (sum(v) - v[3] - v[4], sum(v))
    return out


def busy_sample(interval: float = 0.25) -> dict[int, float]:
    """Fraction of ``interval`` each CPU spent not idle, host-wide (other tenants included)."""
    a = _stat()
    time.sleep(interval)
    b = _stat()
    return {c: ((b[c][0] - a[c][0]) / max(1, b[c][1] - a[c][1])) for c in b if c in a}


def topology(cpus: list[int]) -> tuple[dict[int, int], dict[int, int]] | None:
    """(cpu -> L3 group id, cpu -> physical core id) for the given CPUs, or None when sysfs does not say."""
    l3: dict[int, int] = {}
    core: dict[int, int] = {}
    try:
        for c in cpus:
            shared = _parse_list((_SYS / f"cpu{c}/cache/index3/shared_cpu_list").read_text())
            sibs = _parse_list((_SYS / f"cpu{c}/topology/thread_siblings_list").read_text())
            l3[c] = min(shared)
            core[c] = min(sibs)
    except (OSError, ValueError):
        return None
    return l3, core


def choose(n_wanted: int = 8, *, groups: int = 1, interval: float = 0.5) -> dict:
    """Pick ``groups`` disjoint sets of up to ``n_wanted`` cores.  Returns
    ``{"policy", "sets": [[talker_cpu, client_cpu, ...], ...], "busy_before": {cpu: frac}, "note"}``."""
    allowed = sorted(os.sched_getaffinity(0))
    policy = os.environ.get("NUTS_BENCH_CPUS", "quiet")
    # the old rule: affinity order, cut into equal slices of at most n_wanted CPUs when several replicas share the host;
    # with fewer than two CPUs per replica nothing can be kept apart and every replica gets them all
    per = min(n_wanted, len(allowed) // groups)
    first = {"policy": "first", "busy_before": {}, "note": "affinity order",
             "sets": [allowed] * groups if groups == 1 or per < 2 else [allowed[g * per:(g + 1) * per] for g in range(groups)]}
    if policy == "first" or per < 2:
        return first
    topo = topology(allowed)
    if topo is None:
        first["note"] = "no sysfs cache topology: affinity order"
        return first
    l3, core = topo
    busy = busy_sample(interval)
    # one entry per physical core: its first allowed thread, busy = the busier of its threads (a busy sibling
    # costs the talker almost as much as a busy core)
    cores: dict[int, dict] = {}
    for c in allowed:
        e = cores.setdefault(core[c], {"cpu": c, "l3": l3[c], "busy": 0.0})
        e["busy"] = max(e["busy"], busy.get(c, 0.0))
    by_l3: dict[int, list[dict]] = {}
    for e in cores.values():
        by_l3.setdefault(e["l3"], []).append(e)
    # a group's score = the mean busyness of the n quietest cores it can offer; groups too small to hold a talker
    # and one receiver are useless
    scored = []
    for g, es in by_l3.items():
        es.sort(key=lambda e: (round(e["busy"], 2), e["cpu"]))
        take = es[:n_wanted]
        if len(take) >= 2:
            scored.append((sum(e["busy"] for e in take) / len(take), -len(take), g, take))
    if len(scored) < groups:
        first["note"] = "fewer L3 groups than replicas: affinity order"
        return first
    scored.sort(key=lambda t: (round(t[0], 2), t[1], t[2]))
    sets = [[e["cpu"] for e in t[3]] for t in scored[:groups]]
    used = {c for s in sets for c in s}
    # one thread per core of ONE L3 group can be too few on a small SMT host (2 cores / 4 threads -> talker + 1
    # receiver, which is client-bound, and no core at all for config #5's second talker): top such a set up with the
    # quietest remaining CPUs, sibling threads and cores of the same group first
    need = min(per, NEED_CPUS)
    topped = 0
    for s in sets:
        # ... but the TALKER's own sibling thread (s[0] is the talker's CPU) goes last of all: a busy-polling receiver on
        # it would cost the talker almost as much as a busy core (see above), so it is handed out only when nothing else
        # is left -- by then to the last receiver or to config #5's second talker
        spare = sorted((c for c in allowed if c not in used),
                       key=lambda c: (core[c] == core[s[0]], l3[c] != l3[s[0]], round(busy.get(c, 0.0), 2), c))
        while len(s) < need and spare:
            c = spare.pop(0)
            s.append(c)
            used.add(c)
            topped += 1
    return {"policy": "quiet", "sets": sets, "busy_before": {str(c): round(busy.get(c, 0.0), 3) for c in sorted(used)},
            "note": f"quietest {groups} of {len(scored)} L3 groups over a {interval:.2f}s /proc/stat sample; one thread per core; "
                    "talker on the first CPU of its set"
                    + (f"; topped up with {topped} sibling/neighbouring CPU(s) to reach {need} per set" if topped else "")}


def ordered_cpus(refresh: bool = False) -> list[int]:
    """This process's CPUs, the talker's first.  Chosen once per process (``refresh`` re-samples)."""
    global _cached
    if _cached is None or refresh:
        _cached = choose()
    return list(_cached["sets"][0])


def describe() -> dict:
    ordered_cpus()
    assert _cached is not None
    return {k: v for k, v in _cached.items() if k != "sets"} | {"cpus": _cached["sets"][0][:8]}


def use(cpus: list[int], info: dict | None = None) -> None:
    """Adopt a set chosen elsewhere (rank 0 picks one L3 group per replica and hands them out)."""
    global _cached
    _cached = dict(info or {"policy": "given", "busy_before": {}, "note": ""}, sets=[list(cpus)])
