#!/usr/bin/env python3
"""Rewrite BASELINE.md section 2 (between the FORMAL markers) from the JSON files under profiles/."""
from __future__ import annotations

import json
import re
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
P = REPO / "profiles"


def table(name: str) -> str:
    lines = [l for l in (P / name).read_text().splitlines() if l.strip()]
    return lines[0] + "\n\n" + "\n".join(lines[1:])


def driver_lines() -> list[tuple[str, dict, dict]]:
    """(file name, driver record, parsed bench line) for every BENCH_rNN.json the driver left at the repo root.  These are
    the numbers of record: the driver's own run of bench.py on a fresh box, not the builder's."""
    out = []
    for f in sorted(REPO.glob("BENCH_r*.json")):
        rec = json.loads(f.read_text())
        tail = (rec.get("run") or {}).get("stdout_tail") or ""
        line = None
        for cand in tail.splitlines():
            if cand.startswith("{"):
                try:
                    line = json.loads(cand)
                except json.JSONDecodeError:
                    line = None
        if line is None:            # the tail was cut: the driver's own parse has the contract keys, the cut tail the rest
            line = dict(rec.get("parsed") or {})
            line.update(salvage(tail, line))
        out.append((f.name, rec, line))
    return out


def salvage(tail: str, parsed: dict) -> dict:
    """What a stdout tail that lost the head of its JSON line still holds (BENCH_r03.json: the driver keeps ~8 KB, the
    round-3 line was 12 KB).  Whole sub-objects are decoded where they start inside the tail; nothing is guessed.  The
    restatement leg's repetitions survive only as their diagnostics (wall clock, busy fraction): the rate of each is the
    run's own delivery count over that wall clock, and the entry says how many of the three were visible."""
    dec, got = json.JSONDecoder(), {}

    def after(key: str, start: int = 0):
        i = tail.find(f'"{key}": ', start)
        if i < 0:
            return None, -1
        try:
            val, end = dec.raw_decode(tail, i + len(key) + 4)
            return val, end
        except json.JSONDecodeError:
            return None, -1

    for key in ("configs", "warnings", "extras_errors", "device_floor"):
        val, _ = after(key)
        if val is not None:
            got[key] = val
    head = tail[:tail.find('"configs": ')] if '"configs": ' in tail else ""
    reps, pos = [], 0
    while True:
        i = head.find('{"wall_s": ', pos)
        if i < 0:
            break
        try:
            d, pos = dec.raw_decode(head, i)
            reps.append(d)
        except json.JSONDecodeError:
            break
    la = re.search(r'"loadavg_before": \[([0-9.]+)', head)
    if la:
        got["host"] = {"loadavg_before_run": [float(la.group(1))], "salvaged": "first load average still visible in the cut tail (a restatement repetition's)"}
    deliveries = re.search(r"(\d+) deliveries", (parsed.get("cpu_baseline") or {}).get("sample", ""))
    if reps and deliveries and parsed.get("value"):
        rates = [int(deliveries.group(1)) / d["wall_s"] for d in reps]
        mid = sorted(rates)[(len(rates) - 1) // 2]          # the lower of two: nothing is rounded up in a salvaged entry
        got["cpu_baseline_port"] = {"value": mid, "ratio_to_timed_run": mid / parsed["value"], "rate_all_reps": rates,
                                    "server_cpu_us_per_written_line": reps[rates.index(mid)].get("server_cpu_us_per_written_line"),
                                    "salvaged": f"{len(reps)} of 3 repetitions visible in the driver's cut stdout tail: " + " and ".join(f"{r:,.0f}" for r in rates)
                                                + " = deliveries / wall_s of each"}
    return got


def record() -> str:
    """Section 0: the driver-run lines, one row per round (the number of record is the driver's)."""
    rows = ["| Driver file | command | headline workload | **delivered lines/s** | ms/step | host load average | `roofline.frac` as printed | vs demonstrated peak | vs CPU-time extrapolation | restatement leg (`cpu_baseline_port`) | as-shipped flags: `-O0` build (`cpu_baseline_O0`, from round 5) | configs medians #1 input/s · #2 · #3 (· #3 with six rooms, from round 4) · #4 · #5 |",
            "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for name, rec, j in driver_lines():
        rf = j.get("roofline") or {}
        probe = rf.get("probe") or {}
        ach = rf.get("achieved")
        demo = rf.get("peak") if "peak_extrapolated" in rf else (probe.get("full_open") or {}).get("written_lines_per_s_wall")
        extra = rf.get("peak_extrapolated") or (probe.get("full_closed") or {}).get("written_lines_per_s_cpu")
        port = j.get("cpu_baseline_port") or {}
        ratio = port.get("ratio_to_timed_run") or (port["value"] / j["value"] if port.get("value") and j.get("value") else None)
        cs = {c["name"]: (c.get("delivered_lines_per_s") or c.get("input_lines_per_s")) for c in j.get("configs", []) if "name" in c}
        wl = (j.get("config") or {}).get("workload", "?")
        rows.append(f"| `{name}` (`{rec.get('head', '?')}`) | `{rec.get('cmd', '?')}` | {wl.split(',')[0]} | **{j.get('value', 0):,.0f}** | {j.get('ms_per_step', '?')} | "
                    f"{((j.get('host') or {}).get('loadavg_before_run') or ['?'])[0]} | {rf.get('frac', '—')} ({rf.get('bound', 'no roofline')}) | "
                    f"{ach / demo:.3f} | {ach / extra:.3f} | " if ach and demo and extra else
                    f"| `{name}` (`{rec.get('head', '?')}`) | `{rec.get('cmd', '?')}` | {wl.split(',')[0]} | **{j.get('value', 0):,.0f}** | {j.get('ms_per_step', '?')} | "
                    f"{((j.get('host') or {}).get('loadavg_before_run') or ['?'])[0]} | {rf.get('frac', '—')} | — | — | ")
        rows[-1] += (f"{port['value']:,.0f} (×{ratio:.2f} of the timed run"
                     + (f", {port['server_cpu_us_per_written_line']} µs/line" if port.get("server_cpu_us_per_written_line") else "")
                     + (f"; {port['salvaged']}" if port.get("salvaged") else "") + ")" if port.get("value") else "—")
        o0 = j.get("cpu_baseline_O0") or {}
        rows[-1] += " | " + (f"{o0['value']:,.0f} (×{o0['ratio_to_timed_run']:.2f} of the timed run, {o0.get('server_cpu_us_per_written_line', '?')} µs/line; "
                             f"{' · '.join(f'{x:,.0f}' for x in o0.get('rate_all_reps', []))})" if o0.get("value") else "—")
        rows[-1] += " | " + (" · ".join(f"{cs[k]:,.0f}" for k in ("config1", "config2", "config3", "config3_six_rooms", "config4", "config5") if k in cs) or "—") + " |"
    if len(rows) == 2:
        return ""
    notes = []
    for name, rec, j in driver_lines():
        for w in j.get("warnings") or []:
            notes.append(f"* `{name}` warning: {w}")
        rf = j.get("roofline") or {}
        legs = rf.get("probe_legs") or {k: {"wall_all": v} for k, v in (rf.get("demonstrated_wall_all") or {}).items()}
        if legs.get("open") and legs.get("closed"):
            o, c = sorted(legs["open"]["wall_all"])[len(legs["open"]["wall_all"]) // 2], sorted(legs["closed"]["wall_all"])[len(legs["closed"]["wall_all"]) // 2]
            how = ("; CPU/wall of the probing thread " + " / ".join(f"{k} {legs[k].get('cpu_over_wall')}" for k in ("open", "closed"))
                   + f", load average {legs['open'].get('loadavg_before')}") if "cpu_over_wall" in legs["open"] else \
                  "; the line of that round kept no CPU/wall figure per probe leg, so the cause cannot be read from the record"
            if o < 0.9 * c or any(max(l["wall_all"]) / min(l["wall_all"]) > 1.15 for l in (legs["open"], legs["closed"])):
                notes.append(f"* `{name}` probe legs: open loop {' / '.join(f'{x:,.0f}' for x in legs['open']['wall_all'])}, closed loop "
                             f"{' / '.join(f'{x:,.0f}' for x in legs['closed']['wall_all'])} lines/s — open-loop median {o / c:.2f} × the closed-loop median{how} "
                             f"(`DESIGN.md` §10; `peak` is the max over the six and was a clean closed-loop repetition)")
    # the headline is a +-4 % quantity on a shared host; say so once, computed from the records themselves.
    # Comparable records = round 3 onwards (round 2 ran under the old core placement).
    comp = [(n, j) for n, _r, j in driver_lines() if re.search(r"r(\d+)", n) and int(re.search(r"r(\d+)", n).group(1)) >= 3 and j.get("value")]
    spread = ""
    if len(comp) >= 2:
        vals = [j["value"] for _n, j in comp]
        loads = [((j.get("host") or {}).get("loadavg_before_run") or [None])[0] for _n, j in comp]
        loads = [x for x in loads if isinstance(x, (int, float))]
        cpu = [j.get("server_cpu_us_per_written_line") for _n, j in comp if j.get("server_cpu_us_per_written_line")]
        mid = (max(vals) + min(vals)) / 2
        spread = (f"**Read the headline as a range, not as six digits:** the driver's comparable records {comp[0][0][6:9]}–{comp[-1][0][6:9]} span "
                  f"{min(vals) / 1000:.0f}–{max(vals) / 1000:.0f} k delivered lines/s (±{100 * (max(vals) - min(vals)) / 2 / mid:.0f} % about the middle)"
                  + (f" at host load averages {min(loads):.0f}–{max(loads):.0f}" if loads else "")
                  + " on the shared 256-CPU host (round 2's 692 k used the old core placement); the talker was ≈1.00 busy in each"
                  + (f", and the server CPU per written line moved between {min(cpu):.2f} and {max(cpu):.2f} µs" if len(cpu) >= 2 else "")
                  + " — a slower core on a busier host, not a change in the program. The latest row is the number of record; the spread is its error bar.\n\n")
    return ("## 0. Numbers of record — the driver's own runs of `bench.py` on a fresh MI355X box\n\n"
            + spread +
            "Generated by `tools/refresh_baseline_md.py` from the `BENCH_rNN.json` files the driver leaves at the repository root. Where a builder-run\n"
            "figure elsewhere in this file or in `DESIGN.md` differs, **these are the numbers of record**; the builder's runs on other allocations of the\n"
            "same host are the spread around them. `vs demonstrated peak` = lines written per second by the talker ÷ the best wall-clock rate a\n"
            "system-call-only loop reached in the same line; `vs CPU-time extrapolation` = the same ÷ the closed loop's writes per CPU second.\n\n"
            + "\n".join(rows) + "\n" + ("\n" + "\n".join(notes) + "\n" if notes else "") +
            "\nRound 2's restatement leg read 5.3× low at equal CPU per line and the line could not say why (one repetition, no counters). The\n"
            "driver's own round-3 run shows the same leg clean (talker busy 0.999–1.00 in every visible repetition, `warnings: []`); the cause of\n"
            "round 2's reading stays undetermined and the item is closed (`DESIGN.md` §10). `BENCH_r03.json` holds only the last ~8 KB of a 12 KB\n"
            "line, so its load average, restatement leg and configs above are read out of that cut tail (`salvage()` in the generator; nothing is\n"
            "guessed, the entry says what was visible). From round 4 the line printed on stdout is a compact record under 6 KB that the driver's\n"
            "tail keeps whole; the full record goes to `gpurun_out/bench_full_n<N>.json`, which only builder-run `gpurun` calls bring back (the\n"
            "driver's bench pull is `n1.out`/`n1.err`/`n1.wall` + `smi.*`). From round 5 the line therefore carries, per probe leg, every\n"
            "repetition's rate, the CPU/wall ratio of the probing thread and the load average (`roofline.probe_legs`), a `probe:` warning when the\n"
            "open loop reads below 0.9 × the closed one or a leg spreads more than ×1.15, and `cpu_baseline_O0`: the headline workload on the\n"
            "reference compiled with its own build script's flags (no `-O`, `/root/reference/build:7,15`).\n")


def round2() -> str:
    """Section 2R: what round 2 added -- generated from profiles/*_r02_*."""
    b = json.loads((P / "bench_r02_n1_mi355xhost.json").read_text())
    pr = json.loads((P / "probe_r02_mi355xhost.json").read_text())["shapes"]
    r1 = json.loads((P / "baseline_r01_mi355xhost_reference.json").read_text())["results"]
    r2 = json.loads((P / "baseline_r02_mi355xhost_reference.json").read_text())["results"]
    rep2 = json.loads((P / "bench_r02_replicas2_mi355xhost.json").read_text())
    rep4 = json.loads((P / "bench_r02_replicas4_mi355xhost.json").read_text())
    rf = b["roofline"]
    rows = ["| # | N | delivered lines/s (median) | input lines/s | all repetitions | server µs/line | exact | vs round-1 formal baseline |", "|---|---|---|---|---|---|---|---|"]
    for c in b["configs"]:
        base = r1[c["name"]]
        mine, ref = (c["delivered_lines_per_s"], base["delivered_lines_per_s"]) if base["delivered_lines_per_s"] else (c["input_lines_per_s"], base["input_lines_per_s"])
        rows.append(f"| {c['name'][-1]}{' (headline)' if c.get('includes_headline_run') else ''} | {c['n']} | {c['delivered_lines_per_s']:,.0f} | {c['input_lines_per_s']:,.0f} | "
                    f"{', '.join(f'{x:,.0f}' for x in c['rate_all_reps'])} | {c['server_cpu_us_per_line']:.2f} | {'✓' if c['exact'] else '✗'} | {mine / ref - 1:+.1%} |")
    cfg_table = "\n".join(rows)
    samples = ["| Run (`profiles/…`) | host load average | headline delivered lines/s | µs/line | #1 input/s | #2 | #3 | #4 (median of 3) | #5 | roofline peak | frac | frac vs write-only |", "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for f in ("bench_r02_n1_mi355xhost_run1_load16.json", "bench_r02_n1_mi355xhost.json", "bench_r02_n1_mi355xhost_load40.json"):
        x = json.loads((P / f).read_text())
        cs = {c["name"]: (c["delivered_lines_per_s"] or c["input_lines_per_s"]) for c in x["configs"]}
        samples.append(f"| `{f}` | {x['host']['loadavg_before_run'][0]:.0f} | {x['value']:,.0f} | {x['server_cpu_us_per_written_line']} | "
                       + " | ".join(f"{cs[k]:,.0f}" for k in ("config1", "config2", "config3", "config4", "config5"))
                       + f" | {x['roofline']['peak']:,.0f} | **{x['roofline']['frac']}** | {x['roofline']['frac_write_only']} |")
    samples_table = "\n".join(samples)
    resid = []
    for f in ("bench_r02_n1_mi355xhost_run1_load16.json", "bench_r02_n1_mi355xhost.json", "bench_r02_n1_mi355xhost_load40.json"):
        x = json.loads((P / f).read_text())
        resid.append(1e3 * x["server_cpu_us_per_written_line"] - x["roofline"]["probe"]["full_closed"]["cpu_ns_per_written_line"])
    resid_lo, resid_hi = min(resid), max(resid)
    nl = b["configs"][4]["netlink"]
    prow = ["| Shape | write-only, closed loop (round 1's ceiling) | + 1 select(1024) + 1 read per input line, closed loop (CPU time) | same, open loop (wall clock, demonstrated) | quoted peak | select+read per input line | per write |", "|---|---|---|---|---|---|---|"]
    for name, sh in pr.items():
        L = sh["legs"]
        prow.append(f"| {name} | {sh['peak_write_only']:,.0f} | {sh['peak_closed_loop_cpu_time']:,.0f} | {sh['peak_open_loop_wall_demonstrated']:,.0f} | **{sh['peak']:,.0f}** | "
                    f"{L['full_closed']['cpu_ns_select_read_per_line'] / 1e3:.2f} µs | {L['full_closed']['cpu_ns_per_write'] / 1e3:.3f} µs |")
    probe_table = "\n".join(prow)
    drift = ["| Config | round 1 (median of 3) | round 2 (median of 3) | Δ |", "|---|---|---|---|"]
    for k in ("config1", "config2", "config2_colour_on", "config2_all_send", "config3", "config4", "config4_colour_on", "config5"):
        a, c = r1[k], r2[k]
        va, vc = (a["delivered_lines_per_s"], c["delivered_lines_per_s"]) if a["delivered_lines_per_s"] else (a["input_lines_per_s"], c["input_lines_per_s"])
        drift.append(f"| {k} | {va:,.0f} | {vc:,.0f} | {vc / va - 1:+.1%} |")
    drift_table = "\n".join(drift)
    shipped = (P / "shipped_files_r02_container.md").read_text().strip()
    srv_us = b["server_cpu_us_per_written_line"]
    probe_us = rf["probe"]["full_closed"]["cpu_ns_per_written_line"] / 1e3
    return f"""## 2R. Round 2 — the driver-visible line, the corrected ceiling, the shipped files

Generated by `tools/refresh_baseline_md.py` from `profiles/*_r02_*`. Nothing here changes the round-1 numbers above; it
closes what the round-1 review found short. **These are builder runs; the number of record for round 2 is the driver's (§0): 692,594
delivered lines/s, `frac` 0.971 demonstrated / 0.955 CPU-time. The `frac` column below is round 2's min()-based figure — round 3
redefined the ceiling as the best demonstrated rate (§2S.2).**

### 2R.1 One `bench.py` line now carries all five configurations (MI355X-box host)

`python bench.py` (driver default, N=1): headline = BASELINE `configs[3]`, **1000 clients, `.shout`** — the largest single-talker
configuration (`select(FD_SETSIZE)`, `nuts333.c:94`, caps N just above 1000). `profiles/bench_r02_n1_mi355xhost.json`:
**{b['value']:,.0f} delivered lines/s** ({b['ms_per_step']} ms per 100-line step, {srv_us} µs server CPU per written line, host load average
{b['host']['loadavg_before_run'][0]} when it started). The same line's `configs` array, reference build, formal sizes, #1–#4 three repetitions each:

{cfg_table}

Restatement on the headline workload, same size (`cpu_baseline_port`): {b['cpu_baseline_port']['value']:,.0f} lines/s, {b['cpu_baseline_port']['server_cpu_us_per_written_line']} µs/line.
Independent replicas under `torch.distributed.run` (`--gpus N`, no GPU involved, config #4 each): N=2 → {rep2['value']:,.0f}, N=4 → {rep4['value']:,.0f} lines/s.

Configuration #5's link traffic is **measured**, not modelled: each talker's `write(2)` count (`/proc/<pid>/io`) minus the lines its own
clients received = {nl['writes_t1_to_t2']:,} frames talker1→talker2 (one `ACT` per relayed shout, `nuts333.c:3801`) and {nl['writes_t2_to_t1']:,} talker2→talker1
({nl['expected_msg_frames']:,} `MSG…EMSG`, `c:1302-1305`, + {nl['expected_prm_frames']:,} `PRM`, `c:2181`); a counting relay on the link (`nuts333_amd/linktap.py`) sees the same
numbers by verb (`tests/test_harness.py::test_config5_link_frames_counted_on_the_wire`).

The same command three times, on three allocations of the box (the first with an earlier revision of `bench.py`: same measurement,
fewer context fields in the line):

{samples_table}

The host is shared. At load averages below ≈ 20 the medians of three reproduce round 1 within ±3 % for the saturating configurations
#2–#4; #1 (a latency-bound ping-pong between two cores, talker 60 % busy) and #5 (bound by the delayed-ACK timer) move by ±10 %. The
roofline fraction holds at 0.95–0.97 throughout because the probe meets the same neighbours as the run. With other tenants busier
(load average ≈ 60) single runs came in 5–30 % low on whichever cores a neighbour touched; run order, `TIME_WAIT` sockets and idle time
made no difference (`profiles/hostnoise_r02_mi355xhost.log`). The formal baseline re-run in round 2 was taken on such a busy
allocation and shows it in the 1000-client rows:

{drift_table}

### 2R.2 Host system-call ceiling, with the work the survey said to count

Per input line the algorithm needs 1 `select(FD_SETSIZE)` + 1 `read` + (recipients + 1) `write(2)` (`nuts333.c:94,136,1363`; SURVEY.md §8d). Round 1
priced only the writes. `loadgen --probe-line` does exactly those calls and nothing else, three ways (`profiles/probe_r02_mi355xhost.json`,
medians of 3, lines written per second on one core):

{probe_table}

`bench.py` quotes the lower of the closed-loop CPU-time rate and the open-loop wall-clock rate: here the open loop *demonstrates* a
wall-clock rate at or above the quoted peak, so the ceiling is a rate a run actually reached. Headline run: achieved {rf['achieved']:,.0f},
peak {rf['peak']:,.0f} → **frac {rf['frac']}** (against round 1's write-only ceiling: {rf['frac_write_only']}). The residual is the talker's own user-space
work: {srv_us} µs per written line against {probe_us:.3f} µs for the bare system calls = {1e3 * (srv_us - probe_us):.0f} ns per recipient here, {resid_lo:.0f}–{resid_hi:.0f} ns
over the three runs above — `oracle/pathbench`'s 50–60 ns per-recipient transducer plus the fan-out predicate and list walk
(`profiles/pathbench_r01_mi355xhost.json`), to within what the difference of two noisy measurements allows. There is nothing left to
find on this path, on any device.

### 2R.3 Configurations #1 and #5 on the reference's shipped files (build container only)

{shipped}

The shipped account (`userfiles/Fred.D`) has its prompt on, so every input line costs two writes (acknowledgement + prompt); the generated
tree with an equally-flagged account costs the same per input line within the VM's noise (`tests/test_shipped_files.py` asserts 25 % on
medians of 5). `datafiles/config2` does not boot as shipped (`logging YES`, line 11); the temporary copy fixes that one line. With the
shipped files only `Fred` may shout, so "#5" here is one remote sender and one NEW-level listener: ≈ 43 shouts/s, each waiting out
Nagle + delayed ACK on the link exactly as in the generated-tree run.
"""


def round3() -> str:
    """Section 2S: what round 3 added -- generated from profiles/*_r03_*."""
    import statistics
    files = ["bench_r03_driverargs_run1_mi355xhost.json", "bench_r03_driverargs_run2_mi355xhost.json", "bench_r03_driverargs_run3_mi355xhost.json",
             "bench_r03_n1_mi355xhost_load64.json", "bench_r03_n1_firstcpus_mi355xhost.json", "bench_r03_under_rocprof_mi355xhost.json"]
    rows = ["| Line (`profiles/…`) | args | load average | cores (talker first) | delivered lines/s | µs/line | talker busy · run-queue wait · gaps ≥ 5 ms | `frac` (best demonstrated) | `frac_extrapolated` | restatement ×3: median (all) | ÷ timed run | #1 input/s · #2 · #3 · #4 · #5 (medians) | warnings |",
            "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for f in files:
        x = json.loads((P / f).read_text())
        d, r, p = x["diagnostics"], x["roofline"], x["cpu_baseline_port"]
        cs = {c["name"]: (c["delivered_lines_per_s"] or c["input_lines_per_s"]) for c in x["configs"]}
        rows.append(f"| `{f}` | `--steps {x['steps']} --warmup {x['warmup']}` | {x['host']['loadavg_before_run'][0]:.0f} | {x['host']['placement']['policy']}: "
                    f"{', '.join(map(str, x['host']['placement']['cpus'][:5]))} | **{x['value']:,.0f}** | {x['server_cpu_us_per_written_line']} | "
                    f"{d['server_busy_frac']:.2f} · {d['server_run_delay_frac']:.4f} · {d['progress_gaps']['count']} | {r['frac']} | {r['frac_extrapolated']} | "
                    f"{p['value']:,.0f} ({', '.join(f'{v:,.0f}' for v in p['rate_all_reps'])}) | {p['ratio_to_timed_run']} | "
                    + " · ".join(f"{cs[k]:,.0f}" for k in ("config1", "config2", "config3", "config4", "config5")) + f" | {len(x['warnings'])} |")
    bench_table = "\n".join(rows)
    hunt = []
    for f in ("stall_hunt_r03_mi355xhost_blocked.log", "stall_hunt_r03_mi355xhost_interleaved.log"):
        if not (P / f).exists():
            continue
        runs = [json.loads(l) for l in (P / f).read_text().splitlines() if l.startswith("{")]
        for pol in ("first", "quiet"):
            rs = [r for r in runs if r["policy"] == pol]
            if rs:
                hunt.append(f"| `{f}` | {pol} | {len(rs)} | {min(r['loadavg'] for r in rs):.0f}–{max(r['loadavg'] for r in rs):.0f} | "
                            f"{statistics.median(r['delivered_lines_per_s'] for r in rs):,.0f} ({min(r['delivered_lines_per_s'] for r in rs):,.0f}–{max(r['delivered_lines_per_s'] for r in rs):,.0f}) | "
                            f"{statistics.median(r['server_cpu_us_per_written_line'] for r in rs):.3f} | {min(r['server_busy_frac'] for r in rs):.3f} | "
                            f"{max(r['server_run_delay_frac'] for r in rs):.4f} | {sum(r['progress_gaps']['count'] for r in rs)} | {max(r['ack_latency_us']['max'] for r in rs) / 1e3:.1f} ms | "
                            f"{sum(1 for r in rs if r['attribution'])} |")
    hunt_table = "\n".join(["| Log (`profiles/…`) | placement | runs | load average | restatement delivered lines/s: median (range) | µs/line (median) | lowest talker busy | highest run-queue wait | gaps ≥ 5 ms | slowest ack | stalls attributed |",
                             "|---|---|---|---|---|---|---|---|---|---|---|"] + hunt)
    pr = json.loads((P / "probe_r03_mi355xhost.json").read_text())["shapes"]
    prow = ["| Shape | write-only (CPU time) | full, closed loop (CPU-time extrapolation) | full, open loop (wall, median of 3) | **peak = best demonstrated of 6** | select+read per input line | per write |", "|---|---|---|---|---|---|---|"]
    for name, sh in pr.items():
        L = sh["legs"]
        prow.append(f"| {name} | {sh['peak_write_only']:,.0f} | {sh['peak_extrapolated']:,.0f} | {sh['peak_demonstrated_median_open_loop']:,.0f} | **{sh['peak']:,.0f}** | "
                    f"{L['full_closed']['cpu_ns_select_read_per_line'] / 1e3:.2f} µs | {L['full_closed']['cpu_ns_per_write'] / 1e3:.3f} µs |")
    probe_table = "\n".join(prow)
    rep2 = json.loads((P / "bench_r03_replicas2_mi355xhost.json").read_text())
    rep4 = json.loads((P / "bench_r03_replicas4_mi355xhost.json").read_text())
    return f"""## 2S. Round 3 — a record that is self-consistent and explains itself

Generated by `tools/refresh_baseline_md.py` from `profiles/*_r03_*`. Nothing here changes a number above; §0 holds the numbers of record.

### 2S.1 `bench.py` on the MI355X box's host, round-3 revision (builder runs)

Every leg now carries the counters that say where the talker's wall clock went (`DESIGN.md` §6.5); the restatement leg runs three times; the
roofline ceiling is the best wall-clock rate any of six system-call-only repetitions demonstrated, with the CPU-time extrapolation beside it.

{bench_table}

The talker was the bottleneck in every line (busy 1.00, no run-queue wait, no gap); what differs between them is the cost of a line on a
core that shares its sibling thread and L3 with other tenants. At load average 64 the closed-loop probe was disturbed more than the talker
(`frac_extrapolated` > 1): an extrapolated ceiling can be beaten, a demonstrated one cannot.
Independent replicas (one quiet L3 group each, no GPU opened by any rank): N=2 → {rep2['value']:,.0f}, N=4 → {rep4['value']:,.0f} lines/s.

### 2S.2 The system-call ceiling, round-3 definition (`profiles/probe_r03_mi355xhost.json`)

{probe_table}

### 2S.3 Looking for round 2's stalled leg (`profiles/stall_hunt_r03_experiment.py`)

Round 2's driver line had the restatement at 0.19 of the reference's rate with the same CPU per line, and no counters to explain it
(`DESIGN.md` §10). The exact sequence — three probe legs, then the restatement on 2,000 `.shout` lines to 999 recipients — repeated on the
box with the old placement (`first`: talker on CPU 0, receivers on 1–4) and the new one (`quiet`):

{hunt_table}

Not reproduced. A repeat would now be attributed in the line's `warnings`.
"""


def main() -> None:
    b = json.loads((P / "bench_r01_n1_mi355xhost.json").read_text())
    b4 = json.loads((P / "bench_r01_config4_mi355xhost.json").read_text())
    r2 = json.loads((P / "bench_r01_replicas2_mi355xhost.json").read_text())
    r4 = json.loads((P / "bench_r01_replicas4_mi355xhost.json").read_text())
    pb = json.loads((P / "pathbench_r01_mi355xhost.json").read_text())
    ref = json.loads((P / "baseline_r01_mi355xhost_reference.json").read_text())["results"]
    cref = json.loads((P / "baseline_r01_container_reference.json").read_text())["results"]
    lo = min(ref[k]["delivered_lines_per_s"] for k in ("config2", "config3", "config4"))
    hi = max(ref[k]["delivered_lines_per_s"] for k in ("config2", "config3", "config4"))
    clo = min(cref[k]["delivered_lines_per_s"] for k in ("config2", "config3", "config4"))
    chi = max(cref[k]["delivered_lines_per_s"] for k in ("config2", "config3", "config4"))
    port = json.loads((P / "baseline_r01_mi355xhost_port.json").read_text())["results"]
    fast = json.loads((P / "baseline_r01_mi355xhost_port_fast.json").read_text())["results"]
    rows = ["| Config | restatement, reference cost model | restatement, fast mode | ratio |", "|---|---|---|---|"]
    for k in ("config2", "config2_colour_on", "config3", "config4", "config4_colour_on", "config5"):
        a, f = port[k]["delivered_lines_per_s"], fast[k]["delivered_lines_per_s"]
        rows.append(f"| {k} | {a:,.0f} lines/s · {port[k]['server_cpu_us_per_written_line']:.2f} µs/line | {f:,.0f} lines/s · {fast[k]['server_cpu_us_per_written_line']:.2f} µs/line | ×{f / a:.2f} |")
    fast_table = "\n".join(rows)
    body = f"""## 2. Formal CPU baseline (round 1) — the deliverable

Status: **recorded**. Harness, rules and definitions: `DESIGN.md` §5–§6; raw JSON per run under `profiles/`
(`baseline_r01_*`; this section is generated from them by `tools/refresh_baseline_md.py`). Every run below delivered
*exactly* the expected number of lines to *every* client (`Delivered = expected ✓`). The talker is single-threaded:
all server numbers are one core by construction.

Columns: **Delivered lines/s** = lines written to recipients other than the sender ÷ timed wall (the BASELINE.json
metric); **Server CPU µs/line** = talker CPU (`/proc/<pid>/schedstat`) ÷ every line it wrote, acks included, with the
user-space share from `/proc/<pid>/stat` (10 ms ticks, coarse — `oracle/pathbench` gives the precise figure below);
**Busy** = talker CPU ÷ wall (1.00 = the talker, not the load generator, is the bottleneck); **B/line** = bytes on the
wire per written line. `config2_all_send` = all ten clients send concurrently (saturating variant of #2);
`*_colour_on` = every account has colour on (two `write(2)` per recipient, `nuts333.c:1363,1365`).
The synthetic clients run on 4 cores next to the talker's (`profiles/sweep_loadgen_threads_r01_mi355xhost.log`:
2–7 receiver threads measure the same; spreading receivers over 10+ cores / a second CCD inflates the *talker's*
write cost by ≈45 % through cache-line migration, an artefact of loopback).

### 2.1 On the MI355X box's host cores

{table('baseline_r01_mi355xhost_reference.md')}

BASELINE.json words configuration #3 as "all 6 rooms"; the shipped config has five plus an orphan `shop.R`
(SURVEY.md §4). Both variants, same host (the six-room one adds `shop` off the hallway):

{table('baseline_r01_mi355xhost_config3_rooms.md')}

Same host, reference compiled **without** optimisation (what the reference's own `build` script produces):

{table('baseline_r01_mi355xhost_reference_O0.md')}

Same host, our CPU restatement (`oracle/talker_port`, parity-pinned; `cpu_baseline.kind = "port"`):

{table('baseline_r01_mi355xhost_port.md')}

`bench.py` (BASELINE `configs[1]`, 10 × 2000 lines) on that host: **{b['value']:,.0f} delivered lines/s**,
{b['server_cpu_us_per_written_line']} µs of server CPU per written line, host-syscall roofline fraction **{b['roofline']['frac']}**
(peak {b['roofline']['peak']:,.0f} write(2)/s/core from the closed-loop probe); config #4 shape: {b4['value']:,.0f} lines/s, fraction {b4['roofline']['frac']}.
Independent replicas (`--gpus N`, no GPU involved): N=2 → {r2['value']:,.0f}, N=4 → {r4['value']:,.0f} lines/s.
User-space work per recipient, timed in isolation (`oracle/pathbench`): transduce one 67-byte line {pb['transduce_say_colour_off_ns']} ns
(shout, colour on: {pb['transduce_shout_colour_on_ns']} ns), fan-out predicate {pb['fanout_predicate_ns']} ns, format the line once {pb['format_line_once_ns']} ns.
What touching the GPU costs there: {b['device_floor']['kernel_launch_plus_sync_us']} µs for one trivial kernel launch + sync,
{b['device_floor']['h2d_64B_kernel_d2h_69KB_sync_us']} µs for a 64 B → kernel → 69 KB round trip (`profiles/rocprof_r01_kernel_stats.md`).

### 2.2 In the build container (8 vCPU Xeon @ 2.1 GHz, KVM guest)

{table('baseline_r01_container_reference.md')}

`-O0` reference build:

{table('baseline_r01_container_reference_O0.md')}

Restatement:

{table('baseline_r01_container_port.md')}

Round 2, same container: configuration #1 **on the reference's shipped files** (`datafiles/config`, `fred`/`test`, `.go lounge`) costs
the same per input line as the generated tree with an equally-flagged account — rows and method in §2R.3.

### 2.3 Measured CPU-side headroom (restatement with `NUTS_PORT_FAST=1`, MI355X-box host)

The three changes `INTEGRATION.md` §3 proposes — transduce once per colour variant instead of once per recipient,
send the trailing colour reset in the same `write(2)`, `TCP_NODELAY` on netlink sockets — switched on in the
restatement. The bytes on every socket are unchanged (all recorded sessions replay byte-exact in this mode,
`tests/test_parity_transcripts.py`); only the work differs.

{fast_table}

### 2.4 Reading

* Single talker, any N from 10 to 1000: **≈{lo / 1000:.0f}–{hi / 1000:.0f} k delivered lines/s on one EPYC 9575F core**,
  {min(ref[k]['server_cpu_us_per_written_line'] for k in ('config2', 'config3', 'config4')):.2f}–{max(ref[k]['server_cpu_us_per_written_line'] for k in ('config2', 'config3', 'config4')):.2f} µs of CPU per written line, of which ≈{pb['transduce_say_colour_off_ns'] / 1000:.2f} µs is the
  transducer; ≈{clo / 1000:.0f}–{chi / 1000:.0f} k lines/s and ≈5–7 µs in the (virtualised) build container, where the loopback TCP
  path itself costs ≈5 µs per closed-loop write.
* The talker already runs at ≈{b['roofline']['frac']:.2f} of what one core can do issuing nothing but `write(2)`.
* Optimisation level barely matters (`-O0` vs `-O2` within noise): the time is not in user space.
* Colour on costs ≈25 % (a second 4-byte `write` per recipient).
* Config #5 (netlink): ≈75–100 input lines/s end to end, p99 ack 44–60 ms — the talker↔talker socket never sets
  `TCP_NODELAY`, so consecutive small frames wait for a delayed ACK. 2000 input lines → 20,000 `MSG…EMSG` frames
  talker2→talker1, 1000 `ACT` frames the other way, plus one `PRM` per `ACT`.
* §2.3: with colour off the reference's user-space redundancy is worth nothing measurable (it is ≈4 % of the line);
  with colour on the second `write` is worth ≈25–30 %; and **one `setsockopt(TCP_NODELAY)` on the talker↔talker socket
  is worth ≈×{fast['config5']['delivered_lines_per_s'] / port['config5']['delivered_lines_per_s']:.0f}** on the netlink configuration. None of it involves a GPU.
* The survey-time figures in §3 below (≈2.3 µs/line with a Python driver) are consistent with the container numbers.
"""
    path = REPO / "BASELINE.md"
    text = path.read_text()
    r02 = round2()
    if "<!-- R02:BEGIN -->" not in text:
        text = text.replace("<!-- FORMAL:END -->", "<!-- FORMAL:END -->\n\n<!-- R02:BEGIN -->\n<!-- R02:END -->")
    text = re.sub(r"<!-- R02:BEGIN -->.*?<!-- R02:END -->", lambda m: "<!-- R02:BEGIN -->\n" + r02 + "<!-- R02:END -->", text, flags=re.S)
    r03 = round3()
    if "<!-- R03:BEGIN -->" not in text:
        text = text.replace("<!-- R02:END -->", "<!-- R02:END -->\n\n<!-- R03:BEGIN -->\n<!-- R03:END -->")
    text = re.sub(r"<!-- R03:BEGIN -->.*?<!-- R03:END -->", lambda m: "<!-- R03:BEGIN -->\n" + r03 + "<!-- R03:END -->", text, flags=re.S)
    rec = record()
    if "<!-- RECORD:BEGIN -->" not in text:
        text = text.replace("## 1. Reference's own published numbers", "<!-- RECORD:BEGIN -->\n<!-- RECORD:END -->\n\n## 1. Reference's own published numbers", 1)
    text = re.sub(r"<!-- RECORD:BEGIN -->.*?<!-- RECORD:END -->", lambda m: "<!-- RECORD:BEGIN -->\n" + rec + "<!-- RECORD:END -->", text, flags=re.S)
    if "<!-- FORMAL:BEGIN -->" not in text:
        text = re.sub(r"## 2\. Formal CPU baseline.*?(?=## 3\. Survey-time)", "<!-- FORMAL:BEGIN -->\n<!-- FORMAL:END -->\n\n", text, flags=re.S)
    text = re.sub(r"<!-- FORMAL:BEGIN -->.*?<!-- FORMAL:END -->", lambda m: "<!-- FORMAL:BEGIN -->\n" + body + "<!-- FORMAL:END -->", text, flags=re.S)
    path.write_text(text)


if __name__ == "__main__":
    main()
