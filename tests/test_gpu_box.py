"""Tests marked ``gpu``: what the driver runs on a real MI355X box at round end.

This project has no device code (BASELINE.json north_star: not graft-eligible), so these do
not exercise a kernel.  Together with the ``gpubox`` instances of the parity modules (see
tests/conftest.py) they run, on the GPU box's host, the real suite: every golden transcript
replayed byte-exact (restatement, and the prebuilt reference binary that travels with the
snapshot), every BASELINE configuration delivering exact per-client counts on both
implementations, the measured link frames of the two-talker configuration, the system-call
cost model, the bench line's contract -- and they prove that nothing opens ``/root/reference``,
which does not exist there.  One test records what a trivial kernel launch costs on the
device: evidence for the non-eligibility argument, not a product measurement.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

from nuts333_amd import workloads
from nuts333_amd.talker import PORT_BINARY, REF_BINARY, REF_BINARY_O0, reference_expected_but_missing

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def _binary(impl: str) -> Path:
    if impl == "port":
        assert PORT_BINARY.exists()
        return PORT_BINARY
    lost = reference_expected_but_missing()
    if lost:          # built for this snapshot and gone: red, not "N passed, M skipped" 
        pytest.fail(lost)
    if not REF_BINARY.exists():
        pytest.skip("oracle/_ref/nuts333 was not prebuilt into this snapshot")
    return REF_BINARY


# ------------------------------------------------------------------ nothing here may read the reference tree
_AUDIT = r"""
import json, sys
seen = []
def hook(event, args):
    if event in ("open", "os.listdir", "os.scandir", "os.chdir", "subprocess.Popen", "os.exec", "os.posix_spawn"):
        seen.append(repr(args))
sys.addaudithook(hook)
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import __graft_entry__ as g
from nuts333_amd import workloads
b, _ = workloads.pick_binary()
assert workloads.config5(lines=5, binary=b)["exact"]      # two talkers booted BEFORE smoke(): see below
g.smoke()                                                   # its device probe runs in a child of its own
bad = [s for s in seen if "/root/reference" in s]
print("AUDIT " + json.dumps({"events": len(seen), "bad": bad}))
"""


def test_no_path_under_root_reference_is_opened():
    """smoke() + a two-talker run under a Python audit hook: no open/listdir/exec argument may name /root/reference;
    and no shipped binary may have that path compiled in (the talkers use relative paths only, nuts333.h:3-14).
    No process here forks talkers after touching the GPU: the two-talker run comes first, and smoke()
    measures the device floor in a short-lived child."""
    out = subprocess.run([sys.executable, "-c", _AUDIT, str(REPO)], check=True, stdout=subprocess.PIPE, timeout=300,
                         env=dict(os.environ, REFERENCE="/nonexistent")).stdout.decode()
    audit = json.loads([l for l in out.splitlines() if l.startswith("AUDIT ")][-1][6:])
    assert audit["events"] > 50 and audit["bad"] == []
    for binary in [PORT_BINARY, workloads.LOADGEN_BIN, REPO / "oracle" / "_build" / "libnuts_path.so"] + \
                  ([REF_BINARY] if REF_BINARY.exists() else []):
        assert b"/root/reference" not in binary.read_bytes(), binary


# ------------------------------------------------------------------ parity
# All 20 golden sessions (restatement, prebuilt reference), the live restatement<->reference netlink
# interop and the 314 transducer vectors run in this tier too: tests/test_parity_transcripts.py and
# tests/test_nuts_path.py set BOTH_TIERS (tests/conftest.py), their "gpubox" instances carry the gpu marker.


# ------------------------------------------------------------------ the five BASELINE configurations, exact per client
def _check(res):
    assert res["ok"] and res["exact"] and res["per_client_exact"], res
    assert res["deliveries"] == res["expected_deliveries"]
    assert res["acks"] == res["input_lines"] == res["planned_input_lines"]
    assert all(res.get("servers_alive_after", [res.get("server_alive_after", True)]))


IMPLS = ["reference", "port"]


@pytest.mark.parametrize("impl", IMPLS)
def test_config1_boot_talker_one_client_say_in_lounge(impl):
    res = workloads.config1(lines=2000, warmup=100, binary=_binary(impl))
    _check(res)
    assert res["clients"] == 1 and res["deliveries"] == 0 and res["bytes_per_line"] == 65.0
    s = res["servers"][0]
    assert (s["read_syscalls"], s["write_syscalls"]) == (2000, 2000)


@pytest.mark.parametrize("impl", IMPLS)
def test_config2_ten_clients_one_room_say_fanout(impl):
    res = workloads.config2(lines=3000, warmup=300, binary=_binary(impl))
    _check(res)
    assert res["clients"] == 10 and res["deliveries"] == 27000
    s = res["servers"][0]
    assert (s["read_syscalls"], s["write_syscalls"], s["bytes_written"]) == (3000, 30000, res["bytes_total"])


@pytest.mark.parametrize("impl", IMPLS)
def test_config2_colour_on_two_writes_per_line(impl):
    res = workloads.config2(lines=1000, colour=1, binary=_binary(impl))
    _check(res)
    assert res["servers"][0]["write_syscalls"] == 2 * 10 * 1000           # nuts333.c:1363,1365


@pytest.mark.parametrize("impl", IMPLS)
def test_config3_hundred_clients_all_rooms_mixed_say_shout_tell(impl):
    res = workloads.config3(per_client=40, warmup=2, binary=_binary(impl))
    _check(res)
    assert res["clients"] == 100 and res["input_lines"] == 4000
    assert "5 rooms" in res["workload"] and "70/20/10" in res["workload"]
    s = res["servers"][0]
    assert s["write_syscalls"] == res["lines_total"] and s["read_syscalls"] == 4000


@pytest.mark.parametrize("impl", IMPLS)
def test_config3_six_room_variant(impl):
    res = workloads.config3(per_client=10, n=60, six_rooms=True, binary=_binary(impl))
    _check(res)
    assert "6 rooms" in res["workload"]


@pytest.mark.parametrize("impl", IMPLS)
def test_config4_thousand_clients_single_room_shout(impl):
    res = workloads.config4(lines=100, n=1000, warmup=10, binary=_binary(impl))
    _check(res)
    assert res["clients"] == 1000 and res["deliveries"] == 100 * 999
    assert abs(res["bytes_per_line"] - (999 * 69 + 67) / 1000) < 1e-9
    s = res["servers"][0]
    assert (s["read_syscalls"], s["write_syscalls"]) == (100, 100 * 1000)


@pytest.mark.parametrize("impl", IMPLS)
def test_config5_two_server_netlink_cross_link_shout(impl):
    res = workloads.config5(lines=60, warmup=5, binary=_binary(impl))
    _check(res)
    assert res["deliveries"] == 2 * 60 * 14 and len(res["servers"]) == 2
    nl = res["netlink"]
    assert nl["exact"] and nl["writes_t1_to_t2"] == 60 and nl["writes_t2_to_t1"] == 600 + 60


@pytest.mark.parametrize("impl", IMPLS)
def test_config5_link_frames_counted_on_the_wire(impl):
    res = workloads.config5(lines=25, travellers=4, tap=True, binary=_binary(impl))
    _check(res)
    tap = res["netlink"]["tap"]
    assert res["netlink"]["exact"]
    assert tap["marked_dial_to_accept"] == {"ACT": 25} and tap["marked_accept_to_dial"] == {"MSG": 2 * 25 * 3 + 2 * 25}
    assert tap["accept_to_dial"]["MSG"] == tap["accept_to_dial"]["EMSG"] and tap["accept_to_dial"]["PRM"] == 25 + 4


def test_restatement_costs_the_same_system_calls_as_the_reference():
    ref, port = _binary("reference"), _binary("port")
    def sc(r):
        s = r["servers"][0]
        return s["read_syscalls"], s["write_syscalls"], s["bytes_written"]
    for kw in ({"lines": 500}, {"lines": 500, "colour": 1}):
        assert sc(workloads.config2(binary=ref, **kw)) == sc(workloads.config2(binary=port, **kw))
    assert sc(workloads.config3(per_client=10, n=25, binary=ref)) == sc(workloads.config3(per_client=10, n=25, binary=port))


# ------------------------------------------------------------------ bench line and probes on this host
def test_bench_line_covers_all_five_configs_headline_config4(tmp_path):
    record = tmp_path / "full.json"
    out = subprocess.run([sys.executable, str(REPO / "bench.py"), "--steps", "5", "--warmup", "1"], check=True,
                         env={**os.environ, "NUTS_BENCH_FULL_RECORD": str(record)},
                         stdout=subprocess.PIPE, timeout=1100).stdout.decode().strip().splitlines()
    assert len(out) == 1
    # what the driver's record keeps is the last ~8 KB of stdout + stderr: the line fits whole, and it is the compact copy
    # of the full record written beside it (round 3's 12 KB line lost its load average and restatement leg there)
    assert len(out[0]) <= 6000, len(out[0])
    line, j = json.loads(out[0]), json.loads(record.read_text())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "cpu_baseline",
              "delivered", "expected_delivered", "configs_all_exact", "gpu_used"):
        assert line[k] == j[k], k
    # free text: what compact_line promises -- counts whole, kept entries cut as documented; on a busy box the
    # full record may hold more and longer warnings than the line
    sys.path.insert(0, str(REPO))
    import bench
    assert line["warnings_count"] == len(j["warnings"]) and line["extras_errors_count"] == len(j["extras_errors"])
    assert line["warnings"] in [bench.compact_warnings(j["warnings"], t) for t in bench.TIGHT_LEVELS]
    assert line["extras_errors"] in [bench.compact_errors(j["extras_errors"], t) for t in bench.TIGHT_LEVELS]
    assert line["roofline"]["frac"] == j["roofline"]["frac"] and line["roofline"]["peak"] == j["roofline"]["peak"]
    assert [(c["name"], c["delivered_lines_per_s"], c["exact"]) for c in line["configs"]] == \
           [(c["name"], c["delivered_lines_per_s"], c["exact"]) for c in j["configs"]]
    assert line["host"]["loadavg_before_run"] == j["host"]["loadavg_before_run"] is not None
    assert j["config"]["baseline_config"] == "config4" and j["config"]["clients"] == 1000 and j["gpu_used"] is False
    assert j["delivered"] == j["expected_delivered"] == 5 * 100 * 999
    # BASELINE.json's five configurations, #3 both as shipped (5 rooms) and as worded ("all 6 rooms"), at formal sizes
    assert [c["name"] for c in j["configs"]] == ["config1", "config2", "config3", "config3_six_rooms", "config4", "config5"]
    assert all(c["exact"] for c in j["configs"]) and j["configs_all_exact"] and j["extras_errors"] == []
    assert [c["reps"] for c in j["configs"]] == [3, 3, 3, 3, 3, 1] and [c["n"] for c in j["configs"]] == [1, 10, 100, 100, 1000, 20]
    assert "over 6 rooms" in j["configs"][3]["workload"] and j["configs"][3]["input_lines"] == 20000
    assert "6 rooms" in line["configs"][3]["workload"]
    assert "note" in line["configs"][2] or line["warnings_count"] > 0          # (the tight form, many warnings, drops the notes)
    assert j["configs"][5]["netlink"]["exact"] and j["configs"][5]["netlink"]["writes_t2_to_t1"] == 11000
    assert j["host"]["cgroup_throttled_periods_during_run"] in (0, None)
    assert j["configs"][4]["includes_headline_run"] is False          # 500 lines is not the formal 1000: three fresh repetitions
    r = j["roofline"]
    assert r["per_input_line"]["write"] == 1000 and r["per_input_line"]["select"] == 1 and r["per_input_line"]["read"] == 1
    # a ceiling is a rate something reached: the arithmetic is a hard assertion ...
    demonstrated = r["probe"]["full_open"]["written_lines_per_s_wall_all"] + r["probe"]["full_closed"]["written_lines_per_s_wall_all"]
    assert r["peak"] == max(demonstrated) and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["frac_extrapolated"] - r["achieved"] / r["peak_extrapolated"]) < 1e-3
    # ... and the band is an invariant of the LINE, not of the host: probe and talker run at
    # different moments on a shared host, so a reading outside (0.85, 1.02] is allowed only if the line itself names it
    assert 0.85 < r["frac"] <= 1.02 or any(w.startswith("roofline:") for w in j["warnings"]), (r["frac"], demonstrated, j["warnings"])
    assert 0.5 < r["frac"] < 1.5, (r["frac"], demonstrated)          # beyond this no host noise explains it: the harness is broken
    assert j["cpu_baseline"]["kind"] in ("reference", "port") and j["cpu_baseline"]["cores"] == 1
    # the probe legs explain themselves in the line: rates, CPU/wall, load average; a slow open loop or
    # a wide spread is named there too
    for short, leg in (("open", "full_open"), ("closed", "full_closed")):
        got, src = line["roofline"]["probe_legs"][short], r["probe"][leg]
        assert got["wall_all"] == src["written_lines_per_s_wall_all"] and got["cpu_over_wall"] in src["cpu_over_wall_all"]
        assert got["loadavg_before"] is not None and 0 < got["cpu_over_wall"] <= 1.05
    o_med, c_med = (sorted(r["probe"][k]["written_lines_per_s_wall_all"])[1] for k in ("full_open", "full_closed"))
    if o_med < 0.9 * c_med:
        assert any(w.startswith("probe: open-loop") for w in j["warnings"]), j["warnings"]
    # the as-shipped-flags leg: the -O0 build of the same sources, same workload and size, three times
    if j["cpu_baseline"]["kind"] == "reference" and REF_BINARY_O0.exists():
        o0 = j["cpu_baseline_O0"]
        assert o0["reps"] == 3 and o0["exact"] and len(o0["rate_all_reps"]) == 3 and o0["binary"] == "oracle/_ref/nuts333_O0"
        lo, hi = bench.O0_RATIO_BAND           # a shared host can move between two legs: outside the band is fine IF the line says so
        if not lo <= o0["ratio_to_timed_run"] <= hi:
            assert any(w.startswith("-O0 build/reference delivered-rate ratio") for w in j["warnings"]), (o0, j["warnings"])
        assert line["cpu_baseline_O0"]["value"] == o0["value"] and line["cpu_baseline_O0"]["rate_all_reps"] == o0["rate_all_reps"]
    # the independent second number explains itself: three repetitions, each with its wall clock
    # accounted for, and a ratio to the timed run that is either clean or named in `warnings`
    if j["cpu_baseline"]["kind"] == "reference":
        p = j["cpu_baseline_port"]
        assert p["reps"] == 3 and p["exact"] and len(p["rate_all_reps"]) == 3 and len(p["diagnostics_all_reps"]) == 3
        for d in p["diagnostics_all_reps"]:
            assert {"wall_s", "server_busy_frac", "server_run_delay_frac", "ack_latency_us", "progress_gaps", "loadavg_before",
                    "cgroup_throttled_periods", "sender_receiver_busy_frac"} <= set(d)
        if not 0.9 <= p["ratio_to_timed_run"] <= 1.1:
            assert any("ratio" in w for w in j["warnings"]), j["warnings"]
        assert line["cpu_baseline_port"]["ratio_to_timed_run"] == p["ratio_to_timed_run"]
        assert line["cpu_baseline_port"]["rate_all_reps"] == p["rate_all_reps"]
    for k, d in enumerate([j["diagnostics"]] + j.get("cpu_baseline_port", {}).get("diagnostics_all_reps", [])):
        if d["server_busy_frac"] < 0.9:          # a stalled leg must be attributed in the line, never silent
            assert any("harness stall" in w for w in j["warnings"]), (k, d, j["warnings"])
    assert j["host"]["placement"]["policy"] in ("quiet", "first") and j["diagnostics"]["placement"]["talker_cpus"]
    assert j["host"]["receiver_threads_per_replica"] >= 2 or any("client-bound" in w for w in j["warnings"])
    print("\n[bench line]", out[0])


def test_device_launch_floor_is_recorded():
    """In a child process: the pytest process itself, which boots talkers and load generators in every other test of
    this tier, never initialises HIP."""
    sys.path.insert(0, str(REPO))
    import bench
    floor = bench.device_floor_in_child()
    assert floor, "gpu-marked test needs a GPU"
    assert floor["kernel_launch_plus_sync_us"] > 0
    assert "torch" not in sys.modules or not sys.modules["torch"].cuda.is_initialized()
    print("\n[device floor]", json.dumps(floor))


def test_replicas_never_open_a_gpu(tmp_path):
    """"no GPU is touched by any replica", literally.  libdrm announces every amdgpu device a process
    opens on this box with a complaint about a missing amdgpu.ids file; two replicas must run without one."""
    p = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--lines-per-step", "20"],
                       env={**os.environ, "NUTS_BENCH_FULL_RECORD": str(tmp_path / "full.json")},
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-800:]
    j = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert j["n_gpus"] == 2 and j["delivered"] == j["expected_delivered"] == 2 * 20 * 999
    assert b"amdgpu" not in p.stderr, p.stderr.decode(errors="replace")[-800:]
