#!/usr/bin/env python3
"""bench.py -- the reference's headline metric, measured the only place it exists: on the host CPU.

BASELINE.json: metric = "CPU broadcast lines/sec at N synthetic telnet clients"; north_star =
the reference (NUTS 3.3.3, a select()-driven telnet talker) has no data-parallel hot path, no
HIP kernel applies, record the CPU baseline.  So this bench does NOT time a GPU kernel -- there
is none in this repository -- it times the talker itself (the unmodified reference build when
oracle/_ref/nuts333 is present, else our restatement oracle/_build/talker_port) under the
closed-loop load generator, and says so in every field.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line.
  * a "step" is one batch of ``--lines-per-step`` input lines pushed through the hot path
    (read -> parse -> say -> per-recipient transduce + write(2)); the default K x L is exactly
    BASELINE configs[1]: 10 clients in one room, client 0 says 20,000 lines, 9 recipients each;
  * W warm-up batches run untimed first, inside the same session;
  * the timed region is the load generator's own window (first send -> last expected
    delivery), bracketed by a barrier on both sides when N > 1; MAX over ranks;
  * N > 1 = N independent talker replicas, one per rank ("replicas only"): the
    path does not shard and there is no collective on it.  ``--gpus`` only counts replicas;
    no GPU is used by any of them.
  * ``roofline`` is the host system-call ceiling, not HBM/MFMA: achieved = write(2)-bearing
    lines per second through the talker; peak = the same number of 67-byte writes issued
    by a loop that does nothing else (loadgen --probe-fanout), priced by its CPU time per write.
  * ``cpu_baseline`` is the same run by construction (the CPU path is the only path).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

from nuts333_amd import workloads  # noqa: E402
from nuts333_amd.talker import PORT_BINARY, REF_BINARY  # noqa: E402

METRIC = "delivered_broadcast_lines_per_s"
UNIT = "lines/s"


def ensure_built() -> None:
    """Build what can be built here; never silently substitute."""
    if not PORT_BINARY.exists():
        subprocess.run(["make", "-s", "-C", str(REPO / "oracle"), "port"], check=True)
    workloads.build_loadgen()


def partition_cpus(rank: int, world: int) -> list[int]:
    cpus = sorted(os.sched_getaffinity(0))
    per = len(cpus) // world
    if per < 2:
        return cpus          # oversubscribed: no pinning possible, the numbers will say so
    # at most 8 cores per replica: 1 for the talker, the rest for the synthetic clients
    per = min(per, 8)
    return cpus[rank * per:(rank + 1) * per]


def run_workload(name: str, total_lines: int, warm_lines: int, binary: Path, pin: bool) -> dict:
    if name == "config2":
        return workloads.config2(lines=total_lines, warmup=warm_lines, binary=binary, pin=pin)
    if name == "config2_all":
        return workloads.config2(lines=total_lines, warmup=warm_lines, all_send=True, binary=binary, pin=pin)
    if name == "config4":
        return workloads.config4(lines=total_lines, warmup=warm_lines, binary=binary, pin=pin)
    if name == "config1":
        return workloads.config1(lines=total_lines, warmup=warm_lines, binary=binary, pin=pin)
    raise SystemExit(f"unknown workload {name}")


def device_floor() -> dict | None:
    """What merely *touching* an MI355X costs, for the record: one tiny kernel + sync, and a
    host->device->host round trip the size of the largest broadcast (1000 x 69 B).  Torch only;
    there is no custom kernel to time.  None when no GPU is visible."""
    try:
        import torch
    except Exception:
        return None
    if not torch.cuda.is_available():
        return None
    dev = torch.device("cuda:0")
    x = torch.zeros(64, device=dev, dtype=torch.uint8)
    host_in = torch.zeros(64, dtype=torch.uint8).pin_memory()
    host_out = torch.zeros(69_000, dtype=torch.uint8).pin_memory()
    big = torch.zeros(69_000, device=dev, dtype=torch.uint8)
    for _ in range(50):
        x.add_(1); torch.cuda.synchronize()
    n = 500
    t0 = time.perf_counter()
    for _ in range(n):
        x.add_(1); torch.cuda.synchronize()
    launch_sync_us = (time.perf_counter() - t0) / n * 1e6
    t0 = time.perf_counter()
    for _ in range(n):
        x.copy_(host_in, non_blocking=True); big.add_(1); host_out.copy_(big, non_blocking=True); torch.cuda.synchronize()
    round_trip_us = (time.perf_counter() - t0) / n * 1e6
    out = {"device": torch.cuda.get_device_name(0), "kernel_launch_plus_sync_us": round(launch_sync_us, 2),
           "h2d_64B_kernel_d2h_69KB_sync_us": round(round_trip_us, 2),
           "note": "torch elementwise kernel; no custom HIP kernel exists in this repo"}
    try:
        # the same kernel replayed from a captured hipGraph: takes torch's per-op dispatch out of the figure
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            x.add_(1)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(g):
            x.add_(1)
        for _ in range(50):
            g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay(); torch.cuda.synchronize()
        out["graph_replay_plus_sync_us"] = round((time.perf_counter() - t0) / n * 1e6, 2)
    except Exception as e:  # graphs unavailable: the eager figure stands
        out["graph_replay_plus_sync_us"] = None
        out["graph_note"] = repr(e)[:120]
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="number of independent talker replicas (no GPU is used)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--lines-per-step", type=int, default=2000)
    ap.add_argument("--workload", default="config2", choices=["config1", "config2", "config2_all", "config4"])
    ap.add_argument("--binary", default="auto", choices=["auto", "reference", "port"])
    ap.add_argument("--no-extras", action="store_true", help="skip the syscall probe / port comparison / device floor")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        # not under torchrun: start the replicas ourselves, same code path as the driver's launch
        ensure_built()
        port = workloads.free_ports(1)[0]
        procs = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, __file__] + sys.argv[1:], env=env))
        return max(p.wait() for p in procs)

    dist = None
    if world > 1:
        import torch.distributed as dist  # gloo: the replicas exchange two scalars, nothing on the data path
        # gloo announces its mesh on stdout; the contract is ONE JSON line there, so lend it stderr
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
        if rank == 0:
            ensure_built()
        dist.barrier()
        os.sched_setaffinity(0, set(partition_cpus(rank, world)))
    else:
        ensure_built()
    pin = len(os.sched_getaffinity(0)) >= 2
    binary, kind = workloads.pick_binary(args.binary)

    total = args.steps * args.lines_per_step
    warm = args.warmup * args.lines_per_step

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    def barrier():
        """Barrier + device sync on both sides of the timed region, as the contract asks.  No replica
        queues GPU work, so the sync is a formality; each rank only ever touches ITS OWN device
        (LOCAL_RANK), and none at all if there are fewer devices than ranks."""
        if dist is not None:
            dist.barrier()
        try:
            import torch
            if torch.cuda.device_count() > local_rank:     # device_count() does not initialise the GPU
                torch.cuda.set_device(local_rank)
                torch.cuda.synchronize()
        except Exception:
            pass

    barrier()
    t_outer0 = time.perf_counter()
    res = run_workload(args.workload, total, warm, binary, pin)
    t_outer1 = time.perf_counter()
    barrier()
    if not res["exact"]:
        print(json.dumps({"error": "delivered != expected", "result": res}), file=sys.stderr)
        return 1

    wall = res["wall_s"]
    deliveries = res["deliveries"]
    written = res["lines_total"]
    if dist is not None:
        import torch
        t = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall_max = float(t[0])
        c = torch.tensor([float(deliveries), float(written)], dtype=torch.float64)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        deliveries_all, written_all = float(c[0]), float(c[1])
    else:
        wall_max, deliveries_all, written_all = wall, float(deliveries), float(written)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return 0

    value = deliveries_all / wall_max
    srv = res["servers"][0]
    out = {
        "metric": METRIC, "value": round(value, 1), "unit": UNIT,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall_max / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": res["workload"], "lines_per_step": args.lines_per_step,
                   "implementation": kind, "binary": str(binary.relative_to(REPO)),
                   "parallelism": f"{world} independent talker replica(s); no GPU on the path",
                   "payload_bytes": workloads.PAYLOAD_LEN, "bytes_per_delivered_line": round(res["bytes_per_line"], 2)},
        "gpu_used": False,
        "classification": "misclassified / not graft-eligible (BASELINE.json north_star); CPU baseline only",
        "delivered": int(deliveries_all), "expected_delivered": res["expected_deliveries"] * world if world > 1 else res["expected_deliveries"],
        "input_lines_per_s": round(res["input_lines_per_s"], 1),
        "ack_latency_us": res["ack_latency_us"],
        "server_cpu_us_per_written_line": round(srv["cpu_us_per_written_line"], 3),
        "server_user_frac": srv["user_frac"], "server_busy_frac": round(srv["busy_frac"], 3),
        "outer_wall_s": round(t_outer1 - t_outer0, 3), "login_s": res["login_s"],
    }
    baseline = {"value": round(res["delivered_lines_per_s"], 1), "unit": UNIT, "cores": 1, "kind": kind,
                "sample": f"the timed run itself: {res['input_lines']} input lines, {res['deliveries']} deliveries, one replica"}
    roofline = None
    if not args.no_extras and world == 1:
        cpus = sorted(os.sched_getaffinity(0))
        recipients = max(1, res["expected_deliveries"] // max(1, res["input_lines"]))
        rounds = max(200, 200_000 // recipients)
        cmd = [str(workloads.LOADGEN_BIN), "--probe-fanout", str(round(res["bytes_per_line"])), str(recipients), str(rounds)]
        if len(cpus) >= 2:
            cmd += [str(cpus[0]), str(cpus[1])]
        probe = json.loads(subprocess.run(cmd, check=True, stdout=subprocess.PIPE).stdout)
        achieved = written_all / wall_max
        # the talker is one thread on one core: its ceiling is what that core can issue when it does
        # nothing but write(2) -- CPU time per write, not the probe's wall time (which includes its reader)
        peak = 1e9 / probe["cpu_ns_per_write"]
        roofline = {"bound": "host-syscall", "achieved": round(achieved, 1), "peak": round(peak, 1),
                    "unit": "write(2)/s on one core", "frac": round(achieved / peak, 3), "traffic": None, "probe": probe,
                    "note": "no HBM/MFMA roofline applies: no device kernel exists. peak = 1e9 / (CPU ns per closed-loop "
                            "write(2) of the same size to the same number of loopback sockets, zero user-space work)"}
        if kind == "reference" and PORT_BINARY.exists():
            p = workloads.config2(lines=5000, warmup=1000, binary=PORT_BINARY, pin=pin)
            out["cpu_baseline_port"] = {"value": round(p["delivered_lines_per_s"], 1), "unit": UNIT, "cores": 1, "kind": "port",
                                        "sample": "config2, 5000 input lines", "exact": p["exact"]}
        out["device_floor"] = device_floor()
    out["roofline"] = roofline
    out["cpu_baseline"] = baseline
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
