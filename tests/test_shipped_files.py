"""BASELINE configurations #1 and #5 on the reference's own shipped files (container only; see tests/shipped_tree.py).

``reference``-marked: needs /root/reference and oracle/_ref/nuts333; skipped on the GPU box, where neither the
reference tree nor anything copied from it exists.
"""
from __future__ import annotations

import statistics

import pytest

import shipped_tree as st
from nuts333_amd import workloads

pytestmark = pytest.mark.reference


@pytest.fixture(autouse=True)
def _needs_the_shipped_files(ref_binary):
    if not st.available():
        pytest.skip("/root/reference is not mounted here")
    if not (st.ports_free(st.PORTS_T1) and st.ports_free(st.PORTS_T2)):
        pytest.skip("the shipped configs' fixed ports 7000-7002 / 5000-5002 are in use")


def _cpu_us_per_input_line(res) -> float:
    return res["servers"][0]["cpu_ns"] / 1e3 / res["input_lines"]


def test_config1_on_the_shipped_config_matches_the_generated_tree(ref_binary):
    """"Boot talker with datafiles/config, 1 local telnet client, .say in lounge": exact counts, two writes per input
    line (Fred.D has the prompt on), and the same server cost per input line as the generated tree with an
    equally-flagged account -- medians of five interleaved runs, within 25 % (this VM's run-to-run spread is 10-15 %)."""
    shipped, generated = [], []
    for _ in range(5):
        a = st.config1_shipped(lines=4000, warmup=300)
        b = workloads.config1(lines=4000, warmup=300, prompt=1, binary=ref_binary)
        for r in (a, b):
            assert r["ok"] and r["exact"] and r["deliveries"] == 0 and r["lines_total"] == 2 * 4000
            s = r["servers"][0]
            assert (s["read_syscalls"], s["write_syscalls"]) == (4000, 8000)
        assert a["bytes_total"] == b["bytes_total"]               # same bytes on the wire: "You say: ..." + "<HH:MM, hh:mm, Fred>"
        shipped.append(_cpu_us_per_input_line(a)); generated.append(_cpu_us_per_input_line(b))
    ms, mg = statistics.median(shipped), statistics.median(generated)
    print(f"\n[config1] server CPU per input line: shipped tree {ms:.2f} us, generated tree {mg:.2f} us ({ms / mg - 1:+.1%})")
    assert abs(ms / mg - 1) < 0.25, (shipped, generated)


def test_config5_on_the_shipped_configs_carries_shouts_across_the_link(ref_binary):
    """"Two-server netlink (datafiles/config + config2) with cross-link .shout traffic": config2 boots once its line 11
    is fixed in the temporary copy; Fred travels and shouts; every shout is one ACT frame out and one MSG..EMSG + one
    PRM frame back (measured from the talkers' write(2) counts), and reaches the listener on the far talker."""
    res = st.config5_shipped(lines=60)
    assert res["ok"] and res["exact"] and res["deliveries"] == 60 and all(res["servers_alive_after"])
    nl = res["netlink"]
    assert nl["writes_t1_to_t2"] == nl["expected_act_frames"] == 60
    assert nl["writes_t2_to_t1"] == nl["expected_msg_frames"] + nl["expected_prm_frames"] == 120


def test_shipped_config2_does_not_boot_unmodified(tmp_path, ref_binary):
    """`logging YES` (datafiles/config2:11) is rejected by the 3.3.3 parser (nuts333.c:599-607)."""
    from nuts333_amd.talker import Talker
    st.populate(tmp_path)
    with pytest.raises(RuntimeError, match="Unknown INIT option on line 11"):
        Talker(ref_binary, tmp_path, config_name="config2").start(timeout=5)
