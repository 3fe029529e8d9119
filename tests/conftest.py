"""pytest configuration: markers, import paths, and one-time build of the test infrastructure."""
from __future__ import annotations

import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box only)")
    config.addinivalue_line("markers", "reference: needs oracle/_ref/nuts333 (built where /root/reference exists)")
    config.addinivalue_line("markers", "host_only: in a BOTH_TIERS module, run this test in the host tier only")


# ---------------------------------------------------------------------------------------------------------
# Two tiers from one test body.  The driver runs `-m "not gpu"` here and `-m gpu` on the MI355X box; a marker
# expression cannot select the same item for both, so modules that set ``BOTH_TIERS = True`` get every test
# parametrised over ``tier``: the "host" instance is unmarked, the "gpubox" instance carries the gpu marker.  This
# is how the real parity suite (all golden sessions, the transducer vectors) runs on the box's host as well.
@pytest.fixture(autouse=True)
def tier(request):
    return getattr(request, "param", "host")


def pytest_generate_tests(metafunc):
    if getattr(metafunc.module, "BOTH_TIERS", False):
        if metafunc.definition.get_closest_marker("host_only"):
            # e.g. the restatement's NUTS_PORT_FAST mode: not a SURVEY section-8 row, no place in the box tier
            metafunc.parametrize("tier", [pytest.param("host")], indirect=True)
            return
        metafunc.parametrize("tier", [pytest.param("host"), pytest.param("gpubox", marks=pytest.mark.gpu)], indirect=True)


@pytest.fixture(scope="session", autouse=True)
def built():
    """Compile the restatement and the load generator (seconds; gcc only)."""
    subprocess.run(["make", "-s", "-C", str(REPO / "oracle"), "port"], check=True)
    from nuts333_amd import workloads
    workloads.build_loadgen()


@pytest.fixture(scope="session")
def port_binary(built):
    from nuts333_amd.talker import PORT_BINARY
    assert PORT_BINARY.exists()
    return PORT_BINARY


@pytest.fixture(scope="session")
def ref_binary():
    from nuts333_amd.talker import REF_BINARY, reference_expected_but_missing
    lost = reference_expected_but_missing()
    if lost:          # the marker of __graft_entry__.build() says it was built for this snapshot: fail, do not skip
        pytest.fail(lost)
    if not REF_BINARY.exists():
        pytest.skip("oracle/_ref/nuts333 not built (no /root/reference on this machine)")
    return REF_BINARY
