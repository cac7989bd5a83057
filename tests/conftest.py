"""pytest configuration: markers, paths, and on-demand build of the test-only CPU oracle."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer CPU case")


def free_port():
    """a loop-back TCP port nobody listens on right now (multi-process tests rendezvous there: a fixed number is a
    spurious red on a shared box)"""
    import random
    import socket
    # below the kernel's ephemeral range (32768-60999): a port handed out by bind(0) can be taken as the SOURCE port of somebody's outgoing
    # connection (the gloo meshes of 8-rank tests open dozens) between this probe and the rendezvous server's listen() — seen as EADDRINUSE
    for _ in range(64):
        p = random.randint(20000, 29999)
        with socket.socket() as so:
            try:
                so.bind(("127.0.0.1", p))
                return p
            except OSError:
                continue
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


@pytest.fixture
def port():
    return free_port()


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle is test infrastructure (oracle/): compile it once per session if it is missing/stale."""
    subprocess.run(["make", "-s", "-C", os.path.join(REPO, "oracle"), "oracle"], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    yield
