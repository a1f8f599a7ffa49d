"""NOT part of the suite: a test file whose first_contact child does not survive -- one test passes, the next one hangs (MODE=hang)
or takes the process down with a fault (MODE=fault), a third never gets its turn.  Run by tests/test_bench_host.py."""
import os
import signal
import time

import pytest

pytestmark = pytest.mark.first_contact(timeout=4)


def test_fatal_a_passes():
    pass


def test_fatal_b_takes_the_child_down():
    if os.environ.get("PROBE_MODE") == "hang":
        time.sleep(600)
    os.kill(os.getpid(), signal.SIGSEGV)


def test_fatal_c_never_runs():
    pass
