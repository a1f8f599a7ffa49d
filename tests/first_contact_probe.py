"""NOT part of the suite (the file name does not match test_*.py): `first_contact` tests with known outcomes (tests/first_contact_probe_fatal.py: a file whose child hangs / dies), run by
tests/test_bench_host.py::test_first_contact_tests_run_isolated_and_never_colour_the_suite in a child pytest to check the
isolation mechanism of tests/conftest.py without a GPU."""
import os
import signal
import time

import pytest


@pytest.mark.first_contact(timeout=60)
def test_probe_passes():
    assert os.environ.get("RLIPV2_TEST_FIRST_CONTACT_CHILD") == "1"      # the body only ever runs in the child


@pytest.mark.first_contact(timeout=60)
def test_probe_fails():
    assert 1 + 1 == 3


@pytest.mark.first_contact(timeout=5)
@pytest.mark.parametrize("n", [1, 2])
def test_probe_parametrized(n):
    assert n == 1


def test_probe_ordinary_test_after_the_others():
    assert "RLIPV2_TEST_FIRST_CONTACT_CHILD" not in os.environ
