"""The CPU oracle (oracle/helios_oracle.c) against the committed golden vectors -- the pin that
travels to machines where the reference tree (and hence oracle/_ref) is absent."""
import pytest

import golden_checks as gc


@pytest.mark.parametrize("name", gc.CHAIN_NAMES)
def test_oracle_chain_golden(port, name):
    gc.check_chain(port, name)


def test_oracle_mixing_golden(port):
    gc.check_mixing(port)


def test_all_fixtures_present():
    assert len(gc.CHAIN_NAMES) == 9


@pytest.mark.parametrize("name", gc.MATRIX_NAMES)
def test_oracle_matrix_golden(port, name):
    gc.check_matrix(port, name)
