"""tests/golden_pin.py (what the `oracle` fixture runs on a GPU box before any device comparison) works and compares what it says."""
import numpy as np
import pytest
import golden_pin
import util


def test_pin_compares_the_golden_fixtures(oracle):
    n = golden_pin.pin(oracle)
    assert n >= 150 and util.COUNTS["golden_fields"] == n


def test_pin_fails_on_a_wrong_oracle(oracle, monkeypatch):
    """an oracle whose advection is off by one ulp in one cell does not get through"""
    real = oracle.advect

    def bent(scheme, q, *a, **k):
        real(scheme, q, *a, **k)
        q.reshape(-1)[q.size // 2] = np.nextafter(q.reshape(-1)[q.size // 2], np.float32(np.inf))
        return q
    monkeypatch.setattr(oracle, "advect", bent)
    with pytest.raises(AssertionError):
        golden_pin.pin(oracle)
