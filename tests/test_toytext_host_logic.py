"""The HOST side of the toy_text adapters on the GPU-less box: gym_amd._native.Tab / Blackjack replaced by the oracle-backed FakeTab / FakeBlackjack
(tests/oracle_engine.py), then the very test bodies the GPU suite runs on the device (tests/test_gpu_toytext.py,
test_gpu_toytext_stats.py, test_gpu_toytext_normalize.py): spaces, infos and their dtype quirks, errors, Taxi's helpers through call(),
pickling, the statistics and normalisation wrappers against the reference's own runs (goldens)."""
import numpy as np
import pytest


@pytest.fixture
def fake_tab(monkeypatch):
    from gym_amd import _native
    from oracle_engine import FakeBlackjack, FakeTab

    monkeypatch.setattr(_native, "Tab", FakeTab)
    monkeypatch.setattr(_native, "Blackjack", FakeBlackjack)


def test_tabular_vector_env_contract_on_the_oracle_backed_handle(fake_tab):
    import test_gpu_toytext as t

    t.test_hip_tabular_vector_env_contract()
    t.test_taxi_helpers_through_call()
    for gid in ("FrozenLake-v1", "Taxi-v3"):
        t.test_pickle_round_trip_continues_identically(gid)


def test_blackjack_vector_env_contract_on_the_oracle_backed_handle(fake_tab):
    import test_gpu_blackjack as t

    t.test_hip_blackjack_vector_env_contract()
    t.test_pickle_round_trip_continues_identically()


@pytest.mark.parametrize("tag", ["FrozenLake-v1", "Taxi-v3", "Blackjack-v1"])
def test_statistics_wrappers_over_the_tabular_adapters_replay_the_reference(fake_tab, tag):
    import test_gpu_toytext_stats as t

    t.test_wrappers_over_the_toy_text_adapters_report_what_the_reference_reports(tag)


@pytest.mark.parametrize("tag", ["FrozenLake-v1", "Taxi-v3", "Blackjack-v1"])
def test_per_sub_env_normalize_reward_over_the_tabular_adapters_bit_for_bit(fake_tab, tag, monkeypatch):
    import test_gpu_toytext_normalize as t

    t.test_per_sub_env_normalize_reward_over_the_toy_text_engines_bit_for_bit(tag, False, monkeypatch)
