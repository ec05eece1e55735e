"""world_size-2 gloo tests on CPU of the multi-process path: population-per-rank sharding with the
migration all-to-all (host logic, ordering, record exchange) against the reference fixture `mig2`,
and the bench's N>1 timing collective."""
import numpy as np

from tests import dist_worker


def test_two_rank_migration_matches_reference_fixture_oracle_backend():
    res = dist_worker.launch("oracle")
    assert sorted(r for r, _ in res) == [0, 1]
    for r, msg in res:
        assert msg == "ok", f"rank {r}: {msg}"


def test_locus_split_population_two_ranks_oracle_backend():
    """C2 of SURVEY.md 2.1: one population split along chromosomes over two ranks, per-chromosome A/D all-reduced and
    summed in order == the unsplit reference run (fixture ex1mut), world_size 2 over gloo"""
    res = dist_worker.launch("oracle", target=dist_worker.run_locus_split)
    assert sorted(r for r, _ in res) == [0, 1]
    for r, msg in res:
        assert msg == "ok", f"rank {r}: {msg}"
