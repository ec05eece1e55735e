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


def test_two_populations_times_two_chromosome_shards_with_migration_oracle_backend():
    """world_size 4 over gloo: 2 populations x 2 chromosome shards (fixture mig3c from the real reference): A/D all-reduce inside
    each population, phenotype scaling from the all-reduced totals, migration between the ranks holding the same chromosomes"""
    res = dist_worker.launch("oracle", world=4, target=dist_worker.run_split_migration)
    assert sorted(r for r, _ in res) == [0, 1, 2, 3]
    for r, msg in res:
        assert msg == "ok", f"rank {r}: {msg}"


def test_config4_in_miniature_22_chromosomes_assortative_mating_split_and_migration_oracle_backend():
    """fixture c4mini (real reference): the 22 autosomes of Recom.Map.b37.50KbDiff, assortative mating, mutation map, two
    populations with migration; each population split 11 | 11 chromosomes over two ranks"""
    res = dist_worker.launch("oracle", world=4, target=dist_worker.run_split_migration_c4mini)
    for r, msg in res:
        assert msg == "ok", f"rank {r}: {msg}"


def test_migration_beyond_fixture_size_host_logic_rehearsal():
    """the worker of the GPU test `test_two_rank_migration_beyond_fixture_size_with_full_plane_verification`, with the CPU oracle in
    both roles: checks the worker's own bookkeeping (who moves, sex arrays, collective order) where there is no GPU"""
    res = dist_worker.launch("oracle", world=2, target=dist_worker.run_migration_at_scale)
    for r, msg in res:
        assert msg == "ok", f"rank {r}: {msg}"
