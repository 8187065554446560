"""Pins the CPU oracle (oracle/raft_oracle.c) with known answers hand-derived from the reference
source. The reference's own tests hold no vector for this path, so these KATs —
each citing the lines it was read from — are the pin; the only in-source golden is the quorum
table in member/Leadership.java:121-126 (test_major_indices_golden_table)."""
import itertools
import math

import pytest

from rafting_amd import abi
from tests import kat_scenarios, oracle_lib
from tests.helpers import C, F, L


def mk(groups, cluster, self_slot, pre_vote):
    return oracle_lib.OracleTable(groups, cluster, self_slot, pre_vote)


@pytest.mark.parametrize("scenario", kat_scenarios.SCENARIOS, ids=lambda f: f.__name__)
def test_scenario(scenario):
    scenario(mk)


def test_major_indices_golden_table():
    """member/Leadership.java:121-126: N = followers+1 nodes, `o` marks the sorted slot returned as
    the majority index (N=2:|o|*|  N=3:|x|o|*|  N=4:|x|o|x|*|  N=5:|x|x|o|x|*|  N=6:|x|x|o|x|x|*|
    N=7:|x|x|x|o|x|x|*|), `[0]` is the index replicated everywhere."""
    table = {1: 0, 2: 1, 3: 1, 4: 2, 5: 2, 6: 3}      # followers -> slot of `o` (0-based, ascending)
    # beyond the comment table (N = 8 .. 15, ABI 5): the rule under it, `int majorIndex = matchIndices.length / 2` (member/Leadership.java:127)
    table.update({f: f // 2 for f in range(7, 15)})
    for f, slot in table.items():
        for perm in itertools.islice(itertools.permutations(range(10, 10 + f)), 50):
            full, major = oracle_lib.major_indices(list(perm))
            assert full == 10 and major == 10 + slot, (f, perm)
    # KAT-1 values from SURVEY.md §8c
    assert oracle_lib.major_indices([5]) == (5, 5)
    assert oracle_lib.major_indices([7, 3]) == (3, 7)
    assert oracle_lib.major_indices([9, 1, 5]) == (1, 5)
    assert oracle_lib.major_indices([8, 2, 6, 4]) == (2, 6)
    assert oracle_lib.major_indices([5, 4, 3, 2, 1]) == (1, 3)
    assert oracle_lib.major_indices([6, 5, 4, 3, 2, 1]) == (1, 4)


def _java_step(r):
    """Math.round(Math.log(Math.E + r)) for an int r (member/Leadership.java:105)."""
    x = math.e + r
    if x <= 0:
        return 0                     # log -> NaN / -inf... Math.round(NaN) = 0; ln(e-2) < 0.5 rounds to 0 as well
    return math.floor(math.log(x) + 0.5)


def test_rejection_step_table():
    for r in range(0, 5000):
        assert oracle_lib.rejection_step(r) == _java_step(r), r
    # every threshold of the integer table, both sides, and the margin to the rounding boundary
    k = 2
    while True:
        lo = math.ceil(math.exp(k - 0.5) - math.e)
        if lo > 2**31 - 1:
            break
        assert oracle_lib.rejection_step(lo) == k == _java_step(lo), (k, lo)
        assert oracle_lib.rejection_step(lo - 1) == k - 1 == _java_step(lo - 1), (k, lo)
        for r in (lo - 1, lo):       # a libm 1-ulp difference cannot flip the rounding
            assert abs(math.log(math.e + r) - (k - 0.5)) > 1e-10
        k += 1
    assert k == 22 and oracle_lib.rejection_step(2**31 - 1) == 21
    assert oracle_lib.rejection_step(-1) == 1 and oracle_lib.rejection_step(-2) == 0
    assert oracle_lib.rejection_step(-(2**31)) == 0


def test_is_better_truth_table():
    """KAT-5, member/Membership.java:74-108: (new role, cmp(new term, cur term), cur role)."""
    me, other = 0, 1
    for nr, cr in itertools.product((F, C, L), repeat=2):
        assert oracle_lib.is_better((nr, 6, me), (cr, 5, other)) is True       # :79-81 greater term
        assert oracle_lib.is_better((nr, 4, me), (cr, 5, other)) is False
    same = {
        (L, C): True,                          # :84-86 candidate wins the election
        (L, F): -abi.A_LEADER_UNCHANGED,       # :87-89
        (F, C): True, (F, L): True,            # :92 follower first
        (C, F): False, (C, L): False,
        (L, L): False,                         # :95-97
        (F, F): True,                          # :98-100
    }
    for (nr, cr), want in same.items():
        assert oracle_lib.is_better((nr, 5, me), (cr, 5, me)) == want, (nr, cr)
    assert oracle_lib.is_better((C, 5, me), (C, 5, me)) is False               # :107
    assert oracle_lib.is_better((C, 5, me), (C, 5, other)) == -abi.A_CAND_BALLOT   # :103-105


def test_config_timing_constants():
    """KAT-11 (config sanity only): src/test/resources/raft1.xml:10-13 with RaftConfig.java:187-198 —
    tick 300 ms, heartbeat x1, election x3 randomised in [E, 2E], broadcast x0.5. The replay contract
    leans on broadcast < heartbeat < election (RaftConfig.java:116-118): a reply is deliverable only
    within the broadcast timeout, a node cannot run two elections inside it."""
    tick, heartbeat, election, broadcast = 300, 1.0, 3.0, 0.5
    assert round(heartbeat * tick) == 300 and round(broadcast * tick) == 150
    e = round(election * tick)
    assert (e, 2 * e) == (900, 1800)
    assert broadcast < heartbeat < election
