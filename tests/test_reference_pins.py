"""The literal constants of the reference's decision path and its one in-source golden (the quorum-slot comment table,
member/Leadership.java:121-126) — extracted from the reference SOURCE TEXT by tools/make_reference_pins.py into
tests/golden/reference_pins.json — against the oracle. Whenever the reference checkout is present (this container, not
the GPU box) the committed JSON is also compared with a fresh extraction, so the fixture cannot drift from the source.
CPU only."""
import itertools
import json
import math
import os
import sys

import numpy as np
import pytest

from rafting_amd import abi
from tests import oracle_lib
from tests.helpers import Sim, simple_log

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PINS = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_pins.json")))
REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference checkout not present on this machine")
def test_fixture_equals_a_fresh_extraction_from_the_reference_source():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import make_reference_pins
    finally:
        sys.path.pop(0)
    assert make_reference_pins.extract(REFERENCE) == PINS


def test_quorum_slot_table():
    """`// N = 5, major = 3 : |x|x|o|x|*|`: the `o` cell is the slot of the ascending follower matchIndex values that
    Leadership.State.majorIndices returns as the majority index; slot 0 is the "replicated everywhere" index."""
    assert PINS["major_index_expr"] == "matchIndices.length / 2"
    for n, row in PINS["quorum_table"].items():
        f = row["followers"]
        assert f == int(n) - 1 and row["quorum_slot_of_sorted_followers"] == f // 2
        assert row["major"] == int(n) // 2 + 1                           # the table's own "major" column = majority()
        values = [10 * (k + 1) for k in range(f)]
        for perm in itertools.islice(itertools.permutations(values), 200):
            full, major = oracle_lib.major_indices(list(perm))
            assert (full, major) == (values[0], values[row["quorum_slot_of_sorted_followers"]])


def test_majority():
    assert PINS["majority_expr"] == "cluster.size() / 2 + 1"
    for cluster in range(abi.MIN_CLUSTER, abi.MAX_CLUSTER + 1):
        # a Candidate that has its own vote becomes Leader on the reply that brings the tally to majority(), not before
        s = Sim(oracle_lib.OracleTable(1, cluster, 0, True)).load(role=abi.CANDIDATE, term=3, voted_for=0, role_epoch=2)
        need = cluster // 2 + 1
        for votes, peer in enumerate(range(1, cluster), start=2):
            r = s.rv_reply(peer, 3, True, 2)
            assert (r.role == abi.LEADER) == (votes >= need), (cluster, votes)
            if r.role == abi.LEADER:
                break


def _leader(cluster, last, next_index):
    peers = [(0, next_index, 0, 0, 0)] * (cluster - 1)
    return oracle_lib.OracleTable(1, cluster, 0, True), dict(role=abi.LEADER, term=5, voted_for=0, role_epoch=3, repl_prepared=1,
                                                             log=simple_log(last, 5), peers=peers)


def test_replicate_limits():
    """Leader.replicateLog: fetchLimit = REPLICATE_LIMIT >> (heartbeat ? 1 : 0); a follower is skipped while
    requestInFlight > IN_FLIGHT_LIMIT / (heartbeat ? 10 : 1)."""
    limit, in_flight = PINS["REPLICATE_LIMIT"], PINS["IN_FLIGHT_LIMIT"]
    assert PINS["in_flight_gate_expr"] == "state.requestInFlight > requestLimit"
    t, state = _leader(3, 10 * limit, 2)
    s = Sim(t).load(**state)
    for heartbeat in (0, 1):
        fetch = limit >> (PINS["heartbeat_fetch_shift"] if heartbeat else 0)
        _, send = t.replicate(heartbeat=heartbeat)
        assert send[0]["count"].tolist() == [fetch, fetch] and send[0]["last_index"].tolist() == [1 + fetch] * 2
        gate = in_flight // (PINS["heartbeat_in_flight_divisor"] if heartbeat else 1)
        _, send = t.replicate(heartbeat=heartbeat, in_flight=[[gate, gate + 1]])
        assert send[0]["kind"].tolist() == [abi.SEND_APPEND, abi.SEND_GATED]
    assert s.state().role == abi.LEADER


def test_rejection_step_is_the_rounded_log():
    assert PINS["rejection_step_expr"] == "Math.round(Math.log(Math.E + recentRejection))"
    rng = np.random.default_rng(1)
    for r in list(range(0, 3000)) + rng.integers(0, 2 ** 31 - 1, 2000).tolist() + [2 ** 31 - 1]:
        assert oracle_lib.lib().orc_rejection_step(int(r)) == math.floor(math.log(math.e + r) + 0.5)   # Math.round = floor(x + 0.5)


def test_raft1_xml_timing():
    """RaftConfig: heartbeatInterval = round(heartbeat * tick), electionTimeout uniform in [E, 2E] with
    E = round(election * tick) (`nextInt(E, 2 * E + 1)`), the values of the reference's own 3-node demo."""
    x = PINS["raft1_xml"]
    assert PINS["election_timeout_expr"] == "electTimeout, 2 * electTimeout + 1"
    E, H = round(x["election_ticks"] * x["tick_ms"]), round(x["heartbeat_ticks"] * x["tick_ms"])
    assert (E, H, x["cluster_size"], x["pre_vote"]) == (900, 300, 3, True)
    assert round(x["broadcast_ticks"] * x["tick_ms"]) == 150 and (x["avail_critical_point"], x["recovery_cool_down_ms"]) == (1, 100)
    G = 4096
    t = oracle_lib.OracleTable(G, x["cluster_size"], 0, x["pre_vote"])
    t.timers_configure(E, H, 2024)
    t.timers_arm(5000)
    d = t.timers_read() - 5000
    assert d.min() >= E and d.max() <= 2 * E and d.min() < E + 20 and d.max() > 2 * E - 20      # the whole closed range is used
    assert len(np.unique(d)) > E // 2
