"""CPU checks of the replay workload model (no GPU): it must be deterministic, partition independent,
and — replayed through the oracle — consist of well-formed, non-stale, assertion-free decisions whose
effects the model predicts exactly."""
import numpy as np
import pytest

from rafting_amd import abi, workload
from tests import oracle_lib


@pytest.mark.parametrize("number", [2, 3, 5])
def test_model_agrees_with_oracle(number):
    cfg = workload.config(number, 4096)
    gen = workload.ReplayGenerator(cfg)
    orc = oracle_lib.OracleTable(cfg.groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
    orc.load_state(gen.initial_state())
    kinds = np.zeros(11, dtype=np.int64)
    for _ in range(3):
        b = gen.next_batch(40)
        out = orc.submit(b)
        assert np.all(out.status == abi.OK), np.bincount(out.status)
        kinds += workload.batch_stats(b, cfg.cluster - 1)[2]
    fin = orc.read_state()
    assert np.array_equal(fin.current_term, gen.term)
    assert np.array_equal(fin.role_epoch.astype(np.int64), gen.epoch)
    assert np.array_equal(fin.commit_index, gen.commit)
    assert np.array_equal(fin.last_index, gen.last)
    if number == 2:
        assert kinds[abi.EV_AE_ACK] > 0 and kinds[abi.EV_AE_REQ] == 0
    else:
        for k in (abi.EV_AE_REQ, abi.EV_AE_ACK, abi.EV_RV_REQ, abi.EV_PV_REQ, abi.EV_RV_REPLY, abi.EV_PV_REPLY, abi.EV_TIMEOUT):
            assert kinds[k] > 0, k


def test_conflicting_append_entries_knob():
    """p_conflict: a new leader's AppendEntries overwrites the follower's last uncommitted entries — prevLogIndex below the tail, conflict,
    truncate, append (storage/RocksLog.java:169-225 through member/Follower.java:60-75). Off by default (the BASELINE streams do not
    change); with it the model still predicts the oracle's final state, and the rows really truncate."""
    import dataclasses
    cfg = dataclasses.replace(workload.config(3, 4096), p_conflict=0.005)
    gen = workload.ReplayGenerator(cfg)
    orc = oracle_lib.OracleTable(cfg.groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
    orc.load_state(gen.initial_state())
    trunc = 0
    for _ in range(3):
        b = gen.next_batch(40)
        out = orc.submit(b)
        assert np.all(out.status == abi.OK), np.bincount(out.status)
        trunc += int(np.count_nonzero(out.reply["flags"] & abi.F_LOG_TRUNC))
    fin = orc.read_state()
    assert np.array_equal(fin.current_term, gen.term) and np.array_equal(fin.role_epoch.astype(np.int64), gen.epoch)
    assert np.array_equal(fin.commit_index, gen.commit) and np.array_equal(fin.last_index, gen.last)
    assert trunc > 500                                     # ~0.4 % of 491 520 rows
    plain = workload.ReplayGenerator(workload.config(3, 4096)).next_batch(8)
    again = workload.ReplayGenerator(dataclasses.replace(workload.config(3, 4096), p_conflict=0.0)).next_batch(8)
    assert np.array_equal(plain.ab, again.ab) and np.array_equal(plain.head, again.head)


def test_streams_are_partition_independent():
    cfg = workload.config(3, 8192)
    whole = workload.ReplayGenerator(cfg).next_batch(12)
    for first, count in ((0, 1024), (5120, 2048), (8192 - 512, 512)):
        part = workload.ReplayGenerator(cfg, first, count).next_batch(12)
        for r in range(12):
            w = slice(r * cfg.groups + first, r * cfg.groups + first + count)
            q = slice(r * count, (r + 1) * count)
            assert np.array_equal(whole.head["hdr"][w], part.head["hdr"][q])
            assert np.array_equal(whole.ab[w], part.ab[q]) and np.array_equal(whole.cd[w], part.cd[q])
            is_ae = ((part.head["hdr"][q] & 0xF) == abi.EV_AE_REQ) & ((part.head["hdr"][q] >> 12) > 0)
            # entry offsets differ (they are batch-local) but the terms they point at do not
            wa, pa = whole.head["aux"][w][is_ae], part.head["aux"][q][is_ae]
            assert np.array_equal(whole.entry_terms[wa], part.entry_terms[pa])
            reply = ~((part.head["hdr"][q] & 0xF) == abi.EV_AE_REQ)
            assert np.array_equal(whole.head["aux"][w][reply], part.head["aux"][q][reply])


def test_determinism_and_accounting():
    a = workload.ReplayGenerator(workload.config(5, 2048)).next_batch(20)
    b = workload.ReplayGenerator(workload.config(5, 2048)).next_batch(20)
    assert np.array_equal(a.head, b.head) and np.array_equal(a.entry_terms, b.entry_terms)
    dec, nbytes, kinds = workload.batch_stats(a, 4)
    assert dec == int(kinds[1:9].sum())
    # AE heartbeat 128 B, AE with n entries 144+16n, ack 128+8F, vote request 108, vote reply 48, timeout 72
    k = np.array([abi.EV_AE_REQ, abi.EV_AE_REQ, abi.EV_AE_REQ, abi.EV_AE_ACK, abi.EV_RV_REQ, abi.EV_PV_REPLY, abi.EV_TIMEOUT, abi.EV_CLIENT_APPEND])
    n = np.array([0, 1, 4, 0, 0, 0, 0, 3])
    assert workload.algorithmic_bytes(k, n, 4).tolist() == [128, 160, 208, 160, 108, 48, 72, 0]
    assert workload.algorithmic_bytes(k, n, 2).tolist()[3] == 144
