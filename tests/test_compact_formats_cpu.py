"""The host-side halves of the compact formats (ABI 4), which need no device: rg_batch32_pack_rel (wide rows -> rg_batch32_t rows relative to per-group
index bases) and rg_outcome32_unpack(_rel) (rg_out32_t / rg_persist32_t rows -> the rg_outcome_t image) of the PRODUCT library, against plain numpy
statements of what include/raftgpu.h says — random batches, the zero convention, every refusal."""
import numpy as np
import pytest

from rafting_amd import abi, engine

INDEX_FIELDS = {abi.EV_AE_REQ: 0xA, abi.EV_AE_ACK: 0x6, abi.EV_IS_ACK: 0x2, abi.EV_RV_REQ: 0x2, abi.EV_PV_REQ: 0x2, abi.EV_LOG_FLUSH: 0x1, abi.EV_IS_REQ: 0x2}


def random_wide_batch(rng, rounds, count, base, sparse=False):
    """events whose index fields lie 0 or within (base, base + 2^31) of their group, terms small"""
    gid = np.sort(rng.choice(4 * count, size=count, replace=False)).astype(np.uint32) if sparse else None
    b = abi.Batch(1 if sparse else rounds, count, gid=gid)
    groups = gid if sparse else np.arange(count)
    for r in range(b.rounds):
        for i in range(count):
            kind = int(rng.integers(0, 12))
            ix = INDEX_FIELDS.get(kind, 0)
            f = []
            for k in range(4):
                if (ix >> k) & 1:
                    f.append(0 if rng.random() < 0.2 else int(base[groups[i]]) + int(rng.integers(1, 1 << 30)))
                else:
                    f.append(int(rng.integers(0, 1 << 20)))
            n = int(rng.integers(0, 4)) if kind == abi.EV_AE_REQ else (int(rng.integers(1, 5)) if kind == abi.EV_CLIENT_APPEND else 0)
            ents = None
            if kind == abi.EV_AE_REQ and n:
                t = int(rng.integers(1, 9))
                ents = [t] * n if rng.random() < 0.7 else [t + int(x) for x in rng.integers(0, 2, size=n)]
            b.put(r, i, kind, slot=int(rng.integers(0, 5)), flag=int(rng.integers(0, 2)), a=f[0], b=f[1], c=f[2], d=f[3], aux=int(rng.integers(0, 100)), entries=ents, n=n)
    return b, groups


def test_pack_rel_is_the_plain_packer_with_indices_taken_off_their_bases():
    rng = np.random.default_rng(5)
    for sparse in (False, True):
        groups_total = 4 * 96 if sparse else 96
        base = np.where(rng.random(groups_total) < 0.3, 0, rng.integers(1, 1 << 45, size=groups_total)).astype(np.int64)
        b, groups = random_wide_batch(rng, 3, 96, base, sparse)
        got = engine.pack32(b, index_base=base)
        # the same batch with the subtraction done here, through the plain packer
        kind = b.head["hdr"] & 0xF
        shifted = abi.Batch(b.rounds, b.count, gid=b.gid)
        shifted.head[:], shifted.entry_terms, shifted.entry_count = b.head, b.entry_terms, b.entry_count
        row_base = base[np.tile(groups, b.rounds)]
        for k, (src, dst) in enumerate(((b.ab["x"], shifted.ab["x"]), (b.ab["y"], shifted.ab["y"]), (b.cd["x"], shifted.cd["x"]), (b.cd["y"], shifted.cd["y"]))):
            is_ix = np.array([(INDEX_FIELDS.get(int(kd), 0) >> k) & 1 for kd in kind], dtype=bool)
            dst[:] = np.where(is_ix & (src != 0), src - row_base, src)
        ref = engine.pack32(shifted)
        assert np.array_equal(got.head, ref.head) and np.array_equal(got.abcd, ref.abcd) and got.entry_count == ref.entry_count
        assert np.array_equal(got.entry_terms[:got.entry_count], ref.entry_terms[:ref.entry_count])
        none = kind == abi.EV_NONE
        assert not got.abcd["a"][none].any() and not got.abcd["d"][none].any()          # a row that is not addressed carries nothing
        # zero stays zero, everything else is >= 1
        for k, nm in enumerate("abcd"):
            is_ix = np.array([(INDEX_FIELDS.get(int(kd), 0) >> k) & 1 for kd in kind], dtype=bool)
            col = (b.ab["x"], b.ab["y"], b.cd["x"], b.cd["y"])[k]
            assert np.array_equal(got.abcd[nm][is_ix] == 0, col[is_ix] == 0)


def test_pack_rel_refuses_what_has_no_image():
    base = np.full(8, 1000, dtype=np.int64)
    for kind, field, value in ((abi.EV_AE_REQ, "b", 1000), (abi.EV_AE_REQ, "d", 7), (abi.EV_AE_ACK, "c", 1000 + (1 << 31)), (abi.EV_RV_REQ, "b", 999)):
        b = abi.Batch(1, 8)
        b.put(0, 3, kind, slot=1, **{"a": 5, "b": 0, "c": 0, "d": 0, field: value})
        with pytest.raises(engine.EngineError):
            engine.pack32(b, index_base=base)                   # at or below the base, or 2^31 above it
    ok = abi.Batch(1, 8)
    ok.put(0, 3, abi.EV_AE_REQ, slot=1, a=5, b=1001, c=5, d=0)        # leaderCommit 0 is "none": it travels as 0 whatever the base
    p = engine.pack32(ok, index_base=base)
    assert (int(p.abcd["b"][3]), int(p.abcd["d"][3])) == (1, 0)
    ok.put(0, 4, abi.EV_AE_REQ, slot=1, a=1 << 31, b=1001, c=5, d=0)   # a TERM beyond int32 is refused with or without bases
    with pytest.raises(engine.EngineError):
        engine.pack32(ok, index_base=base)


def test_unpack_restores_the_wide_image():
    rng = np.random.default_rng(9)
    rounds, count = 5, 64
    base = np.where(rng.random(count) < 0.5, 0, rng.integers(1, 1 << 44, size=count)).astype(np.int64)
    raw = abi.Outcome32(rounds * count, wide=True)
    flags = rng.integers(0, 1 << 8, size=rounds * count).astype(np.uint32) | (rng.integers(0, 3, size=rounds * count).astype(np.uint32) << abi.F_ROLE_SHIFT)
    flags[rng.random(rounds * count) < 0.05] |= np.uint32(abi.NEED_HOST << abi.F_STATUS_SHIFT)
    raw.row["flags"] = flags
    raw.row["resp_term"] = rng.integers(0, 1 << 20, size=rounds * count)
    raw.row["commit_index"] = np.where(rng.random(rounds * count) < 0.2, 0, rng.integers(1, 1 << 30, size=rounds * count))
    raw.row["log_from"] = rng.integers(0, 1 << 30, size=rounds * count)
    raw.persist["term"] = rng.integers(0, 1 << 20, size=rounds * count)
    raw.persist["voted_for"] = rng.integers(-1, 5, size=rounds * count)
    raw.persist["role_epoch"] = rng.integers(1, 1 << 20, size=rounds * count)
    raw.persist["role"] = (flags >> abi.F_ROLE_SHIFT) & 3
    # three rows that the 64-bit body flagged: their truth is in the overflow columns
    wide_rows = [7, 130, 299]
    for w in wide_rows:
        raw.row["flags"][w] |= abi.F_WIDE_VALUES
        raw.wide.reply[w] = (1 << 40, int(flags[w]), 77 + w)
        raw.wide.logfx[w] = ((1 << 41) + w, (1 << 42) + w)
        raw.wide.persist[w] = ((1 << 43), 2, 1)
    ep0 = rng.integers(1, 100, size=count).astype(np.uint32)
    got, ep_after = engine.unpack32(raw, rounds, count, ep0, index_base=base)
    # the plain statement
    ep = ep0.copy()
    fl2 = flags.reshape(rounds, count)
    for r in range(rounds):
        for i in range(count):
            row = r * count + i
            f = int(fl2[r, i])
            has_lfx = (f & (abi.F_COMMIT | abi.F_LOG_APPEND | abi.F_LOG_TRUNC)) != 0 or ((f >> abi.F_STATUS_SHIFT) & 0xFF) == abi.NEED_HOST
            has_from = (f & (abi.F_LOG_APPEND | abi.F_LOG_TRUNC)) != 0 or ((f >> abi.F_STATUS_SHIFT) & 0xFF) == abi.NEED_HOST
            has_per = (f & abi.F_PERSIST) != 0
            if row in wide_rows:
                assert tuple(got.reply[row]) == tuple(raw.wide.reply[row])
                assert tuple(got.logfx[row]) == (tuple(raw.wide.logfx[row]) if has_lfx else (0, 0))
                assert tuple(got.persist[row]) == (tuple(raw.wide.persist[row]) if has_per else (0, 0, 0))
                ep[i] = raw.wide.reply["role_epoch"][row]
                continue
            if has_per:
                ep[i] = raw.persist["role_epoch"][row]
            on = lambda v: 0 if v == 0 else int(v) + int(base[i])        # noqa: E731
            assert tuple(got.reply[row]) == (int(raw.row["resp_term"][row]) if f & abi.F_REPLIED else 0, f, int(ep[i]))
            assert tuple(got.logfx[row]) == ((on(raw.row["commit_index"][row]), on(raw.row["log_from"][row]) if has_from else 0) if has_lfx else (0, 0))
            assert tuple(got.persist[row]) == ((int(raw.persist["term"][row]), int(raw.persist["voted_for"][row]), int(raw.persist["role"][row])) if has_per else (0, 0, 0))
    assert np.array_equal(ep, ep_after)
    # a flagged row without overflow columns cannot be restored
    raw.wide = None
    with pytest.raises(engine.EngineError):
        engine.unpack32(raw, rounds, count, ep0, index_base=base)
