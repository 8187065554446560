#!/usr/bin/env python
"""Writes tests/golden/kryo_bodies.json: RPC bodies of the reference in Kryo 4.0.2's format, built by tests/kryo_ref.py (an independent
Python restatement of the format rules). UNVERIFIED AGAINST A JVM — Kryo cannot run in this image; INTEGRATION.md holds the JUnit test
that prints the same hex on a machine with a JDK."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import kryo_ref  # noqa: E402

NODES = [("127.0.0.1", 6001), ("127.0.0.1", 6002), ("127.0.0.1", 6003), ("raft-node-with-a-long-name.cluster.example.org", 65000), ("h", 1)]
CASES = [
    dict(name="heartbeat", method=1, term=7, node=0, x=41, y=7, leader_commit=40, entry_terms=[]),
    dict(name="two entries of one term", method=1, term=7, node=1, x=41, y=7, leader_commit=41, entry_terms=[7, 7]),
    dict(name="entries of two terms", method=1, term=9, node=2, x=1000000, y=8, leader_commit=999998, entry_terms=[8, 9, 9]),
    dict(name="first AppendEntries of an empty log", method=1, term=1, node=0, x=0, y=0, leader_commit=0, entry_terms=[1]),
    dict(name="large values", method=1, term=(1 << 40) + 5, node=0, x=(1 << 62) - 3, y=1 << 40, leader_commit=(1 << 62) - 3, entry_terms=[(1 << 40) + 5]),
    dict(name="negative values travel too", method=1, term=-1, node=1, x=-2, y=-3, leader_commit=-(1 << 63), entry_terms=[]),
    dict(name="long host name", method=3, term=12, node=3, x=77, y=11),
    dict(name="one-character host name", method=2, term=13, node=4, x=78, y=12),
    dict(name="requestVote", method=3, term=8, node=1, x=41, y=7),
    dict(name="preVote", method=2, term=8, node=2, x=41, y=7),
    dict(name="installSnapshot", method=4, term=8, node=0, x=500, y=6),
]
RESPONSES = [(7, True), (7, False), (0, False), ((1 << 63) - 1, True), (-5, False), (300, True)]

out = {"_status": "UNVERIFIED AGAINST A JVM: built by tests/kryo_ref.py from Kryo 4.0.2's published format rules, not by Kryo itself "
                  "(INTEGRATION.md: 'Checking the Kryo format' shows the JUnit test that prints these bodies on a machine with a JDK)",
       "nodes": ["%s:%d" % n for n in NODES], "requests": [], "responses": []}
for c in CASES:
    body = kryo_ref.request(NODES, c["method"] == 1, c["term"], c["node"], c["x"], c["y"], c.get("leader_commit", 0), c.get("entry_terms", ()))
    out["requests"].append(dict(c, hex=body.hex()))
for term, ok in RESPONSES:
    out["responses"].append(dict(term=term, success=ok, hex=kryo_ref.response(term, ok).hex()))
with open(os.path.join(ROOT, "tests", "golden", "kryo_bodies.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote %d requests, %d responses" % (len(out["requests"]), len(out["responses"])))
