#!/usr/bin/env python
"""Extracts the literal constants and the one in-source golden table of the reference's decision path from its SOURCE
TEXT (nothing is executed; there is no JVM here) into tests/golden/reference_pins.json.

    python tools/make_reference_pins.py [/root/reference]

tests/test_reference_pins.py checks (a) that the committed JSON still equals a fresh extraction whenever the reference
checkout is present, and (b) that the oracle behaves according to every pinned value."""
import json
import os
import re
import sys
import xml.etree.ElementTree as ET

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "src/main/java/io/lubricant/consensus/raft"


def _read(ref, rel):
    with open(os.path.join(ref, rel), encoding="utf-8") as f:
        return f.read()


def _one(pattern, text, what):
    m = re.search(pattern, text)
    if not m:
        raise SystemExit("reference changed: cannot find %s" % what)
    return m


def extract(ref):
    pins = {}
    lead = _read(ref, SRC + "/context/member/Leadership.java")
    pins["REPLICATE_LIMIT"] = int(_one(r"int\s+REPLICATE_LIMIT\s*=\s*(\d+)", lead, "REPLICATE_LIMIT").group(1))
    pins["IN_FLIGHT_LIMIT"] = int(_one(r"int\s+IN_FLIGHT_LIMIT\s*=\s*(\d+)", lead, "IN_FLIGHT_LIMIT").group(1))
    # the comment table above `int majorIndex = matchIndices.length / 2`: "// N = 5, major = 3 : |x|x|o|x|*|"
    table = {}
    for n, major, cells in re.findall(r"//\s*N\s*=\s*(\d+),\s*major\s*=\s*(\d+)\s*:\s*\|([xo*|]+)\|", lead):
        slots = cells.split("|")
        assert slots[-1] == "*" and slots.count("o") == 1, cells
        table[n] = {"major": int(major), "followers": len(slots) - 1, "quorum_slot_of_sorted_followers": slots.index("o")}
    if sorted(table) != ["2", "3", "4", "5", "6", "7"]:
        raise SystemExit("reference changed: quorum table rows %s" % sorted(table))
    pins["quorum_table"] = table
    pins["major_index_expr"] = _one(r"int\s+majorIndex\s*=\s*([^;]+);", lead, "majorIndex").group(1).strip()
    pins["rejection_step_expr"] = _one(r"long\s+step\s*=\s*([^;]+?)\s*;", lead, "rejection step").group(1).strip()

    leader = _read(ref, SRC + "/context/member/Leader.java")
    pins["heartbeat_in_flight_divisor"] = int(_one(r"IN_FLIGHT_LIMIT\s*/\s*\(heartbeat\s*\?\s*(\d+)\s*:\s*1\)", leader,
                                                   "in-flight divisor").group(1))
    pins["heartbeat_fetch_shift"] = int(_one(r"REPLICATE_LIMIT\s*>>\s*\(heartbeat\s*\?\s*(\d+)\s*:\s*0\)", leader,
                                             "fetch shift").group(1))
    pins["in_flight_gate_expr"] = _one(r"if\s*\((state\.requestInFlight\s*>\s*requestLimit)\)", leader, "in-flight gate").group(1)

    ctx = _read(ref, SRC + "/context/RaftContext.java")
    pins["majority_expr"] = _one(r"int\s+majority\(\)\s*\{\s*return\s+([^;]+);", ctx, "majority()").group(1).strip()

    conf = _read(ref, SRC + "/support/RaftConfig.java")
    pins["election_timeout_expr"] = _one(r"nextInt\(([^)]*)\)", conf, "electionTimeout range").group(1).strip()

    xml = ET.parse(os.path.join(ref, "src/test/resources/raft1.xml")).getroot()
    pins["raft1_xml"] = {
        "pre_vote": xml.findtext("timeout/pre-vote").strip() == "true",
        "tick_ms": float(xml.findtext("timeout/tick")),
        "heartbeat_ticks": float(xml.findtext("timeout/heartbeat")),
        "election_ticks": float(xml.findtext("timeout/election")),
        "broadcast_ticks": float(xml.findtext("timeout/broadcast")),
        "avail_critical_point": int(xml.findtext("metrics/avail-critical-point")),
        "recovery_cool_down_ms": int(xml.findtext("metrics/recovery-cool-down")),
        "cluster_size": 1 + len(xml.findall("cluster/remote")),
    }
    return pins


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    pins = extract(ref)
    out = os.path.join(ROOT, "tests", "golden", "reference_pins.json")
    with open(out, "w") as f:
        json.dump(pins, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
