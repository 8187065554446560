"""The product's device decision code (rafting_amd/csrc/rg_device.hpp, rg_kernels.hip) and C-ABI host code
(raftgpu.cpp), compiled UNCHANGED for the host against a stand-in for the HIP headers (tests/devemu/hip/hip_runtime.h:
heap for device memory, a lane-serial grid for a launch), run against the oracle — so that `-m "not gpu"` already
catches a logic error in the decision code. This is test infrastructure: nothing in rafting_amd/ can load that library,
and libraftgpu.so has no CPU path."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "devemu")
LIB = os.path.join(EMU, "libraftgpu_emu.so")
SOURCES = [os.path.join(ROOT, "rafting_amd", "csrc", f) for f in ("rg_kernels.hip", "raftgpu.cpp", "rg_device.hpp", "rg_step.hpp")] + [
    os.path.join(EMU, "emu_runtime.cpp"), os.path.join(EMU, "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "raftgpu.h")]


@pytest.fixture(scope="module")
def emulation_library():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in SOURCES):
        subprocess.run(["g++", "-O1", "-g0", "-std=c++17", "-fPIC", "-shared", "-w", "-pthread", "-I" + EMU, "-I" + os.path.join(ROOT, "include"),
                        "-x", "c++", SOURCES[0], SOURCES[1], os.path.join(EMU, "emu_runtime.cpp"), "-o", LIB], check=True, cwd=EMU)
    return LIB


def test_device_decision_code_on_the_host_matches_the_oracle(emulation_library):
    env = dict(os.environ, RG_LIB=emulation_library, RG_SPLIT="0", RG_ALLOW_HOST_EMULATION="1", PYTHONPATH=ROOT)
    env.pop("RG_FAST", None)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(EMU, "emu_cases.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-6000:] + p.stderr[-3000:]
    assert " passed" in p.stdout and "failed" not in p.stdout


def test_bench_contract_end_to_end_on_the_emulation(emulation_library):
    """bench.py itself (tests/devemu/bench_dry.py stubs only torch's GPU probes): one JSON line with the contract's keys, roofline and
    cpu_baseline objects, the three host-memory legs, and the check of the first rounds against the translated reference."""
    import json
    env = dict(os.environ, RG_LIB=emulation_library, RG_SPLIT="1", RG_EMU_WAVES="1", RG_ALLOW_HOST_EMULATION="1", PYTHONPATH=ROOT)   # wavefront mode: bench.py's
    env.pop("RG_FAST", None)                                                    # default path is the two-wavefront compact-format kernel
    p = subprocess.run([sys.executable, os.path.join(EMU, "bench_dry.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "pcie_inclusive_value"):
        assert k in d, k
    assert d["steps"] == 2 and d["warmup"] == 1 and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"] == "int64"
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "valu", "per_launch", "pmc_passes")) <= set(d["roofline"]) and d["roofline"]["bound"] == "valu"
    pl = d["roofline"]["per_launch"]                                          # round 6: one event pair per launch, the same launches once more
    assert "error" not in pl and pl["launches"] == 2 and 0 < pl["min_ms"] <= pl["median_ms"] <= pl["max_ms"]
    assert d["roofline"]["pmc_passes"] is None and d["roofline"]["valu"] is None      # (no counters, no device properties on the emulation)
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] == "port"
    legs = d["pcie_inclusive"]
    assert legs["serial_rg_submit"] > 0 and legs["pipelined_rg_submit_async"] > 0 and legs["pipelined_rg_submit_async_packed"] > 0
    assert legs["packed_bytes_per_decision"] < legs["wide_bytes_per_decision"]
    rt = d["cpu_baseline"].get("reference_translated")
    assert rt is None or "error" not in rt, rt
    r = d["roofline"]                                                           # round 5: `frac` is moved bytes over time over the peak (the PMC quotation, or — here, no
    assert r["layout_estimate_bytes"] > 0 and r["layout_estimate_bytes"] < r["algorithmic_bytes_per_launch"]     # entry for an emulation build — the layout estimate, named as such)
    assert r["traffic"] is None and r["frac_basis"].startswith("layout estimate") and abs(r["frac"] - r["frac_of_layout_estimate"]) < 1e-12
    assert r["traffic_measured_in_this_run"] is False and r["work_rate_algorithmic_gbps"] > 0 and r["outcome_format"] == "rg_outcome32_t"
    assert r["ms_int64_body"] > 0 and r["value_int64_body"] > 0 and r["int64_body_counters_equal"] is True
    assert "golden" in d and "model_overcharges" in r
    tk = d["tick_latency"]                                                      # round 5: the once-per-tick path, both ways
    assert "error" not in tk, tk
    assert tk["rg_tick_launch"]["p50_us"] > 0 and tk["rg_submit_async_packed"]["p99_us"] > 0 and tk["device_us_per_single_round_launch"] > 0
    assert tk["rg_tick_launch"]["max_us"] >= tk["rg_tick_launch"]["p99_us"] and tk["device_us_per_resident_tick"] > 0      # round 6: the device-resident tick (rg_tick2_*)
    ll = d["long_lived_groups"]                                                 # round 5: groups at 2^40 on the 32-bit body (index bases), checked against the oracle in the run
    assert "error" not in ll, ll
    assert d["value_long_lived_groups"] > 0 and ll["int64_body_workgroups"] == 0 and d["int64_body_workgroups"] == 0 and "bit-identical" in ll["checked"]
    lg = d["long_launches"]                                                     # round 6: the same stream in longer launches, reported beside `value`
    assert d["value_long_launches"] > 0 and lg["rounds_per_launch"] == 8 and lg["launches"] == 2 and lg["int64_body_workgroups"] == 0
    adv = d["adverse_mix"]                                                      # round 5: the adverse mix as a line of the default run
    assert "error" not in adv, adv
    assert d["value_adverse_mix"] > 0 and adv["need_host_as_expected"] is True and adv["counters"]["need_host"] > 0 and adv["counters"]["asserts"] == 0
    assert 0.02 < adv["mix"]["election_rows_share"] < 0.2 and 0.0 < adv["mix"]["rows_parked_share"] < 0.5


def test_bench_with_two_ranks_is_config4_sharded_over_gloo(emulation_library):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per GPU), on the emulation: config 4's stream, rank r
    decides block r, nothing but a barrier and three scalars crosses ranks (gloo), rank 0 prints ONE line with the whole-job sum."""
    import json
    import socket
    with socket.socket() as sk:                            # a port nobody is listening on right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, RG_LIB=emulation_library, RG_SPLIT="1", RG_EMU_WAVES="1", RG_ALLOW_HOST_EMULATION="1", PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    env.pop("RG_FAST", None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(EMU, "bench_dry.py"), "--gpus", "2", "--device", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["config_number"] == 4 and d["config"]["seed"] == "0xc0ffee03"
    assert d["config"]["groups_per_gpu"] == 256 and d["config"]["groups_total"] == 512
    assert d["config"]["decisions_per_step_per_gpu"] > 0 and d["value"] > 0
    assert d["cpu_baseline"] is None and d["pcie_inclusive_value"] is None          # rank-0-at-N=1 legs only
    # round 5: EVERY rank checked its first launch against the digest the reference's own code produced for its block of the table
    assert d["golden"] == [{"rank": 0, "case": "config4_shard0_emulation_launch", "outcomes": "ok"},
                           {"rank": 1, "case": "config4_shard1_emulation_launch", "outcomes": "ok"}], d["golden"]
    assert len(lines[0]) > 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("[Gloo]")]      # gloo's connection chatter stays off stdout: ONE line there


def test_a_rank_whose_first_launch_misses_the_reference_digest_fails_the_run(emulation_library, tmp_path):
    """the same two-rank run against a golden file whose case for rank 1 was tampered with: no result line, exit code != 0, the verdicts on stderr"""
    import json
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "replay_digests.json")))
    doc["cases"]["config4_shard1_emulation_launch"]["outcomes"] = "0" * 64
    bad = tmp_path / "tampered.json"
    bad.write_text(json.dumps(doc))
    env = dict(os.environ, RG_LIB=emulation_library, RG_SPLIT="1", RG_EMU_WAVES="1", RG_ALLOW_HOST_EMULATION="1", PYTHONPATH=ROOT, OMP_NUM_THREADS="1",
               RG_GOLDEN_FILE=str(bad), RG_BENCH_ENTRY=os.path.join(EMU, "bench_dry.py"))
    env.pop("RG_FAST", None)
    p = subprocess.run([sys.executable, os.path.join(EMU, "bench_dry.py"), "--gpus", "2", "--device", "0"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")], p.stdout[-1000:]
    assert "does not reproduce the reference's digest" in p.stderr


def test_bench_starts_its_own_ranks_when_no_launcher_does(emulation_library):
    """`python bench.py --gpus 2` with NO torch.distributed.run around it (the first hardware SCALE run must not die on the launcher):
    bench.py spawns one process per GPU itself, the ranks rendezvous on 127.0.0.1, rank 0 prints the one line — with every rank's own figures
    in `per_gpu` and the whole-job sum in `value`."""
    import json
    env = dict(os.environ, RG_LIB=emulation_library, RG_SPLIT="1", RG_EMU_WAVES="1", RG_ALLOW_HOST_EMULATION="1", PYTHONPATH=ROOT, OMP_NUM_THREADS="1",
               RG_BENCH_ENTRY=os.path.join(EMU, "bench_dry.py"))
    for k in ("RG_FAST", "WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(EMU, "bench_dry.py"), "--gpus", "2", "--device", "0"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["config_number"] == 4 and d["config"]["groups_total"] == 512 and "itself" in d["launcher"]
    assert [g["rank"] for g in d["per_gpu"]] == [0, 1] and all(g["value"] > 0 for g in d["per_gpu"])
    # the whole-job value is the sum of the ranks' decisions over the slowest rank's time: between the slower rank's doubled and the sum of both
    assert 2 * min(g["value"] for g in d["per_gpu"]) * 0.7 <= d["value"] <= sum(g["value"] for g in d["per_gpu"]) * 1.001
    # a rank that cannot start takes the run down with a non-zero exit instead of a hang
    bad = subprocess.run([sys.executable, os.path.join(EMU, "bench_dry.py"), "--gpus", "2", "--device", "0", "--config", "9"], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=300)
    assert bad.returncode != 0


def test_kernels_that_need_lanes_to_meet_on_emulated_wavefronts(emulation_library):
    """RG_EMU_WAVES=1: every lane of a workgroup is an OS thread, shuffles / ballots meet per 64-lane wavefront, barriers per
    workgroup — the two-wavefront step kernel with its LDS rings, the decision counters, the ballot-compacted timer list."""
    env = dict(os.environ, RG_LIB=emulation_library, RG_SPLIT="1", RG_EMU_WAVES="1", RG_ALLOW_HOST_EMULATION="1", PYTHONPATH=ROOT)
    env.pop("RG_FAST", None)
    # (185 cases, each a grid of OS threads: four pytest-xdist workers where the plugin is there — the cases share nothing)
    par = ["-n", "4"] if __import__("importlib.util").util.find_spec("xdist") else []
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(EMU, "emu_cases_waves.py"), "-x", "-q", "-p", "no:cacheprovider"] + par,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-6000:] + p.stderr[-3000:]
    assert " passed" in p.stdout and "failed" not in p.stdout


def test_cpp_host_mirror_runs_a_whole_exchange_on_the_emulation(emulation_library):
    """rafting_amd/host (RaftContext, ContextManager, StableStore) above the C-ABI: three nodes from start-up timeouts
    through PreVote, election, client commands, back-off replication, commit, fencing, the readiness gate, step-down and
    the durability journal (tests/devemu/host_flow.cpp); every decision by the device code on the host emulation."""
    exe = os.path.join(ROOT, "build", "devemu_host_flow")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    host = os.path.join(ROOT, "rafting_amd", "host")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-I" + host, "-I" + os.path.join(ROOT, "include"),
                    os.path.join(EMU, "host_flow.cpp"), os.path.join(host, "raft_host.cpp"), os.path.join(host, "stable_store.cpp"),
                    "-L" + EMU, "-l:libraftgpu_emu.so", "-Wl,-rpath," + EMU, "-pthread", "-o", exe], check=True)
    p = subprocess.run([exe], env=dict(os.environ, RG_SPLIT="0"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "host flow ok" in p.stdout, p.stdout + p.stderr


def test_ingress_pipeline_from_socket_bytes_to_response_bytes_on_the_emulation(emulation_library, tmp_path):
    """rafting_amd/host/ingress_pipeline.cpp — reader threads feeding the ingress, the flush thread sealing into rg_submit_async_packed
    (the sealed bank is the upload buffer), the durability journal, emitter threads — with the compact-row step kernel on the wavefront
    emulation: every follower's log ends where its requests said, every request got one successful response frame, no row needed the host."""
    exe = os.path.join(ROOT, "build", "devemu_ingress_pipeline")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    host = os.path.join(ROOT, "rafting_amd", "host")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-I" + host, "-I" + os.path.join(ROOT, "include")] +
                   [os.path.join(host, f) for f in ("ingress_pipeline.cpp", "ingress.cpp", "wire.cpp", "kryo_body.cpp", "stable_store.cpp")] +
                   ["-L" + EMU, "-l:libraftgpu_emu.so", "-Wl,-rpath," + EMU, "-pthread", "-o", exe], check=True)
    for args in (["256", "12", "4", "2", "2", "5"], ["200", "9", "3", "3", "1", "16"]):
        p = subprocess.run([exe] + args + [str(tmp_path / "journal")], env=dict(os.environ, RG_SPLIT="1", RG_EMU_WAVES="1"), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "ingress pipeline ok=1" in p.stdout, p.stdout + p.stderr


def test_ingress_flusher_repairs_from_real_logs_and_applies_effects_on_the_emulation(emulation_library, tmp_path):
    """rafting_amd/host/ingress_flusher.cpp — seal, decide, apply log effects, repair RG_NEED_HOST from MemoryLogs (six term runs each, the device
    caches four: every group misses in its second row and the two rows behind it are skipped), a row beyond int32 beside the batch, the durability
    journal before the replies, response frames (tests/devemu/ingress_flusher_flow.cpp): every MemoryLog, the table, the journal and the responses
    agree with what the requests said."""
    exe = os.path.join(ROOT, "build", "devemu_ingress_flusher_flow")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    host = os.path.join(ROOT, "rafting_amd", "host")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-I" + host, "-I" + os.path.join(ROOT, "include"), os.path.join(EMU, "ingress_flusher_flow.cpp")] +
                   [os.path.join(host, f) for f in ("ingress_flusher.cpp", "ingress.cpp", "wire.cpp", "kryo_body.cpp", "raft_host.cpp", "stable_store.cpp")] +
                   ["-L" + EMU, "-l:libraftgpu_emu.so", "-Wl,-rpath," + EMU, "-pthread", "-o", exe], check=True)
    for args in (["70"], ["70", "2"], ["130", "3"], ["12", "3"]):         # one table; a sharded ingress in front of 2 / 3 tables (the row beside the batch in shard 0 / 1)
        p = subprocess.run([exe, args[0], str(tmp_path / "journal")] + args[1:], env=dict(os.environ, RG_SPLIT="0"), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "ingress flusher ok=1" in p.stdout, p.stdout + p.stderr


def test_three_nodes_that_exchange_nothing_but_wire_bytes_on_the_emulation(emulation_library, tmp_path):
    """BASELINE configs[0] on the wire path only (tests/devemu/ingress_cluster_flow.cpp): three nodes — table + Ingress + IngressFlusher + MemoryLogs
    each — start from nothing, time out, PreVote, RequestVote, elect a leader per group, take client commands, replicate (rg_replicate ->
    encode_sends), commit; a leader is cut off and comes back (re-election, step-down). Every decision by the device code on the emulation, every
    message a frame of the reference's protocol. Election safety and the stability / agreement of committed entries are checked at every tick."""
    exe = os.path.join(ROOT, "build", "devemu_ingress_cluster_flow")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    host = os.path.join(ROOT, "rafting_amd", "host")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-I" + host, "-I" + os.path.join(ROOT, "include"), os.path.join(EMU, "ingress_cluster_flow.cpp")] +
                   [os.path.join(host, f) for f in ("ingress_flusher.cpp", "ingress.cpp", "wire.cpp", "kryo_body.cpp", "raft_host.cpp", "stable_store.cpp")] +
                   ["-L" + EMU, "-l:libraftgpu_emu.so", "-Wl,-rpath," + EMU, "-pthread", "-o", exe], check=True)
    for args in (["6", "500"], ["24", "260"]):
        p = subprocess.run([exe] + args + ["wide", str(tmp_path / "cluster")], env=dict(os.environ, RG_SPLIT="0"), capture_output=True, text=True, timeout=900)
        assert p.returncode == 0 and "ingress cluster ok=1" in p.stdout, p.stdout + p.stderr[-3000:]


def test_the_product_binding_refuses_the_emulation_library(emulation_library):
    """rafting_amd.engine must not be talked into a CPU path by pointing RG_LIB at the test artefact"""
    env = dict(os.environ, RG_LIB=emulation_library, PYTHONPATH=ROOT)
    env.pop("RG_ALLOW_HOST_EMULATION", None)
    p = subprocess.run([sys.executable, "-c", "from rafting_amd import engine; engine.Table(4, 3)"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "test-only host emulation" in p.stderr


def test_multi_device_manager_matches_a_single_table_on_the_emulation(emulation_library):
    """rafting_amd/host/multi_device.cpp: three tables behind one routing manager, each drained by its own feeder thread, against one
    table holding the same contexts under the same random traffic (rafting_amd/host/multi_device_unit.cpp) — on the host emulation"""
    exe = os.path.join(ROOT, "build", "devemu_multi_device_unit")
    host = os.path.join(ROOT, "rafting_amd", "host")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-I" + host, "-I" + os.path.join(ROOT, "include"),
                    os.path.join(host, "multi_device_unit.cpp"), os.path.join(host, "multi_device.cpp"), os.path.join(host, "raft_host.cpp"),
                    os.path.join(host, "stable_store.cpp"), "-L" + EMU, "-l:libraftgpu_emu.so", "-Wl,-rpath," + EMU, "-pthread", "-o", exe], check=True)
    p = subprocess.run([exe, "48", "3", "120"], env=dict(os.environ, RG_SPLIT="0"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "multi-device ok=1" in p.stdout, p.stdout + p.stderr


def test_multi_device_manager_is_race_free_under_thread_sanitizer(emulation_library):
    """createContext / getContext from one thread while another runs flushAll (the last phase of multi_device_unit.cpp) —
    the same program built with -fsanitize=thread must finish with `ok=1` and without a single data-race report on the routing table."""
    exe = os.path.join(ROOT, "build", "devemu_multi_device_unit_tsan")
    host = os.path.join(ROOT, "rafting_amd", "host")
    r = subprocess.run(["g++", "-O1", "-g", "-fsanitize=thread", "-std=c++17", "-I" + host, "-I" + os.path.join(ROOT, "include"),
                        os.path.join(host, "multi_device_unit.cpp"), os.path.join(host, "multi_device.cpp"), os.path.join(host, "raft_host.cpp"),
                        os.path.join(host, "stable_store.cpp"), "-L" + EMU, "-l:libraftgpu_emu.so", "-Wl,-rpath," + EMU, "-pthread", "-o", exe],
                       capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime for g++ here: " + r.stderr[-200:])
    p = subprocess.run([exe, "48", "3", "40"], env=dict(os.environ, RG_SPLIT="0"), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "multi-device ok=1" in p.stdout, p.stdout + p.stderr[-3000:]
    assert "ThreadSanitizer" not in p.stderr, p.stderr[-3000:]
