"""Static checks of the compiled gfx950 code of the step kernels (no GPU: hipcc cross-compiles; F = 4 only, ~15 s). They pin properties the
measurements of DESIGN.md section 6 rest on and that a harmless-looking source change can lose without any test failing:
  * no FLAT memory instruction in a step kernel — a pointer that lost its address space (e.g. through inline assembly) turns every access
    into flat_load / flat_store, which also count on lgkmcnt, so the LDS hand-over barrier of the two-wavefront kernels would wait for the
    whole global prefetch every round (seen in session r03k before it reached the GPU);
  * no scratch (private segment) in the kernel the bench runs on, and register budgets that keep the occupancy the launch shapes assume;
  * a round of the deciding wavefront stays within its instruction budget on the two paths a cluster in operation takes (tools/spine.py),
    and the I/O wavefront's round within its own: the two share a SIMD, the sum is what a round costs."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
K32 = "_ZN2rg13step32_kernelILi4ELb0ELi%dELb0ELi1EEEvNS_10StepParamsE"      # <F = 4, dense, WAVES, wide outcome columns, one I/O wavefront>
K32C = K32.replace("ELb0ELi1EEEv", "ELb1ELi1EEEv")      # the same with compact outcome rows (rg_submit32c)
WIDE = lambda k: k.replace("13step32_kernel", "18step32_wide_kernel").replace("ELi1EEEv", "EEEv")     # noqa: E731  (the 64-bit body as a kernel of its own: no I/O-wavefront parameter)
STEP_KERNELS = [K32 % 1, K32 % 4, K32C % 1, K32C % 4, WIDE(K32C % 1), WIDE(K32 % 1), WIDE(K32 % 4), "_ZN2rg17step_split_kernelILi4ELb0EEEvNS_10StepParamsE", "_ZN2rg11step_kernelILi4ELb0EEEvNS_10StepParamsE"]


@pytest.fixture(scope="module")
def assembly(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("isa") / "rg.s")
    subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-DRG_BUILD_ONLY_F4", "-S",
                    "--cuda-device-only", "-o", out, os.path.join(ROOT, "rafting_amd", "csrc", "rg_kernels.hip")], check=True, capture_output=True, timeout=600)
    return out


def kernel_text(path, name):
    lines = open(path).read().split("\n")
    start = next(i for i, ln in enumerate(lines) if ln.startswith(name + ":"))
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i])
    return lines[start:end]


def descriptor(lines, key):
    return int(next(re.search(r"(\d+)", ln.split(key)[1]).group(1) for ln in lines if key in ln))


@pytest.mark.parametrize("kernel", STEP_KERNELS)
def test_step_kernels_use_no_flat_memory_instructions(assembly, kernel):
    flat = [ln.strip() for ln in kernel_text(assembly, kernel) if re.match(r"\s+flat_(load|store|atomic)", ln)]
    assert not flat, "%s: %d FLAT instructions, e.g. %s" % (kernel, len(flat), flat[:3])


def test_register_and_scratch_budgets(assembly):
    for k in (K32 % 1, K32 % 4, WIDE(K32 % 1), WIDE(K32 % 4), K32C % 1, K32C % 4, WIDE(K32C % 1), WIDE(K32C % 4)):
        text = kernel_text(assembly, k)
        assert descriptor(text, ".amdhsa_private_segment_fixed_size") == 0, k       # nothing in scratch, in either body of either variant
        assert descriptor(text, ".amdhsa_next_free_vgpr") <= 128, k                  # four wavefronts per SIMD: eight workgroups per CU
        assert descriptor(text, ".amdhsa_group_segment_fixed_size") <= 20 * 1024, k  # ... whose LDS fits 160 KB


def test_a_round_stays_within_its_instruction_budget(assembly):
    def spine(kernel):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "spine.py"), assembly, kernel], capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr[-2000:]
        main, election = (int(x) for x in re.search(r"main (\d+) instructions.*election (\d+) ", p.stdout).groups())
        io = int(re.search(r"I/O wavefront: (\d+) instructions", p.stdout).group(1))
        counted, all_classes = (int(x) for x in re.search(r"vote reply counted (\d+),.*all classes (\d+)", p.stdout).groups())
        return main, election, io, counted, all_classes, p.stdout
    main, election, io, counted, all_classes, text = spine(K32 % 1)
    # round 3's tier 1 (bool predicates: v_cmp + s_and + v_cndmask): 322 / 534 and 148; the sign-word tier of rg_tier1n.hpp with the I/O wavefront's
    # tables: 186 / 336 and 139; round 5, one block per election row class: a round with a vote reply that is merely counted — three election
    # rounds of four — 268, every class at once 344. Every instruction is four cycles of every round of every SIMD.
    assert 0 < main <= 195 and 0 < counted <= 285 and election == all_classes and 0 < all_classes <= 360, text
    assert 0 < io <= 150, text
    # compact outcome rows (rg_submit32c): one unconditional 16-byte store instead of a reply and a conditional effect row
    main_c, _, io_c, counted_c, _, text_c = spine(K32C % 1)
    assert main_c == main and counted_c == counted and 0 < io_c <= 130, text_c


def test_sign_word_primitives_and_the_io_wavefronts_tables(tmp_path):
    """tests/native/tier1n_tables.cpp, compiled for the host through the header shim of tests/devemu: s_lt / s_ne / s_pos on the borders of their
    domain (and the NO_NODE exception tier 1 relies on), the predicate word's two tables against expand_predicates() for every 14-bit word,
    the class word (table entry + per-row corrections) against the plain definition of every class bit for 7.3 M (cluster, self, kind, slot,
    flag, same-term, n, aux, field) combinations in and out of the 32-bit tier's domain."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "tier1n_tables")
    subprocess.run(["g++", "-O1", "-std=c++17", "-w", "-I" + os.path.join(ROOT, "tests", "devemu"), "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "native", "tier1n_tables.cpp"), os.path.join(ROOT, "tests", "devemu", "emu_runtime.cpp"), "-pthread", "-o", exe],
                   check=True, timeout=600)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "tier1n tables ok" in p.stdout, p.stdout[-3000:]
