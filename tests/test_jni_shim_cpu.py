"""The reference-side binding as files (integration/). No JDK in this image, so:
  * integration/jni/raftgpu_jni.c is TYPE-CHECKED against include/raftgpu.h + include/raftwire.h with a stand-in jni.h (tests/jni_stub/jni.h — the JNI
    functions the shim uses, with the specification's signatures; marked as a stand-in);
  * every `native` method of the Java classes has a Java_... function of the same arity in the shim, and the other way round;
  * the shim is DRIVEN through a fake JNIEnv on the host emulation of the kernels (tests/native/jni_harness.c): create / option / loadState /
    submit / readState / hostAlloc through the shim equal the C-ABI called directly."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "integration", "jni", "raftgpu_jni.c")
JAVA = os.path.join(ROOT, "integration", "java", "io", "lubricant", "consensus", "raft", "gpu")
INC = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "jni_stub")]


def test_the_jni_shim_type_checks_against_both_headers():
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    p = subprocess.run(["gcc", "-fsyntax-only", "-std=c11", "-Wall", "-Wextra", "-Werror"] + INC + [SHIM], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "STAND-IN" in open(os.path.join(ROOT, "tests", "jni_stub", "jni.h")).read()


def _split_args(text):
    """top-level comma split (the Java declarations hold generics-free parameter lists, the C ones plain ones)"""
    return [a for a in (x.strip() for x in text.replace("\n", " ").split(",")) if a]


def test_every_native_method_has_its_jni_function_with_the_same_arity():
    c = open(SHIM).read()
    c_funcs = {(m.group(1), m.group(2)): len(_split_args(m.group(3))) - 2           # minus JNIEnv *, jclass
               for m in re.finditer(r"JNICALL J\((\w+), (\w+)\)\(([^)]*)\)", c)}
    java = {}
    for cls in ("GpuTable", "GpuIngress"):
        src = open(os.path.join(JAVA, cls + ".java")).read()
        for m in re.finditer(r"native\s+[\w\[\]]+\s+(\w+)\(([^)]*)\)", src):
            java[(cls, m.group(1))] = len(_split_args(m.group(2)))
    assert java, "no native methods found"
    assert set(java) == set(c_funcs), "only in Java: %s; only in C: %s" % (sorted(set(java) - set(c_funcs)), sorted(set(c_funcs) - set(java)))
    wrong = {k: (java[k], c_funcs[k]) for k in java if java[k] != c_funcs[k]}
    assert not wrong, "arity (java, c): %r" % wrong
    for name in ("GpuRaftFactory.java", "GpuContextManager.java"):
        assert os.path.exists(os.path.join(JAVA, name))
    assert os.path.exists(os.path.join(ROOT, "integration", "java-test", "io", "lubricant", "consensus", "raft", "KryoVectorsTest.java"))


def test_the_shim_behind_a_fake_jnienv_equals_the_c_abi(tmp_path):
    """links the shim + the harness against the host emulation of libraftgpu (tests/devemu: the product's device and host sources compiled for the CPU)"""
    if shutil.which("gcc") is None or shutil.which("g++") is None:
        pytest.skip("no compiler")
    emu = os.path.join(ROOT, "tests", "devemu", "libraftgpu_emu.so")
    if not os.path.exists(emu):
        subprocess.run(["bash", os.path.join(ROOT, "tools", "build_emu.sh")], check=True, timeout=900)
    exe = str(tmp_path / "jni_harness")
    # (the ingress functions of the shim need libraftwire; the harness exercises the table half: stub them out of the link with --unresolved-symbols)
    subprocess.run(["gcc", "-O1", "-std=c11", "-Wall", "-Wextra"] + INC + [os.path.join(ROOT, "tests", "native", "jni_harness.c"), SHIM, emu,
                    "-Wl,--unresolved-symbols=ignore-in-object-files", "-Wl,-rpath," + os.path.dirname(emu), "-lstdc++", "-lpthread", "-o", exe], check=True, timeout=300)
    # wide rows on the lane-serial emulation, then everything again — plus compact rows in / compact outcome rows out / unpack32 — with one OS thread per lane
    for extra, marker in (({"RG_SPLIT": "0"}, "jni shim ok"), ({"RG_SPLIT": "1", "RG_EMU_WAVES": "1"}, "the device-resident tick through the shim")):
        env = dict(os.environ, RG_ALLOW_HOST_EMULATION="1", **extra)
        env.pop("RG_EMU_WAVES", None) if "RG_EMU_WAVES" not in extra else None
        p = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
        assert p.returncode == 0 and "jni shim ok" in p.stdout and marker in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
        assert ("compact outcome rows through the shim" in p.stdout) == ("RG_EMU_WAVES" in extra)
