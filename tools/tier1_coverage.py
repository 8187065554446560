#!/usr/bin/env python
"""Share of rows (and of wave-rounds) of a BASELINE stream that leave tier 1 for the general handlers in step32_kernel's deciding wavefront —
counted by the HOST EMULATION of the kernels (tests/devemu, wavefront mode; the GPU build carries no such counter). Test infrastructure.
usage: RG_LIB=tests/devemu/libraftgpu_emu.so RG_EMU_WAVES=1 RG_ALLOW_HOST_EMULATION=1 python tools/tier1_coverage.py [groups=2048] [rounds=32]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rafting_amd import engine, workload  # noqa: E402


def main():
    groups = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    L = engine.lib()
    a, b, c = C.c_long(), C.c_long(), C.c_long()
    import dataclasses
    for number in (3, 5, 2, "2f", "adverse"):
        if number == "adverse":      # bench.py's value_adverse_mix (without the cache misses: they park a group, which is not a general-handler visit per row)
            cfg = dataclasses.replace(workload.config(3, groups), p_conflict=0.005, p_higher_term=0.01, p_timeout=0.03, p_vote_req=0.008, name="config3 adverse mix")
        else:
            cfg = workload.config(number, groups)
        gen = workload.ReplayGenerator(cfg)
        t = engine.Table(cfg.groups, cfg.cluster, cfg.self_slot, cfg.pre_vote)
        t.load_state(gen.initial_state())
        t.submit32(gen.next_batch(rounds))                 # the stream's first launch settles the election pipeline of the generator
        L.rg_emu_slow_counts(C.byref(a), C.byref(b), C.byref(c))
        for _ in range(3):
            t.submit32(gen.next_batch(rounds))
        L.rg_emu_slow_counts(C.byref(a), C.byref(b), C.byref(c))
        waves = a.value / 64.0
        print("config %s (%d groups x %d rounds x 3 launches): %.3f %% of the rows leave tier 1; %.1f %% of the wave-rounds visit the general handlers"
              % (number, groups, rounds, 100.0 * b.value / max(a.value, 1), 100.0 * c.value / max(waves, 1)))
        t.close()


if __name__ == "__main__":
    main()
