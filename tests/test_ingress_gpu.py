"""N2 + a8's host half in front of the real kernels: wire frames and the host's own rows -> rafting_amd/host/ingress -> the sealed multi-round
compact batch through rg_submit32 (step32_kernel on the GPU), its wide leftovers through a sparse rg_submit. tests/ingress_flow.py holds every
group's rows, replies and response frames to the oracle's row-by-row decisions (the same flow runs on the host emulation of the kernels in
the CPU suite: tests/devemu/emu_cases_waves.py)."""
import pytest

from rafting_amd import engine
from tests import ingress_flow
from tests.helpers import compare_states

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("groups,cluster,self_slot,pre_vote,rounds,max_rounds,seed", [(192, 5, 2, True, 48, 64, 91), (1000, 3, 0, True, 20, 6, 92),
                                                                                       (130, 7, 6, False, 30, 1, 93)])
def test_ingress_batches_are_decided_like_the_history_row_by_row(groups, cluster, self_slot, pre_vote, rounds, max_rounds, seed):
    st0, batches, outs, final = ingress_flow.history(groups, cluster, self_slot, pre_vote, rounds, seed, view=engine.Table(groups, cluster, self_slot, pre_vote))
    gpu = engine.Table(groups, cluster, self_slot, pre_vote)
    gpu.load_state(st0)
    nodes = [("10.1.0.%d" % i, 7000 + i) for i in range(cluster)]
    sealed = ingress_flow.drive(lambda b32: gpu.submit32(b32), lambda sp: gpu.submit(sp), groups, cluster, batches, outs, max_rounds, nodes)
    assert sealed >= (rounds + max_rounds - 1) // max_rounds
    compare_states(final, gpu.read_state(), "after the ingress")
