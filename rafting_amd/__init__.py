"""rafting_amd — MI355X-native batched multi-Raft decision engine.

Host-side mirror of the one hot path of curioloop/rafting that runs on the GPU: the per-RaftContext
EventLoop decision logic (io.lubricant.consensus.raft.context.**).  The product is the C-ABI shared
library `rafting_amd/libraftgpu.so` (include/raftgpu.h); this package only binds it.
"""
from . import abi  # noqa: F401
