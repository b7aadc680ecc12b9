"""procgen2_amd — MI355X-native vectorised Procgen2 environment engine.

    procgen2_amd.lib        ctypes binding of the C ABI (include/procgen2_vec.h)
    procgen2_amd.cenv       CEnv: the reference's single-env Python surface over include/procgen2_cenv.h
    procgen2_amd.vec_env    ProcgenVecEnv: N envs, zero-copy torch tensors, optional RCCL gather
    procgen2_amd.build      hipcc build recipe (gfx950)

The compute path is the HIP shared library under procgen2_amd/lib/; nothing here computes on the CPU.
"""
import os as _os

# Several engines on one GPU (BASELINE.json configs[4]: seven games side by side, two HIP streams each) want more
# hardware queues than the HIP runtime's default of four: streams that share a queue run their kernels one after the
# other, and the seven step chains stop overlapping (measured on MI355X: 75.3 M env-steps/s with 4 queues, 77.7 with 8,
# 82.5 with 12 or 16).  The runtime reads the variable when it initialises — at the first HIP call of the process — so it
# is set here, at import, unless the caller has chosen a value.  The shared library does the same when it is loaded
# (engine.hip), for callers that bind the C ABI directly.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

__all__ = ["lib", "cenv", "vec_env", "build"]
