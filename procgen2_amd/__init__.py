"""procgen2_amd — MI355X-native vectorised Procgen2 environment engine.

    procgen2_amd.lib        ctypes binding of the C ABI (include/procgen2_vec.h)
    procgen2_amd.cenv       CEnv: the reference's single-env Python surface over include/procgen2_cenv.h
    procgen2_amd.vec_env    ProcgenVecEnv: N envs, zero-copy torch tensors, optional RCCL gather
    procgen2_amd.build      hipcc build recipe (gfx950)

The compute path is the HIP shared library under procgen2_amd/lib/; nothing here computes on the CPU.
"""
__all__ = ["lib", "cenv", "vec_env", "build"]
