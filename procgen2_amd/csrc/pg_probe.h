// Measurement scaffolding of the render / logic kernels — every macro here is NOTHING in the product build.
//
// The product (procgen2_amd/build.py `build()`) defines none of PG_ABLATE / PG_MARKS / PG_TIMELINE; a CPU test
// (tests/test_build_flags.py) holds the recipe to that.  The experiment builds that do define them are
//   -DPG_ABLATE    lib/libprocgen2_hip_ablate.so (`python -m procgen2_amd.build --ablate`, tools/ablate_render.py):
//                  bits of the debug word switch parts of a kernel off — wrong pictures, right cost;
//   -DPG_MARKS     tools/isa_phases.py: comments in the assembly at which it splits a kernel's instruction inventory;
//   -DPG_TIMELINE  tools/build_exp.py + tools/probe/wave_timeline.py: s_memtime at the phase boundaries of a render
//                  wavefront, left in the first words of the env's own observation (the frame is garbage there).
#pragma once

#ifdef PG_ABLATE
#define PG_ABL(flags, bits) ((flags) & (bits))
#else
#define PG_ABL(flags, bits) 0
#endif

#ifdef PG_MARKS
#define PG_MARK(name) asm volatile("; PGMARK " name)
#else
#define PG_MARK(name) ((void)0)
#endif

#ifdef PG_TIMELINE
#define PG_TL_BEGIN(n) unsigned long long pg_tl_[n]
#define PG_TL(k)                                     \
    do {                                             \
        __builtin_amdgcn_s_waitcnt(0);               \
        __builtin_amdgcn_wave_barrier();             \
        pg_tl_[k] = __builtin_amdgcn_s_memtime();    \
    } while (0)
// the last stamp (index n - 1), then lane 0 leaves all n where `bytes_at` points (when `cond`)
#define PG_TL_END(n, cond, bytes_at)                                                                \
    do {                                                                                            \
        PG_TL((n) - 1);                                                                             \
        if ((threadIdx.x & 63) == 0 && (cond)) {                                                    \
            unsigned long long* pg_tl_out_ = reinterpret_cast<unsigned long long*>(bytes_at);       \
            for (int pg_tl_k_ = 0; pg_tl_k_ < (n); pg_tl_k_++) pg_tl_out_[pg_tl_k_] = pg_tl_[pg_tl_k_]; \
        }                                                                                           \
    } while (0)
#else
#define PG_TL_BEGIN(n) ((void)0)
#define PG_TL(k) ((void)0)
#define PG_TL_END(n, cond, bytes_at) ((void)0)
#endif
