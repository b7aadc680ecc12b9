// Shared definitions for the procgen2_amd engine (host + gfx950 device code).
//
// Every arithmetic helper that must be bit-exact against the reference lives in a header of
// `PG_HD` inline functions, so the same source is compiled by hipcc for the kernels and by g++
// for the CPU unit tests that compare it with the real libstdc++ / glibc behaviour
// (tests/test_primitives.py).  The product path only ever runs the device instantiation.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PG_HD __host__ __device__ __forceinline__
#define PG_D __device__ __forceinline__
#else
#define PG_HD inline
#define PG_D inline
#endif

// Distribution modes are compile-time constants in the reference (`System_Tilemap::Config`); here a
// game's source is compiled once per mode it offers, with PG_VARIANT selecting the constants, each copy in its own
// namespace with its own factory (build.py VARIANTS; engine.hip make_game maps (game, mode) → factory).
#ifndef PG_VARIANT
#define PG_VARIANT 0
#endif
#define PG_CAT2(a, b) a##b
#define PG_CAT(a, b) PG_CAT2(a, b)
#define PG_VARIANT_NS PG_CAT(variant, PG_VARIANT)
#define PG_FACTORY(name) PG_CAT(PG_CAT(name, _v), PG_VARIANT)

namespace pg {

constexpr int kObsW = 64;
constexpr int kObsH = 64;
constexpr int kObsBytes = kObsW * kObsH * 3;  // coinrun.cpp:24-25,185 — 64×64×3 uint8
constexpr int kNumActions = 15;               // coinrun.cpp:26

// helpers.h:8-9
constexpr float kUnitPx = 16.0f;
constexpr float kPxUnit = 1.0f / 16.0f;

enum GameId : int32_t { kGameCoinrun = 0, kGameMaze = 1, kGameBossfight = 2, kGameClimber = 3, kGameCaveflyer = 4, kGameChaser = 5, kGameJumper = 6, kNumGames = 7 };

}  // namespace pg
