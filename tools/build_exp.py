#!/usr/bin/env python3
"""Experiment builds: one game's default-mode object compiled with extra -D flags, linked with the product's other
objects into procgen2_amd/lib/libpg_exp_<tag>.so (travels to the GPU box; tools/perf_quick.py --lib takes it).

    python tools/build_exp.py bossfight g16w8 -DPG_BOSSFIGHT_GANG=16 -DPG_BOSSFIGHT_WAVES=8
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from procgen2_amd import build as b  # noqa: E402

game, tag, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
b.build(verbose=False)
tmp = os.path.join(b.OBJ + "_exp")
os.makedirs(tmp, exist_ok=True)
obj = os.path.join(tmp, "%s_%s.o" % (game, tag))
subprocess.run([b.hipcc()] + b.COMMON + flags + ["-DPG_VARIANT=0", "-c", os.path.join(b.CSRC, game + ".hip"), "-o", obj], check=True)
objs = [os.path.join(b.OBJ, f) for f in sorted(os.listdir(b.OBJ)) if f.endswith(".o") and not f.startswith("engine_g")
        and f != game + "_v0.o"]
out = os.path.join(b.LIB, "libpg_exp_%s.so" % tag)
subprocess.run([b.hipcc(), "--offload-arch=" + b.ARCH, "-shared", "-fPIC", "-o", out, obj] + objs +
               [os.path.join(b.OBJ, "engine_g0.o"), "-lz", "-ldl"], check=True)
print("built", os.path.relpath(out))
