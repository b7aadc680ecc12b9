"""Static instruction inventory of a render kernel by phase.

Compiles one game's .hip with -DPG_MARKS (pg_render.h PG_MARK: an assembler comment at each phase boundary), takes the
render kernel out of the assembly and counts the instructions between consecutive marks by kind.  Textual order, not
execution order: both sides of a branch are counted, loops once.  Good for "where do the instructions live".

    python tools/isa_phases.py coinrun
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
game = sys.argv[1] if len(sys.argv) > 1 else "coinrun"
extra = sys.argv[2:]
src = os.path.join(ROOT, "procgen2_amd", "csrc", game + ".hip")
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-DPG_MARKS", "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off",
                    "-DPG_VARIANT=0", "-S", "--cuda-device-only", "-w", "-o", out, src] + extra, check=True)
    text = open(out).read().split("\n")


def kind(op):
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


start = None
for i, line in enumerate(text):
    if re.match(r"^_Z.*render_kernel.*:\s*(;.*)?$", line):
        start = i
        break
assert start is not None, "render kernel not found"
phases, cur = collections.OrderedDict(), "(entry)"
phases[cur] = collections.Counter()
for line in text[start + 1:]:
    t = line.strip()
    if t.startswith(".Lfunc_end") or t.startswith("s_endpgm") and False:
        break
    m = re.match(r";\s*PGMARK\s+(\S+)", t)
    if m:
        cur = "-> " + m.group(1)
        phases.setdefault(cur, collections.Counter())
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    phases[cur][kind(t.split()[0])] += 1
cols = ["valu", "salu", "lds", "vmem", "smem", "wait", "branch", "barrier", "other"]
print("%-28s" % "instructions up to mark" + "".join("%8s" % c for c in cols) + "%8s" % "all")
tot = collections.Counter()
for name, c in phases.items():
    print("%-28s" % name + "".join("%8d" % c[k] for k in cols) + "%8d" % sum(c.values()))
    tot.update(c)
print("%-28s" % "total" + "".join("%8d" % tot[k] for k in cols) + "%8d" % sum(tot.values()))
for line in text[start:]:
    if "vgpr_count" in line or "sgpr_count" in line or ".lds_size" in line or "NumVgprs" in line or "Occupancy" in line:
        print(line.strip())
    if line.strip().startswith(".Lfunc_end"):
        break
