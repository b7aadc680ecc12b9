#!/bin/bash
# usage: PG_GAME=jumper tools/kstats_libs.sh now r05 EXP_TAG ...      (GPU box)
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of a short steady-state run of one game through
# tools/perf_quick.py, once per named library, seconds apart on one box: `now` = the tree's own build, `r05` =
# procgen2_amd/lib_ref/libprocgen2_hip_r05.so, anything else = procgen2_amd/lib/libpg_exp_<tag>.so (tools/build_exp.py).
# The logic kernels' A/B of round 6's second half (static rows, jumper's sub-steps, the gang walk) were taken with it.
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16
R=$GRAFT_REPO_ROOT
G=${PG_GAME:-jumper}
for L in $@; do
  if [ $L = now ]; then A=""; elif [ $L = r05 ]; then A="--lib $R/procgen2_amd/lib_ref/libprocgen2_hip_r05.so"; else A="--lib $R/procgen2_amd/lib/libpg_exp_$L.so"; fi
  rm -rf /tmp/kq
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kq -- python3 $R/tools/perf_quick.py --games $G --check 0x0 --settle 300 --steps 200 $A > /tmp/kq.log 2>&1
  f=$(ls /tmp/kq/*/*kernel_stats.csv | head -1)
  python3 - "$f" $L $G <<'PY'
import csv, sys
out=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    for k in ("logic_kernel","resolve_kernel","setup_kernel","render_kernel"):
        if k in n and sys.argv[3] in n: out.append("%s %.1f"%(k.split('_')[0], float(r["AverageNs"])/1e3))
print(sys.argv[2], " ".join(sorted(out)))
PY
done
