#!/bin/bash
# SQ instruction counters and HBM traffic of every game's render kernel (default modes, 65 536 envs): three PMC passes
# per game (SQ set, FETCH_SIZE, WRITE_SIZE — separate passes as the TCC counters require).  Output: gpurun_out/<tag>.json
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04_s}
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16   # the profiler initialises HIP before the library can: same configuration as the bench line
for G in coinrun maze bossfight climber caveflyer chaser jumper; do
  for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    D=/tmp/pmc_${G}_$(echo $SET | tr ' ' '_' | cut -c1-24)
    rm -rf $D
    timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $D -- python3 $R/bench.py --game $G --steps 16 --warmup 4 --no-cpu-baseline > /dev/null 2>&1
  done
done
python3 - <<PY
import csv,glob,collections,json
out={}
for g in ["coinrun","maze","bossfight","climber","caveflyer","chaser","jumper"]:
    for kernel in ("render_kernel","setup_kernel"):   # setup_kernel: the render pre-pass of the games that have one
        agg=collections.defaultdict(float); cnt=collections.Counter()
        for fn in glob.glob("/tmp/pmc_%s_*/*/*counter_collection.csv"%g):
            for r in csv.DictReader(open(fn)):
                if kernel not in r["Kernel_Name"]: continue
                agg[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
        c={k:agg[k]/cnt[k] for k in agg}
        if not c: continue
        waves=c.get("SQ_WAVES",1)
        d={"per_launch":c,
           "valu_per_wave":c.get("SQ_INSTS_VALU",0)/waves,"salu_per_wave":c.get("SQ_INSTS_SALU",0)/waves,
           "vmem_reads_per_wave":c.get("SQ_INSTS_VMEM_RD",0)/waves,"lds_per_wave":c.get("SQ_INSTS_LDS",0)/waves}
        if "GRBM_GUI_ACTIVE" in c and "SQ_ACTIVE_INST_VALU" in c:
            d["valu_active_fraction_of_simd_time"]=4*c["SQ_ACTIVE_INST_VALU"]/(1024*c["GRBM_GUI_ACTIVE"]/8)
            if c.get("SQ_INSTS_VALU"): d["clocks_per_valu_instruction"]=4*c["SQ_ACTIVE_INST_VALU"]/c["SQ_INSTS_VALU"]
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            d["hbm_bytes_per_launch_corrected"]=(2*c["FETCH_SIZE"]+c["WRITE_SIZE"])*1024   # gfx950: FETCH_SIZE counts 128-B requests as 64 B
            d["algorithmic_bytes_per_launch"]=65536*12297
        out[g if kernel=="render_kernel" else g+"::setup_kernel"]=d
json.dump({"note":"rocprofv3 --kernel-trace --pmc passes of bench.py --game G --steps 16 (tools/pmc_all_games.sh); per-launch averages of each game's render kernel (and of its render pre-pass, <game>::setup_kernel, where it has one); SQ_* cycle counters in quad-cycles summed over SIMDs, GRBM_GUI_ACTIVE summed over the 8 XCDs","games":out},open("$R/gpurun_out/${TAG}_all_games_render_counters.json","w"),indent=1)
for g,d in out.items(): print(g, "VALU/wave %.0f"%d["valu_per_wave"], "reads/wave %.0f"%d["vmem_reads_per_wave"], "valu_active %.2f"%d.get("valu_active_fraction_of_simd_time",0), "traffic/alg %.2f"%(d.get("hbm_bytes_per_launch_corrected",0)/d.get("algorithmic_bytes_per_launch",1)))
PY
