#!/usr/bin/env python3
"""The bound table: for every kernel of every game's step, how busy each unit of the machine was
while it ran — vector ALU, scalar unit, LDS, the texture-address path (TA) and its data return (TD), the memory system —
from the counter passes of tools/bound_table.sh, so that the binding one is named by a measurement instead of rediscovered
by ablation every round.

    python tools/bound_table.py gpurun_out/r05_bounds_raw.json [--copy-gbps 5300] > profiles/r05_bounds.md

Units (MI355X: 256 CUs, 1 024 SIMDs, 8 XCDs):
  cycles      GRBM_GUI_ACTIVE / 8 (the counter is summed over the XCDs)
  VALU        4 x SQ_ACTIVE_INST_VALU / (1 024 x cycles): share of SIMD issue time spent on vector instructions
  scalar      4 x SQ_ACTIVE_INST_SCA / (1 024 x cycles) (SALU + SMEM, per wave; the scalar unit is one per CU, so x 4 for its own busy share)
  LDS         SQ_LDS_IDX_ACTIVE / (256 x cycles): share of time a CU's LDS works on indexed operations (+ bank conflicts beside it)
  TA          TA_TA_BUSY_sum / (256 x cycles): the texture-address unit of a CU has work (gathers, loads, stores)
  TD          TD_TD_BUSY_sum / (256 x cycles)
  memory      (2 x FETCH_SIZE + WRITE_SIZE) KB, the gfx950 correction of MI355X_MICROARCH.md, / duration / the box's measured copy rate
  wait        SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: share of a wave's life spent waiting to issue

A second table prices the kernel's own counts with measured unit costs, as if each unit worked alone (µs):
  t_valu      SQ_INSTS_VALU x 4.5 clocks / (1 024 SIMDs x clock)        (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 4.4–4.6 in every kernel here)
  t_gather    SQ_INSTS_VMEM_RD x 17 clocks / (256 CUs x clock)          (tools/probe/gather_rate.hip: 13–17 clocks a wave-level load over <= 8 lines,
                                                                         21 with few lanes active, 45 over 64 L1-resident lines: 17 is the floor of a gather)
  t_lds       SQ_INSTS_LDS x 6 clocks / (256 CUs x clock)               (gather_rate.hip: ds_read_b32 4.3–5.9; wider reads more)
  t_bytes     corrected bytes / the box's copy rate
and names the largest beside the measured time: where measured >> every t_*, the kernel is a chain of waits, not a throughput.
"""
import json
import sys

SIMDS, CUS = 1024.0, 256.0


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    copy_gbps = 5300.0
    for i, a in enumerate(sys.argv):
        if a == "--copy-gbps":
            copy_gbps = float(sys.argv[i + 1])
    args = [a for a in args if a != str(copy_gbps) and a != str(int(copy_gbps))]
    raw = json.load(open(args[0]))
    print("# Bound table — every kernel of every game's step, 65 536 envs, steady state\n")
    print("Source: `%s` (tools/bound_table.sh: one `rocprofv3 --kernel-trace --pmc` pass per counter set over "
          "tools/pmc_quick.py, per-launch averages over the last 8 launches).  Shares are of the kernel's own duration; "
          "memory is against %.0f GB/s, what this pool's boxes copy at (bench.py `roofline.peak_measured`).  "
          "**bold** = the busiest unit.\n" % (args[0], copy_gbps))
    print("| game | kernel | µs | waves | VALU / SALU / LDS / loads per wave | VALU | scalar | LDS (+conflict) | TA | TD | memory | traffic MB (× algorithmic) | wait |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for game, kernels in raw.items():
        for name, c in sorted(kernels.items(), key=lambda kv: -kv[1].get("duration_ns", 0)):
            dur = c.get("duration_ns", 0.0)
            if dur < 1500:  # (make_kernel and the like)
                continue
            short = name.split("::")[-1] if "level_kernel" not in name else "level_kernel (install / generator)"
            cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            waves = c.get("SQ_WAVES", 0.0)

            def share(x):
                return x if x is None else max(0.0, x)
            f = {}
            if cycles:
                if "SQ_ACTIVE_INST_VALU" in c: f["VALU"] = 4 * c["SQ_ACTIVE_INST_VALU"] / (SIMDS * cycles)
                if "SQ_ACTIVE_INST_SCA" in c: f["scalar"] = 4 * c["SQ_ACTIVE_INST_SCA"] / (SIMDS * cycles)
                if "SQ_LDS_IDX_ACTIVE" in c: f["LDS"] = c["SQ_LDS_IDX_ACTIVE"] / (CUS * cycles)
                if "TA_TA_BUSY_sum" in c: f["TA"] = c["TA_TA_BUSY_sum"] / (CUS * cycles)
                if "TD_TD_BUSY_sum" in c: f["TD"] = c["TD_TD_BUSY_sum"] / (CUS * cycles)
            traffic = None
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c and dur:
                traffic = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
                f["memory"] = traffic / (dur * 1e-9) / (copy_gbps * 1e9)
            top = max(f, key=f.get) if f else None

            def cell(k, extra=""):
                if k not in f: return "–"
                s = "%.2f%s" % (f[k], extra)
                return "**%s**" % s if k == top else s
            conflict = ""
            if cycles and "SQ_LDS_BANK_CONFLICT" in c:
                conflict = " (+%.2f)" % (c["SQ_LDS_BANK_CONFLICT"] / (CUS * cycles))
            per_wave = "–"
            if waves:
                per_wave = "%.0f / %.0f / %.0f / %.1f" % (c.get("SQ_INSTS_VALU", 0) / waves, c.get("SQ_INSTS_SALU", 0) / waves,
                                                        c.get("SQ_INSTS_LDS", 0) / waves, c.get("SQ_INSTS_VMEM_RD", 0) / waves)
            wait = "–"
            if c.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in c:
                wait = "%.2f" % (c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"])
            tr = "–"
            if traffic is not None:
                tr = "%.0f" % (traffic / 1e6)
                if "render_kernel" in name: tr += " (%.2f)" % (traffic / (65536 * 12297.0))
            print("| %s | %s | %.1f | %.0f | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (
                game, short, dur / 1e3, waves, per_wave, cell("VALU"), cell("scalar"), cell("LDS", conflict), cell("TA"), cell("TD"),
                cell("memory"), tr, wait))
    print()
    print("## The same kernels priced by their own counts (µs, each unit as if alone; clock from GRBM_GUI_ACTIVE)\n")
    print("Caveats.  t_bytes takes FETCH_SIZE doubled, the guide's correction for wide coalesced loads; for scattered dword gathers it may "
          "count twice — with one backdrop for every env coinrun's render kernel loses 37 % of these bytes and 5 % of its time "
          "(DESIGN.md §3), so `bytes` as the largest entry is an upper bound, not a verdict.  t_gather prices every vector-memory read at "
          "the 17-clock floor; a gather over many lines costs more (gather_rate.hip), a coalesced load less.  measured ÷ largest near 1.2–1.5 "
          "with three entries within a factor of two of each other is what a balanced kernel looks like; 2.5 and more is a chain of "
          "round trips on too few wavefronts (the logic kernels, the level kernels).\n")
    print("| game | kernel | measured µs | t_valu | t_gather (17 clk) | t_lds (6 clk) | t_bytes | largest | measured ÷ largest |")
    print("|---|---|---|---|---|---|---|---|---|")
    for game, kernels in raw.items():
        for name, c in sorted(kernels.items(), key=lambda kv: -kv[1].get("duration_ns", 0)):
            dur = c.get("duration_ns", 0.0)
            if dur < 4000 or not c.get("GRBM_GUI_ACTIVE") or not c.get("SQ_WAVES"):
                continue
            short = name.split("::")[-1] if "level_kernel" not in name else "level_kernel (install / generator)"
            clock = c["GRBM_GUI_ACTIVE"] / 8.0 / dur  # cycles per ns = GHz
            t = {"valu": c.get("SQ_INSTS_VALU", 0) * 4.5 / (SIMDS * clock) / 1e3,
                 "gather": c.get("SQ_INSTS_VMEM_RD", 0) * 17.0 / (CUS * clock) / 1e3,
                 "lds": c.get("SQ_INSTS_LDS", 0) * 6.0 / (CUS * clock) / 1e3}
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                t["bytes"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / (copy_gbps * 1e9) * 1e6
            top = max(t, key=t.get)
            print("| %s | %s | %.1f | %.1f | %.1f | %.1f | %s | %s | %.2f |" % (
                game, short, dur / 1e3, t["valu"], t["gather"], t["lds"], ("%.1f" % t["bytes"]) if "bytes" in t else "–", top,
                dur / 1e3 / t[top] if t[top] else 0.0))
    print()


if __name__ == "__main__":
    main()
