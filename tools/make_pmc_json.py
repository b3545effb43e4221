#!/usr/bin/env python
"""
Build profiles/r02_pmc.json -- the per-kernel constants bench.py's `roofline` is computed from -- out of the rocprofv3
PMC passes collected by tools/collect_profiles.sh (rocpd sqlite databases, one counter group per pass):

    python tools/make_pmc_json.py gpurun_out/prof_c2 gpurun_out/prof_c4 > profiles/r02_pmc.json

Each directory holds sq/ fetch/ write/ sub-runs of ONE bench configuration.  Per stepping kernel:
  valu_insts_per_wave_step = SQ_INSTS_VALU per dispatch / ((paths / 64) * steps)
  hbm_bytes                = 2 * FETCH_SIZE KB * 1024 (the microarch guide's gfx950 correction: 128-B read requests
                             are tallied at 64 B) + WRITE_SIZE KB * 1024 (as reported)
"""
import glob
import json
import os
import sqlite3
import sys

KERNELS = ("logsv_rng_kernel", "logsv_chain_rng_kernel", "logsv_w_kernel")
NOTES = {
    "logsv_chain_rng_kernel": "traffic above the 48 + 8 M algorithmic bytes per path is register-spill scratch: at its 64-VGPR "
                              "budget (8 waves per SIMD) the whole-chain kernel parks 68 B per lane of values that are dead "
                              "inside the time loop (x, qvar, slice bookkeeping, polynomial start constants) in scratch at the "
                              "8 slice boundaries -- ~0.33 GB per launch = 70 GB/s over 4.7 ms, under 1 % of HBM peak and "
                              "overlapped with the VALU-bound stepping (132 B / 0.87 GB before the slice epilogue switched from "
                              "the device libm's exp to exp_full); lifting the cap to 72 / 80 VGPRs (7 / 6 waves) measured the "
                              "same time (profiles/r02_ab_chain_residency.jsonl)",
}


def counters(db_path):
    out = {}
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    pcols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    if not pcols:
        return out
    kn = "kernel_name" if "kernel_name" in pcols else "name"
    for name, counter, n, avg in cur.execute(f"select {kn}, counter_name, count(*), avg(value) from counters_collection "
                                             f"group by {kn}, counter_name"):
        for k in KERNELS:
            if k + "<" in name or k + "(" in name or name.strip().endswith(k):
                out.setdefault(k, {})[counter] = (avg, n)
    return out


def main():
    res = {"_note": "rocprofv3 --pmc passes (SQ_INSTS_VALU; FETCH_SIZE; WRITE_SIZE: separate runs) of `python bench.py "
                    "--config <cfg> --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs`; FETCH_SIZE doubled per "
                    "/opt/skills/guides/MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B), WRITE_SIZE "
                    "as reported; quarter-rate instructions per step (v_rcp_f64, v_rsq_f64) counted in the ISA"}
    for d in sys.argv[1:]:
        meta = json.load(open(os.path.join(d, "config.json")))
        merged = {}
        for sub in ("sq", "fetch", "write"):
            for db in glob.glob(os.path.join(d, sub, "**", "*.db"), recursive=True):
                for k, c in counters(db).items():
                    merged.setdefault(k, {}).update(c)
        for k, c in merged.items():
            paths, steps = meta["paths"], meta["steps"]
            if k == "logsv_w_kernel":
                paths, steps = meta.get("streamed_paths", paths), meta.get("streamed_steps", steps)
            e = {"config": {"paths": paths, "steps": steps}, "bench_config": meta["config"]}
            if "SQ_INSTS_VALU" in c:
                e["sq_insts_valu_per_dispatch"] = c["SQ_INSTS_VALU"][0]
                e["valu_insts_per_wave_step"] = c["SQ_INSTS_VALU"][0] / ((paths / 64.0) * steps)
                e["quarter_rate_insts_per_step"] = 2
                e["source"] = "rocprofv3 --pmc SQ_INSTS_VALU (profiles/r02_pmc.json)"
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                e["fetch_size_kb_raw"], e["write_size_kb_raw"] = c["FETCH_SIZE"][0], c["WRITE_SIZE"][0]
                e["hbm_bytes"] = 2.0 * c["FETCH_SIZE"][0] * 1024.0 + c["WRITE_SIZE"][0] * 1024.0
            for extra in ("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_LDS"):
                if extra in c:
                    e[extra.lower() + "_per_dispatch"] = c[extra][0]
            if k in NOTES:
                e["note"] = NOTES[k]
            if k not in res or meta["config"] == "c2":
                res[k] = e
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
