#!/usr/bin/env python
"""
Build profiles/r03_pmc.json -- the per-kernel constants bench.py's `roofline` is computed from -- out of the rocprofv3
PMC passes collected by tools/collect_profiles.sh (rocpd sqlite databases, one counter group per pass):

    python tools/make_pmc_json.py gpurun_out/prof_c2 gpurun_out/prof_c4 > profiles/r03_pmc.json

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
NOTES = {}


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
        if meta.get("lib_sha256"):
            if res.get("lib_sha256", meta["lib_sha256"]) != meta["lib_sha256"]:
                raise SystemExit("the passes were collected on different builds of libsvmc.so")
            res["lib_sha256"] = meta["lib_sha256"]
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
                e["source"] = "rocprofv3 --pmc SQ_INSTS_VALU"
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                e["fetch_size_kb_raw"], e["write_size_kb_raw"] = c["FETCH_SIZE"][0], c["WRITE_SIZE"][0]
                e["hbm_bytes"] = 2.0 * c["FETCH_SIZE"][0] * 1024.0 + c["WRITE_SIZE"][0] * 1024.0
            for extra in ("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE",
                          "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS"):
                if extra in c:
                    e[extra.lower() + "_per_dispatch"] = c[extra][0]
            if k in NOTES:
                e["note"] = NOTES[k]
            if k not in res or meta["config"] == "c2":
                res[k] = e
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
