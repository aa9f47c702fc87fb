#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: mean counter value per dispatch and mean duration, per kernel.

    python tools/pmc_summary.py OUT.json DIR [DIR ...] [--match composite] [--config C3]

The summary is stamped with csrc_sha (bench.csrc_sha(): the device sources it was taken with) and the git HEAD, so that
bench.py only quotes it for the build it belongs to.

Each DIR is the -d directory of one rocprofv3 --pmc pass (separate passes for FETCH_SIZE and WRITE_SIZE, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes).  FETCH_SIZE/WRITE_SIZE are reported by rocprofv3 in KiB;
hbm_bytes_fetch_x2 applies the gfx950 correction (FETCH_SIZE counts 128-B read requests as 64 B)."""
import csv, glob, json, os, sys
from collections import defaultdict


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match, config = None, "C3"
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        args.remove(match)
    if "--config" in sys.argv:
        config = sys.argv[sys.argv.index("--config") + 1]
        args.remove(config)
    out_path, dirs = args[0], args[1:]
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = row["Kernel_Name"]
                    if match and match not in k:
                        continue
                    acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                    dur[k][(f, row["Dispatch_Id"])] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    res = {}
    for k, cs in acc.items():
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        e["dispatches"] = max(len(v) for v in cs.values())
        e["mean_ns_under_pmc"] = sum(dur[k].values()) / max(1, len(dur[k]))
        if "GRBM_GUI_ACTIVE" in e and e["mean_ns_under_pmc"] > 0:
            cyc = e["GRBM_GUI_ACTIVE"] / 8.0                              # summed over the 8 XCDs
            e["clock_GHz"] = cyc / e["mean_ns_under_pmc"]
            if "SQ_INSTS_VALU" in e:
                # compute-side roofline: a SIMD issues at most one wave64 VALU instruction per 2 cycles (v_fma_f32, MI355X guide)
                e["valu_issue_frac"] = e["SQ_INSTS_VALU"] * 2.0 / (1024.0 * cyc)
                e["cycles_per_valu_inst_per_simd"] = 1024.0 * cyc / e["SQ_INSTS_VALU"]
            if "SQ_ACTIVE_INST_VALU" in e and "SQ_INSTS_VALU" in e:
                # SQ_ACTIVE_INST_VALU counts 4-cycle issue quanta per wave: 1 per ordinary instruction, 2 per transcendental
                # (measured: the ratio equals 1 + the transcendental share of the loop).  Two waves of a SIMD overlap their
                # quanta, so quanta x 4 / SIMD-cycles runs up to 2.0 and is NOT a utilisation; it is not reported as one.
                e["valu_quanta_per_inst"] = e["SQ_ACTIVE_INST_VALU"] / e["SQ_INSTS_VALU"]
            if "SQ_WAVE_CYCLES" in e:                                     # quad-cycles of wave residency
                e["mean_waves_per_simd"] = e["SQ_WAVE_CYCLES"] * 4.0 / (1024.0 * cyc)
        if "FETCH_SIZE" in e:
            e["hbm_bytes_fetch_x2"] = e["FETCH_SIZE"] * 1024.0 * 2.0
        if "WRITE_SIZE" in e:
            e["hbm_bytes_write"] = e["WRITE_SIZE"] * 1024.0
        if "hbm_bytes_fetch_x2" in e and "hbm_bytes_write" in e:
            e["hbm_bytes_total"] = e["hbm_bytes_fetch_x2"] + e["hbm_bytes_write"]
        res[k] = e
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    stamp = {"source": dirs, "config": config}
    try:
        import subprocess
        import bench
        stamp["csrc_sha"] = bench.csrc_sha()
        stamp["git_head"] = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception as ex:                                               # the GPU box snapshot has no .git: the sha is what matters
        stamp.setdefault("csrc_sha", None); stamp["stamp_error"] = str(ex)
    with open(out_path, "w") as fh:
        json.dump({**stamp, "kernels": res,
                   "units": "counter values are means per dispatch; FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 reports them; hbm_bytes_fetch_x2 = "
                            "FETCH_SIZE*1024*2 (gfx950 correction, MI355X_MICROARCH.md); valu_issue_frac = SQ_INSTS_VALU*2/(1024 SIMDs * "
                            "GRBM_GUI_ACTIVE/8); mean_waves_per_simd = SQ_WAVE_CYCLES*4/(1024 * GRBM_GUI_ACTIVE/8)"}, fh, indent=1)
    for k, e in sorted(res.items(), key=lambda kv: -kv[1]["mean_ns_under_pmc"]):
        print(f"{e['mean_ns_under_pmc'] / 1e3:10.1f} us  {k[:90]}")
        print("            " + "  ".join(f"{c}={v:.4g}" for c, v in e.items() if c not in ("mean_ns_under_pmc",)))


if __name__ == "__main__":
    main()
