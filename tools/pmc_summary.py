#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: mean counter value per dispatch and mean duration, per kernel.

    python tools/pmc_summary.py OUT.json DIR [DIR ...] [--match composite]

Each DIR is the -d directory of one rocprofv3 --pmc pass (separate passes for FETCH_SIZE and WRITE_SIZE, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes).  FETCH_SIZE/WRITE_SIZE are reported by rocprofv3 in KiB;
hbm_bytes_fetch_x2 applies the gfx950 correction (FETCH_SIZE counts 128-B read requests as 64 B)."""
import csv, glob, json, os, sys
from collections import defaultdict


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        args.remove(match)
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
            if "SQ_ACTIVE_INST_VALU" in e:                                # quad-cycles over 1024 SIMDs
                e["valu_busy_frac"] = e["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc)
            if "SQ_INSTS_VALU" in e:
                e["valu_cycles_per_inst_if_all_simds_busy"] = 1024.0 * cyc / e["SQ_INSTS_VALU"]
        if "FETCH_SIZE" in e:
            e["hbm_bytes_fetch_x2"] = e["FETCH_SIZE"] * 1024.0 * 2.0
        if "WRITE_SIZE" in e:
            e["hbm_bytes_write"] = e["WRITE_SIZE"] * 1024.0
        if "hbm_bytes_fetch_x2" in e and "hbm_bytes_write" in e:
            e["hbm_bytes_total"] = e["hbm_bytes_fetch_x2"] + e["hbm_bytes_write"]
        res[k] = e
    with open(out_path, "w") as fh:
        json.dump({"source": dirs, "kernels": res}, fh, indent=1)
    for k, e in sorted(res.items(), key=lambda kv: -kv[1]["mean_ns_under_pmc"]):
        print(f"{e['mean_ns_under_pmc'] / 1e3:10.1f} us  {k[:90]}")
        print("            " + "  ".join(f"{c}={v:.4g}" for c, v in e.items() if c not in ("mean_ns_under_pmc",)))


if __name__ == "__main__":
    main()
