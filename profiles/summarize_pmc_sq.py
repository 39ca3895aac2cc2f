"""Per-kernel MFMA / LDS / wave counters from rocprofv3 SQ PMC passes (each pass run separately, kernel trace only, CSV output):

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
        -d gpurun_out/pmc_sq -o mfma -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --roofline-steps 0 --condition-seconds 0 --launch stream
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace ... -o lds -- <same command>
    python profiles/summarize_pmc_sq.py gpurun_out/pmc_sq profiles/r05_mfma_util_per_kernel.json mfma lds

Per kernel symbol (template arguments kept): launches, average duration under the counter pass (dispatches are serialised there, so this is the kernel
ALONE, not next to the other queue), every counter averaged per launch, and the derived figures

  mfma_util      = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)      (MfmaUtil of counter_defs.yaml; GRBM_GUI_ACTIVE arrives SUMMED over the 8 XCDs:
                   847 118 for a 41.6 us kernel = 8 x 2.54 GHz x 41.6 us; one 16x16x32 f16 MFMA = 32 MOPS = 16 busy cycles on its SIMD)
  mfma_tflops    = SQ_INSTS_VALU_MFMA_MOPS_F16 x 512 flops / duration                  (what the matrix cores actually executed, padding included)
  mfma_frac_peak = mfma_tflops / 2500                                                  (dense f16 peak, MI355X_MICROARCH.md)
  lds_wait_frac  = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES,  wait_any_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES     (both in quad-cycles)
  lds_conflict   = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_pmc import build_info, short      # noqa: E402

PEAK_TF = 2500.0
SIMDS = 1024.0
XCDS = 8.0


def find(d, prefix, suffix):
    c = glob.glob(os.path.join(d, "**", "%s_%s.csv" % (prefix, suffix)), recursive=True) + glob.glob(os.path.join(d, "%s_%s.csv" % (prefix, suffix)))
    return c[0] if c else None


def main():
    d, out, prefixes = sys.argv[1], sys.argv[2], sys.argv[3:]
    per = collections.defaultdict(lambda: {"launches": {}, "dur_ns": {}, "ctr": collections.defaultdict(float)})
    for pf in prefixes:
        cc, kt = find(d, pf, "counter_collection"), find(d, pf, "kernel_trace")
        if not cc:
            print("no counter file for pass", pf)
            continue
        dur = {}
        if kt:
            with open(kt) as f:
                for row in csv.DictReader(f):
                    dur[row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        seen = set()
        with open(cc) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                p = per[k]
                p["ctr"][(pf, row["Counter_Name"])] += float(row["Counter_Value"])
                did = row["Dispatch_Id"]
                if (pf, did) not in seen:
                    seen.add((pf, did))
                    p["launches"][pf] = p["launches"].get(pf, 0) + 1
                    p["dur_ns"][pf] = p["dur_ns"].get(pf, 0) + dur.get(did, 0)
    res = {"_build": build_info(), "_note": __doc__.split("Per kernel symbol")[1].strip()}
    rows = []
    for k, p in per.items():
        n = max(p["launches"].values())
        c = {name: v / max(1, p["launches"].get(pf, 0)) for (pf, name), v in p["ctr"].items()}      # averaged over the launches of ITS pass
        dur_pf = next((q for q in prefixes if p["dur_ns"].get(q)), None)
        avg_ns = p["dur_ns"][dur_pf] / p["launches"][dur_pf] if dur_pf else 0.0
        r = {"launches": n, "avg_us_alone": round(avg_ns * 1e-3, 2), "counters_per_launch": {a: round(b, 1) for a, b in sorted(c.items())}}
        g = c.get("GRBM_GUI_ACTIVE")
        if g and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            r["mfma_util"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (g / XCDS * SIMDS), 4)
        if avg_ns and "SQ_INSTS_VALU_MFMA_MOPS_F16" in c:
            tf = c["SQ_INSTS_VALU_MFMA_MOPS_F16"] * 512.0 / avg_ns * 1e-3
            r["mfma_tflops"] = round(tf, 1)
            r["mfma_frac_peak"] = round(tf / PEAK_TF, 4)
        w = c.get("SQ_WAVE_CYCLES")
        if w:
            if "SQ_WAIT_INST_LDS" in c:
                r["lds_wait_frac"] = round(c["SQ_WAIT_INST_LDS"] / w, 4)
            if "SQ_WAIT_ANY" in c:
                r["wait_any_frac"] = round(c["SQ_WAIT_ANY"] / w, 4)
        if c.get("SQ_ACTIVE_INST_LDS") and "SQ_LDS_BANK_CONFLICT" in c:
            r["lds_conflict"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_ACTIVE_INST_LDS"], 4)
        rows.append((n * avg_ns, k, r))
    for _, k, r in sorted(rows, key=lambda t: -t[0]):
        res[k] = r
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    for _, k, r in sorted(rows, key=lambda t: -t[0])[:14]:
        print(k[:70], {a: b for a, b in r.items() if a != "counters_per_launch"})


if __name__ == "__main__":
    main()
