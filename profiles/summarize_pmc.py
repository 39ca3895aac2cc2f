"""Per-kernel HBM traffic from two rocprofv3 PMC passes (run separately, CSV output):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc -o FETCH_SIZE -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --roofline-steps 0
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc -o WRITE_SIZE -- <same command>
    python profiles/summarize_pmc.py gpurun_out/pmc profiles/r01_pmc_fetch_write_per_kernel.json

FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE under-counts by 2x (64-B requests counted as 32 B,
MI355X_MICROARCH.md HBM section) - the raw value is stored here and doubled by the consumer (bench.py)."""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"(\w+_kernel)I?(.*?)EEvNS", name) or re.search(r"(\w+_kernel)", name)
    if not m:
        return name[:60]
    base = m.group(1)
    base = re.sub(r"^\d+", "", base.split("_GLOBAL__N_1")[-1])
    base = re.sub(r"^\d+", "", base)
    targs = m.group(2) if m.lastindex and m.lastindex >= 2 else ""
    return "%s<%s>" % (base, targs) if targs else base


def load(path, counter):
    per = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            k = short(row["Kernel_Name"])
            per[k][0] += 1
            per[k][1] += float(row["Counter_Value"])
    return per


def build_info():
    """seg_build_info() of the library the passes ran (pytorchdeeplearing_amd/build.py stamps a hash of the sources into it): bench.py only
    reports `traffic` from a summary whose build matches the library it has loaded"""
    import ctypes
    import os
    try:
        lib = os.environ.get("SEGENGINE_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pytorchdeeplearing_amd", "lib", "libsegengine.so")
        l = ctypes.CDLL(lib)
        l.seg_build_info.restype = ctypes.c_char_p
        return l.seg_build_info().decode()
    except Exception as ex:
        return "unknown (%s)" % type(ex).__name__


def main():
    d, out = sys.argv[1], sys.argv[2]
    fetch = load("%s/FETCH_SIZE_counter_collection.csv" % d, "FETCH_SIZE")
    write = load("%s/WRITE_SIZE_counter_collection.csv" % d, "WRITE_SIZE")
    res = {}
    for k in sorted(fetch, key=lambda k: -fetch[k][1]):
        n = fetch[k][0]
        res[k] = {"launches": n, "fetch_kb_raw_per_launch": round(fetch[k][1] / n, 1),
                  "write_kb_per_launch": round(write[k][1] / write[k][0], 1) if k in write and write[k][0] else None}
    tot_f = sum(v[1] for v in fetch.values()); tot_w = sum(v[1] for v in write.values())
    res["_build"] = build_info()
    res["_total"] = {"fetch_gb_corrected_all_launches": round(2 * tot_f / 1e6, 3), "write_gb_all_launches": round(tot_w / 1e6, 3)}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    for k in list(res)[:12]:
        print(k, res[k])


if __name__ == "__main__":
    main()
