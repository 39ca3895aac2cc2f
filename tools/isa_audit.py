"""ISA audit of the gfx950 kernels (round 3): compile a translation unit to assembly and report, per kernel,

  * the register allocation (VGPRs -> resident waves per SIMD: a 512-entry file, granule 8), and
  * "serial load chains": global / buffer loads that are each followed by an `s_waitcnt vmcnt(0)` before the next load is issued.
    vmcnt retires in order, so such a chain is one memory round trip per load however many loads the source "puts in flight".
    hipcc produces them (a) around runtime-optional sources (`if (r2) load`, `c ? load(a) : load(b)`), (b) in rolled
    load -> LDS-store loops, (c) for values loaded before a loop through a dynamic index into the argument segment, whose wait then
    lands in front of every use inside the loop.

    python tools/isa_audit.py norm.hip wgrad.hip            # table on stdout
    python tools/isa_audit.py --chains-only conv.hip

`audit(src)` returns {kernel symbol: {"vgprs": n, "waves": n, "spill": n, "chains": n}} and is what tests/test_isa_audit.py asserts on."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pytorchdeeplearing_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
sys.path.insert(0, ROOT)
from pytorchdeeplearing_amd.build import FLAGS as BUILD_FLAGS          # noqa: E402  (the product's own compiler flags: register counts depend on them)
FLAGS = [f for f in BUILD_FLAGS if f != "-fPIC"] + ["--cuda-device-only", "-S"]


def assembly(src):
    path = src if os.path.isabs(src) else os.path.join(CSRC, src)
    out = os.path.join(tempfile.mkdtemp(prefix="isa_audit_"), os.path.basename(src) + ".s")
    subprocess.run([HIPCC] + FLAGS + [path, "-o", out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    with open(out) as f:
        return f.read()


def chains(body):
    """count load -> vmcnt(0) -> load -> vmcnt(0) sequences (no store in between) in one kernel's instruction list"""
    seq, n = [], 0
    for ln in body:
        t = ln.strip()
        if re.match(r"(global|buffer|flat)_load", t):
            seq.append("L")
        elif re.match(r"(global|buffer|flat)_store", t):
            seq.append("S")
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            seq.append("W")
        elif t.startswith(".LBB"):
            seq.append("|")
        else:
            continue
        if re.search(r"L[|]*W[|]*L[|]*W", "".join(seq[-6:])):
            n += 1
            seq = []
    return n


def audit(src):
    txt = assembly(src)
    res, cur, body = {}, None, []
    for ln in txt.split("\n"):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is None:
            continue
        if ln.strip().startswith("s_endpgm"):
            res.setdefault(cur, {})["chains"] = chains(body)
            cur = None
            continue
        body.append(ln)
    for m in re.finditer(r"\.name:\s+(\S+)\n((?:.*\n){1,14}?)\s+\.vgpr_spill_count:\s+(\d+)", txt):
        name, blk, spill = m.group(1), m.group(2), int(m.group(3))
        v = re.search(r"\.vgpr_count:\s+(\d+)", blk)
        if name in res and v:
            n = int(v.group(1))
            res[name].update(vgprs=n, waves=min(8, 512 // max(8, (n + 7) // 8 * 8)), spill=spill)
    return res


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only = "--chains-only" in sys.argv
    for src in args or ["norm.hip"]:
        for k, v in sorted(audit(src).items(), key=lambda kv: (-kv[1].get("chains", 0), -kv[1].get("vgprs", 0))):
            if only and not v.get("chains"):
                continue
            print("%-12s chains %2d  vgprs %3s  waves/SIMD %s  spill %s  %s" % (src, v.get("chains", 0), v.get("vgprs", "?"), v.get("waves", "?"),
                                                                               v.get("spill", "?"), k[:140]))
