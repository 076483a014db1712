#!/usr/bin/env python3
"""Per-kernel code-object metadata of libhekaton.so (or any object with embedded gfx950 code objects):
VGPRs / AGPRs, scratch bytes per lane (.private_segment_fixed_size), dynamic-stack flag, spill counts, LDS.

    python tools/kernel_meta.py [file ...] [--check]   # default: hekaton_system_amd/lib/libhekaton.so

--check enforces the bounds DESIGN.md §"BLS12-381 fault" establishes (the build runs it, see csrc/Makefile):
no kernel may use a dynamic stack, and no kernel may need more private memory per lane than SCRATCH_LIMIT_BYTES.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# The largest frame any shipped kernel needs is 2.8 KB per lane (12-limb G2 tails).  ROCr sizes a queue's scratch
# ring for (bytes per lane) x 64 lanes x the device's wave slots; above HSA_SCRATCH_SINGLE_LIMIT (140 MB on this
# stack) a dispatch takes the slow "use once" path.  4 KB per lane keeps every kernel far from the flat-scratch
# addressing limits as well (13-bit immediate offsets hold +-4 KB).
SCRATCH_LIMIT_BYTES = 4096


def code_objects(path):
    tmp = tempfile.mkdtemp(prefix="hk_meta_")
    local = os.path.join(tmp, os.path.basename(path))
    shutil.copy(path, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    cos = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
    return tmp, [os.path.join(tmp, f) for f in cos]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), text=True,
                         capture_output=True, check=True).stdout.splitlines()
    return dict(zip(names, out))


def kernels_of(co):
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True, capture_output=True,
                           check=True).stdout
    ks = []
    for blk in re.split(r"\n\s+- \.agpr_count:", "\n" + notes)[1:]:
        blk = ".agpr_count:" + blk
        get = lambda key, d=None: (re.search(r"\.%s:\s+(\S+)" % key, blk) or [None, d])[1]
        ks.append(dict(symbol=get("name"), vgpr=int(get("vgpr_count", 0)), agpr=int(get("agpr_count", 0)),
                       sgpr=int(get("sgpr_count", 0)), scratch=int(get("private_segment_fixed_size", 0)),
                       lds=int(get("group_segment_fixed_size", 0)),
                       dyn_stack=get("uses_dynamic_stack", "false") == "true",
                       vgpr_spill=int(get("vgpr_spill_count", 0)), sgpr_spill=int(get("sgpr_spill_count", 0)),
                       max_wg=int(get("max_flat_workgroup_size", 0))))
    return ks


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    check = "--check" in sys.argv
    files = args or [os.path.join(ROOT, "hekaton_system_amd", "lib", "libhekaton.so")]
    rows = []
    for f in files:
        tmp, cos = code_objects(f)
        try:
            for co in cos:
                rows += kernels_of(co)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    dm = demangle([r["symbol"] for r in rows])
    rows.sort(key=lambda r: (-r["scratch"], dm[r["symbol"]]))
    print("%-6s %-5s %-5s %-7s %-6s %-9s %-4s %s" % ("vgpr", "agpr", "sgpr", "scratch", "lds", "spill v/s", "dyn", "kernel"))
    bad = []
    for r in rows:
        name = re.sub(r"\s*\[clone.*", "", dm[r["symbol"]])
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*", "", name) if len(name) > 150 else name
        print("%-6d %-5d %-5d %-7d %-6d %-9s %-4s %s" % (r["vgpr"], r["agpr"], r["sgpr"], r["scratch"], r["lds"],
                                                         "%d/%d" % (r["vgpr_spill"], r["sgpr_spill"]),
                                                         "YES" if r["dyn_stack"] else "-", name[:150]))
        if r["dyn_stack"] or r["scratch"] > SCRATCH_LIMIT_BYTES:
            bad.append(name)
    if check and bad:
        print("\nkernel_meta --check FAILED (dynamic stack or scratch > %d B per lane):" % SCRATCH_LIMIT_BYTES, file=sys.stderr)
        for b in bad:
            print("  " + b[:200], file=sys.stderr)
        sys.exit(1)
    if check:
        print("\nkernel_meta --check ok: %d kernels, no dynamic stack, max scratch %d B per lane (limit %d)" % (
            len(rows), max(r["scratch"] for r in rows), SCRATCH_LIMIT_BYTES))


if __name__ == "__main__":
    main()
