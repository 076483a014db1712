#!/usr/bin/env python3
"""Per-kernel code-object metadata of libhekaton.so (or any object with embedded gfx950 code objects):
VGPRs / AGPRs, scratch bytes per lane (.private_segment_fixed_size), dynamic-stack flag, spill counts, LDS.

    python tools/kernel_meta.py [file ...] [--check]   # default: hekaton_system_amd/lib/libhekaton.so

--check enforces the bounds DESIGN.md §3a ("The BLS12-381 GPU fault of round 1") establishes; the build runs it
(csrc/Makefile, `__graft_entry__.build()`):
  1. no device FUNCTION (non-kernel symbol) may be larger than the reach of `s_cbranch` (+-32767 dwords = 131 068 B):
     above it hipcc's branch relaxation emits long jumps through `s_getpc_b64 s[30:31]` / `s_setpc_b64 s[30:31]` in
     leaf functions, i.e. through the function's own return address, which a leaf never saves;
  2. no `s_getpc_b64 s[30:31]` ... `s_setpc_b64 s[30:31]` sequence anywhere (the direct signature of that miscompile);
  3. no kernel may use a dynamic stack or need more private memory per lane than SCRATCH_LIMIT_BYTES;
  4. no "half-defined" VGPR read at the join of a divergent if / else (tools/isa_lanecheck.py: the signature of the
     hipcc miscompile behind the psi(0, 0) anomaly of round 2).
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Private memory per lane.  The deepest frames are the serial Fq12 call chains of the pairing kernels (Miller loop ->
# line product -> Fq12 -> Fq6 -> Fq2 product, each level holding a few 384 / 576-byte temporaries): 5.1 KB per lane
# on BLS12-381; the MSM tails need up to 2.8 KB.  ROCr sizes a queue's scratch ring for (bytes per lane) x 64 lanes x
# the device's wave slots and serves a dispatch above HSA_SCRATCH_SINGLE_LIMIT from a use-once allocation, so large
# frames cost launch latency, not correctness (the round-1 faults were NOT scratch).  8 KB is the
# fence against an accidental fully-inlined frame.
SCRATCH_LIMIT_BYTES = 8192
# Scratch budget of the DEFAULT configuration (the library enforces the same bound at run time in
# hk_ctx_create with the box's own numbers): every hardware queue that has run a kernel keeps a ring of frame x 64 lanes x
# (256 CUs x 32 wave slots) bytes, the rings share the agent's 32 GiB, the bindings export GPU_MAX_HW_QUEUES=20.
# Kernels reachable only with HK_PAIR_SERIAL=1 (the one-lane-per-pair debugging path) are not on the default path.
SCRATCH_AGENT_LIMIT = 32 << 30
SCRATCH_RESERVE = 1 << 30
DEFAULT_HW_QUEUES = 20
WAVE_SLOTS = 256 * 32
SERIAL_ONLY = ("k_pair_miller", "k_f12_product", "k_final_exp")
PRESIZE = "k_scratch_presize"       # hk_core.hip: one-wave kernels that only SIZE a queue's ring to the deepest real frame
BRANCH_REACH_BYTES = 32767 * 4      # s_cbranch_*: signed 16-bit dword offset


def function_sizes(co):
    """(size, is_kernel, mangled name) of every function symbol of a code object."""
    out = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "--wide", co], text=True, capture_output=True,
                         check=True).stdout
    funcs, kds = [], set()
    for line in out.splitlines():
        f = line.split()
        if len(f) >= 8 and f[3] == "OBJECT" and f[7].endswith(".kd"):
            kds.add(f[7][:-3])
    for line in out.splitlines():
        f = line.split()
        if len(f) >= 8 and f[3] == "FUNC":
            funcs.append((int(f[2]), f[7] in kds, f[7]))
    return funcs


def retaddr_long_branches(co):
    """Number of long-branch expansions through the return-address pair: `s_getpc_b64 s[30:31]` whose value reaches an
    `s_setpc_b64 s[30:31]` within the next few instructions (getpc / add / addc / setpc).  A lone `s_getpc_b64 s[30:31]`
    is ordinary pc-relative addressing (a kernel loading a constant table, a call sequence ending in s_swappc) and a lone
    `s_setpc_b64 s[30:31]` is an ordinary return."""
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], text=True,
                         capture_output=True, check=True).stdout
    lines = [l.strip() for l in dis.splitlines() if l.startswith("\t") or l.startswith(" ")]
    n = 0
    for i, l in enumerate(lines):
        if l.startswith("s_getpc_b64 s[30:31]"):
            if any(x.startswith("s_setpc_b64 s[30:31]") for x in lines[i + 1:i + 5]):
                n += 1
    return n


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
    rows, funcs, n_getpc = [], [], 0
    for f in files:
        tmp, cos = code_objects(f)
        try:
            for co in cos:
                rows += kernels_of(co)
                funcs += function_sizes(co)
                if check:
                    n_getpc += retaddr_long_branches(co)
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
        if r["dyn_stack"] or (r["scratch"] > SCRATCH_LIMIT_BYTES and PRESIZE not in name):
            bad.append(name)
    fdm = demangle([f[2] for f in funcs]) if funcs else {}
    big = sorted({(sz, k, fdm[n]) for sz, k, n in funcs if sz > BRANCH_REACH_BYTES}, reverse=True)
    largest = max(funcs) if funcs else (0, False, "")
    print("\nlargest function: %d B (%s); s_cbranch reach %d B; %d function(s) above it" % (
        largest[0], fdm.get(largest[2], "")[:80], BRANCH_REACH_BYTES, len(big)))
    for sz, k, n in big:
        print("  %7d B  %s  %s" % (sz, "kernel  " if k else "FUNCTION", n[:150]))
        if not k:
            bad.append("device function larger than the s_cbranch reach (%d B): %s" % (sz, n))
    if check and n_getpc:
        bad.append("%d long branch(es) through the return-address pair s[30:31]" % n_getpc)
    deep = max((r for r in rows if not any(k in dm[r["symbol"]] for k in SERIAL_ONLY + (PRESIZE,))), key=lambda r: r["scratch"],
               default=None)
    if deep:
        presized = min((r["scratch"] for r in rows if PRESIZE in dm[r["symbol"]] and r["scratch"] >= deep["scratch"]),
                       default=deep["scratch"])          # what hk_ctx_create sizes every ring to
        ring = presized * 64 * WAVE_SLOTS
        need = DEFAULT_HW_QUEUES * ring + SCRATCH_RESERVE
        print("scratch budget of the default configuration: deepest default-path frame %d B (%s) -> ring %.2f GiB per queue; "
              "%d queues + reserve = %.1f GiB of %d GiB" % (deep["scratch"], re.sub(r"\(.*", "", dm[deep["symbol"]])[:60], ring / 2**30,
                                                           DEFAULT_HW_QUEUES, need / 2**30, SCRATCH_AGENT_LIMIT >> 30))
        if check and need > SCRATCH_AGENT_LIMIT:
            bad.append("scratch budget: %d queues x %.2f GiB + reserve exceed the agent's %d GiB (frame %d B in %s)" % (
                DEFAULT_HW_QUEUES, ring / 2**30, SCRATCH_AGENT_LIMIT >> 30, deep["scratch"], dm[deep["symbol"]][:80]))
    n_lane = None
    if check:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import isa_lanecheck
        nfunc, hits = isa_lanecheck.run(files)
        n_lane = len(hits)
        for name, ins, P, S in hits[:20]:
            bad.append("half-defined VGPR read at 0x%x (%s) in %s" % (ins.addr, ins.mn, name[:110]))
    if check and bad:
        print("\nkernel_meta --check FAILED (dynamic stack, scratch > %d B per lane, oversized device function, "
              "s[30:31] long branch or half-defined VGPR read):" % SCRATCH_LIMIT_BYTES, file=sys.stderr)
        for b in bad:
            print("  " + b[:200], file=sys.stderr)
        sys.exit(1)
    if check:
        print("kernel_meta --check ok: %d kernels, no dynamic stack, max scratch %d B per lane (limit %d), no device "
              "function above the s_cbranch reach, no s[30:31] long branch, no half-defined VGPR read at an if / else "
              "join" % (len(rows), max(r["scratch"] for r in rows if PRESIZE not in dm[r["symbol"]]), SCRATCH_LIMIT_BYTES))


if __name__ == "__main__":
    main()
