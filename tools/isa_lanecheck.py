#!/usr/bin/env python3
"""Lane-level "half-defined VGPR" check over the gfx950 code objects of libhekaton.so.

What it looks for.  hipcc (ROCm 7.2) miscompiled the first form of k_points_psi4<Bls381FqP>: after
`SI Optimize VGPR LiveRange` had given the else-only operands of a divergent if / else IMPLICIT_DEF phi inputs, the
join block's live-in lane mask of a 4-register tuple lacked one register, and `Machine Copy Propagation` deleted the
else arm's only copy into it as dead.  In the emitted ISA the signature is: a VGPR that is READ at the join of a
divergent if / else, while every write that reaches the read sits in ONE arm of that if (the lanes of the other arm read
whatever the register held before - for that kernel an uninitialised register, hence a value that changed from launch
to launch).

How.  Per function: basic blocks from the branch targets, EXEC regions tracked along CFG edges (s_and_saveexec = enter
the then-arm of a new if, s_andn2_saveexec / s_or_saveexec+s_xor exec = switch to its else-arm, s_or exec = leave),
so every instruction has a path of (if, arm) pairs.  Forward data flow of "paths of the writes that may reach here" per
VGPR; a write under path W replaces the reaching writes made under W or deeper.  A read under path P is COVERED when
some reaching write was made under a prefix of P (an enclosing region: all of P's lanes were written), or when both arms
of an if directly under P are covered.  It is REPORTED when it is not covered and every reaching write lies strictly
below P - the read is at the join, the writes are in arms.  (A read in a sibling region - `if (c) x = ..; if (c) use(x)`
- is not reported: lane sets of sibling regions cannot be compared here.)

Known benign hit (round 3, k_pair_lines with a 64-bit `t % n`): the expansion of a 64-bit division joins a 32-bit fast path
and a 64-bit slow path and then runs `v_mad_u64_u32 v[6:7], .., v[6:7]` for a result of which only the LOW register is
used - the high input is written in one arm only and does not matter.  The check cannot see that the high result is dead;
the kernel got a 2-D grid instead of the division.

    python tools/isa_lanecheck.py [libhekaton.so | file.s-disassembly ...] [--all] [-v]

Exit status 1 when anything is reported.  `make` runs it through tools/kernel_meta.py --check.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile
from collections import defaultdict

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RE_FUNC = re.compile(r"^([0-9a-f]+) <(.+)>:$")
RE_INSN = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
RE_TGT = re.compile(r"<(.+)\+0x([0-9a-fA-F]+)>\s*$|<([^+>]+)>\s*$")
RE_V = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
RE_S = re.compile(r"\bs\[(\d+):(\d+)\]")

STORES = ("global_store", "scratch_store", "flat_store", "buffer_store", "ds_write", "ds_store", "global_atomic",
          "flat_atomic", "buffer_atomic", "ds_add", "ds_sub", "ds_min", "ds_max", "ds_and", "ds_or", "ds_xor",
          "ds_inc", "ds_dec", "ds_gws", "buffer_wbl2", "buffer_inv", "global_wb", "global_inv")
RMW_DST = ("v_fmac", "v_mac", "v_pk_fmac", "v_writelane", "v_dot2c", "v_dot4c", "v_dot8c", "v_mfma", "v_smfmac",
           "v_movreld", "v_permlane", "v_swap")


def vregs(tok):
    out = []
    for m in RE_V.finditer(tok):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out += list(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_ops(s):
    return [t.strip() for t in s.split(",")] if s else []


class Insn:
    __slots__ = ("addr", "mn", "ops", "raw", "target", "writes", "reads")


def parse(path):
    """{function name: [Insn]} from `llvm-objdump -d --no-show-raw-insn` text."""
    funcs, cur, base = {}, None, {}
    text = open(path, errors="replace").read().splitlines()
    for line in text:
        m = RE_FUNC.match(line)
        if m:
            cur = []
            funcs[m.group(2)] = cur
            base[m.group(2)] = int(m.group(1), 16)
            continue
        if cur is None:
            continue
        m = RE_INSN.match(line)
        if not m:
            continue
        ins = Insn()
        ins.mn, opstr, ins.addr = m.group(1), m.group(2), int(m.group(3), 16)
        ins.raw = line
        ins.target = None
        if ins.mn.startswith("s_cbranch") or ins.mn == "s_branch":
            t = RE_TGT.search(line)
            if t:
                if t.group(1):
                    ins.target = (t.group(1), int(t.group(2), 16))
                else:
                    ins.target = (t.group(3), 0)
            opstr = ""
        ins.ops = split_ops(opstr)
        classify(ins)
        cur.append(ins)
    for name, insns in funcs.items():
        for ins in insns:
            if ins.target:
                ins.target = base.get(ins.target[0], None) and base[ins.target[0]] + ins.target[1]
    return funcs


def classify(ins):
    mn, ops = ins.mn, ins.ops
    ins.writes, ins.reads = [], []
    if mn.startswith("s_") or not ops:
        return
    if mn.startswith(STORES):
        # returning atomics (sc0 / glc) write their first operand; plain stores write nothing
        ret = any(o.split()[-1] in ("sc0", "glc") or " sc0" in o or " glc" in o for o in ops) and "atomic" in mn
        if ret:
            ins.writes = vregs(ops[0])
            ops = ops[1:]
        for o in ops:
            ins.reads += vregs(o)
        return
    if mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
        for o in ops[1:] if not mn.startswith("v_cmpx") and not mn.endswith("_e32") else ops:
            ins.reads += vregs(o)
        return
    if mn == "v_nop":
        return
    ndst = 2 if mn.startswith("v_swap") else 1
    for o in ops[:ndst]:
        ins.writes += vregs(o)
    for o in ops[ndst:]:
        ins.reads += vregs(o)
    if mn.startswith(RMW_DST) or "_dpp" in mn or "_sdwa" in mn:
        ins.reads += ins.writes
    if mn.startswith("s_swappc") or mn.startswith("s_setpc"):
        return


def is_prefix(a, b):
    return len(a) <= len(b) and b[:len(a)] == a


def covered(P, S):
    """True when the union of the writes' lane sets (paths S) contains the lanes of path P."""
    for W in S:
        if is_prefix(W, P):
            return True
    below = [W for W in S if len(W) > len(P) and W[:len(P)] == P]
    by_if = defaultdict(set)
    for W in below:
        by_if[W[len(P)][0]].add(W)
    for ifid, ws in by_if.items():
        if covered(P + ((ifid, "T"),), ws) and covered(P + ((ifid, "E"),), ws):
            return True
    return False


def analyse(name, insns, is_kernel, verbose=False, report_all=False):
    if not insns:
        return []
    addr_ix = {ins.addr: i for i, ins in enumerate(insns)}
    leaders = {0}
    for i, ins in enumerate(insns):
        if ins.mn.startswith("s_cbranch") or ins.mn == "s_branch":
            if ins.target in addr_ix:
                leaders.add(addr_ix[ins.target])
            if i + 1 < len(insns):
                leaders.add(i + 1)
        if ins.mn in ("s_endpgm", "s_setpc_b64") and i + 1 < len(insns):
            leaders.add(i + 1)
    starts = sorted(leaders)
    bid_of = {}
    blocks = []
    for k, s in enumerate(starts):
        e = starts[k + 1] if k + 1 < len(starts) else len(insns)
        blocks.append((s, e))
        bid_of[s] = k
    succ = [[] for _ in blocks]
    for k, (s, e) in enumerate(blocks):
        last = insns[e - 1]
        if last.mn == "s_branch":
            if last.target in addr_ix:
                succ[k].append(bid_of[addr_ix[last.target]])
        elif last.mn.startswith("s_cbranch"):
            if last.target in addr_ix:
                succ[k].append(bid_of[addr_ix[last.target]])
            if e < len(insns):
                succ[k].append(bid_of[e])
        elif last.mn in ("s_endpgm", "s_setpc_b64"):
            pass
        elif e < len(insns):
            succ[k].append(bid_of[e])

    # ---- EXEC regions along CFG edges: stack of (if id, arm, save register) ------------------------------
    entry_stack = {0: ()}
    ins_path = [None] * len(insns)
    unclear = []          # why the EXEC regions of this function could not be followed (then nothing is reported for it)
    work = [0]
    next_if = [0]
    seen_exec = {}
    has_else = set()      # ifs for which the compiler emitted an else arm

    def step(stack, i):
        # stack entries: (if id, arm, names of the SGPR pairs that hold this if's saved / remaining mask)
        ins = insns[i]
        mn, ops = ins.mn, ins.ops
        if mn == "s_and_saveexec_b64" and len(ops) == 2:
            key = ("if", ins.addr)
            if key not in seen_exec:
                seen_exec[key] = next_if[0]
                next_if[0] += 1
            # a save register that an open region still names has been reused: that region's mask lives elsewhere now
            stack = tuple((a, b, tuple(n for n in c if n != ops[0])) for a, b, c in stack)
            return stack + ((seen_exec[key], "T", (ops[0],)),)
        if mn == "s_xor_b64" and len(ops) == 3 and ops[1] == "exec" and ops[0] != "exec":
            # sD = exec ^ saved: the lanes left for the else arm, kept under another name
            for d in range(len(stack) - 1, -1, -1):
                if ops[2] in stack[d][2]:
                    return stack[:d] + ((stack[d][0], stack[d][1], stack[d][2] + (ops[0],)),) + stack[d + 1:]
            return stack
        if mn in ("s_andn2_saveexec_b64", "s_or_saveexec_b64") and len(ops) == 2 and ops[1] != "-1":
            for d in range(len(stack) - 1, -1, -1):
                if ops[1] in stack[d][2]:
                    has_else.add(stack[d][0])
                    # s_andn2_saveexec: EXEC = the else lanes at once.  s_or_saveexec: EXEC = then + else lanes (the whole
                    # enclosing region) until the `s_xor_b64 exec, exec, D` that follows - writes in between reach every
                    # lane of the parent, so the region is "pending" ("P": not part of an instruction's path) until then
                    arm = "E" if mn == "s_andn2_saveexec_b64" else "P"
                    return stack[:d] + ((stack[d][0], arm, (ops[0],)),)
            return stack
        if mn == "s_or_b64" and len(ops) == 3 and ops[0] == "exec" and ops[1] == "exec":
            for d in range(len(stack) - 1, -1, -1):
                if ops[2] in stack[d][2]:
                    return stack[:d]
            if not (stack and stack[-1][1] == "U"):
                unclear.append("0x%x: s_or_b64 exec with a mask that no open region saved" % ins.addr)
            while stack and stack[-1][1] == "U":                 # lanes re-joining after a loop with divergent exits
                stack = stack[:-1]
            return stack
        if mn == "s_mov_b64" and len(ops) == 2 and ops[0] == "exec":
            if stack and stack[-1][1] == "W" and ops[1] in stack[-1][2]:
                return stack[:-1]                                # end of a whole-wave section (SGPR spill code)
            unclear.append("0x%x: EXEC set from %s" % (ins.addr, ops[1]))
            return stack
        if mn == "s_or_saveexec_b64" and len(ops) == 2 and ops[1] == "-1":
            return stack + ((-1, "W", (ops[0],)),)
        if mn in ("s_and_b64", "s_andn2_b64", "s_xor_b64") and len(ops) == 3 and ops[0] == "exec" and "exec" in ops[1:]:
            other = ops[2] if ops[1] == "exec" else ops[1]
            if mn == "s_xor_b64" and stack and stack[-1][1] == "P" and other in stack[-1][2]:
                return stack[:-1] + ((stack[-1][0], "E", stack[-1][2]),)     # second half of the else lowering
            # EXEC narrowed to a lane set this analysis cannot name (a mask collected inside arms, a loop's live lanes):
            # reads below it are never reported
            key = ("u", ins.addr)
            if key not in seen_exec:
                seen_exec[key] = next_if[0]
                next_if[0] += 1
            return stack + ((seen_exec[key], "U", ()),)
        return stack

    while work:
        b = work.pop()
        stack = entry_stack[b]
        s, e = blocks[b]
        for i in range(s, e):
            ins_path[i] = tuple((x[0], x[1]) for x in stack if x[1] != "P")
            stack = step(stack, i)
        for t in succ[b]:
            if t not in entry_stack:
                entry_stack[t] = stack
                work.append(t)
            elif ([(x[0], x[1]) for x in entry_stack[t] if x[1] in "TE"] != [(x[0], x[1]) for x in stack if x[1] in "TE"]
                  and insns[blocks[t][0]].mn != "s_endpgm"):
                # (tail duplication gives one source-level if several s_and_saveexec instructions that share a join
                # block; the first path to reach a block names its regions)
                unclear.append("block 0x%x entered with different EXEC regions" % insns[blocks[t][0]].addr)

    # ---- reaching write paths per VGPR ---------------------------------------------------------------------
    IN = [dict() for _ in blocks]
    OUT = [None] * len(blocks)
    reports = {}
    # registers that hold values on entry: the work-item id of a kernel (v0, packed; v1 / v2 on older layouts); the
    # argument registers and the callee-saved groups (v40-47, v56-63, ...: saved by reading them) of a function
    top = frozenset({()})
    if is_kernel:
        for r in (0, 1, 2):
            IN[0][r] = top
    else:
        for r in range(0, 32):
            IN[0][r] = top
        for g in range(40, 256, 16):
            for r in range(g, g + 8):
                IN[0][r] = top

    def transfer(b, state, collect):
        s, e = blocks[b]
        for i in range(s, e):
            ins = insns[i]
            P = ins_path[i]
            if P is None:
                continue
            if collect:
                for r in ins.reads:
                    S = state.get(r)
                    if not S:
                        continue
                    if covered(P, S):
                        continue
                    if all(len(W) > len(P) and W[:len(P)] == P for W in S):
                        arms = {W[len(P)] for W in S}
                        # the signature of the miscompile: every reaching write in ONE arm of ONE if directly under the
                        # read's region, and the compiler DID emit the other arm for that if - which then lacks the
                        # write (the deleted copy).  Without an else arm, or with writes spread over several ifs, the
                        # read belongs to the structurizer's accumulated-mask regions (`a && b` conditions, guarded
                        # uses) whose lane sets this analysis cannot compare: listed only with --all.
                        if (len(arms) == 1 and next(iter(arms))[0] in has_else) or report_all:
                            reports.setdefault((ins.addr, r), (ins, P, frozenset(S)))
            for r in ins.writes:
                S = state.get(r, frozenset())
                S = frozenset(W for W in S if not is_prefix(P, W)) | {P}
                state[r] = S
            if ins.mn.startswith("s_swappc"):                    # a call defines the return registers for all its lanes
                for r in range(0, 32):
                    S = state.get(r, frozenset())
                    state[r] = frozenset(W for W in S if not is_prefix(P, W)) | {P}
        return state

    changed = True
    order = sorted(entry_stack)
    rounds = 0
    while changed and rounds < 50:
        changed = False
        rounds += 1
        for b in order:
            st = transfer(b, dict(IN[b]), False)
            if OUT[b] != st:
                OUT[b] = st
                for t in succ[b]:
                    tin = IN[t]
                    for r, S in st.items():
                        u = tin.get(r, frozenset()) | S
                        if u != tin.get(r):
                            tin[r] = u
                            changed = True
    out = [("unclear", unclear)] if unclear else []
    for b in order:
        transfer(b, dict(IN[b]), True)
    return out + list(reports.values())


def disassemble(path):
    """[.dis files] for a shared object with embedded code objects, or [path] if it already is a disassembly."""
    if path.endswith((".dis", ".txt")):
        return None, [path]
    tmp = tempfile.mkdtemp(prefix="hk_lane_")
    local = os.path.join(tmp, os.path.basename(path))
    shutil.copy(path, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, cwd=tmp,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    cos = sorted(f for f in os.listdir(tmp) if "amdgcn" in f) or [os.path.basename(path)]
    out = []
    for co in cos:
        d = os.path.join(tmp, co + ".dis")
        with open(d, "w") as fh:
            subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(tmp, co)],
                           check=True, stdout=fh)
        out.append(d)
    return tmp, out


def kernel_symbols(co):
    out = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "--wide", co], text=True, capture_output=True).stdout
    return {f.split()[7][:-3] for f in out.splitlines() if len(f.split()) >= 8 and f.split()[7].endswith(".kd")}


def demangle(name):
    return subprocess.run(["c++filt", name], text=True, capture_output=True).stdout.strip() or name


UNCLEAR = []


def run(files, verbose=False, report_all=False):
    total, nfunc = [], 0
    for f in files:
        tmp, dis = disassemble(f)
        try:
            for d in dis:
                funcs = parse(d)
                kernels = kernel_symbols(os.path.splitext(d)[0]) if d.endswith(".dis") and os.path.exists(os.path.splitext(d)[0]) else None
                for name, insns in funcs.items():
                    nfunc += 1
                    is_kernel = (name in kernels) if kernels is not None else insns[-1].mn != "s_setpc_b64" and any(i.mn == "s_endpgm" for i in insns)
                    for rep in analyse(name, insns, is_kernel, verbose, report_all):
                        if rep[0] == "unclear":
                            UNCLEAR.append((name, rep[1]))
                        else:
                            total.append((name,) + rep)
        finally:
            if tmp:
                shutil.rmtree(tmp, ignore_errors=True)
    return nfunc, total


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    files = args or [os.path.join(ROOT, "hekaton_system_amd", "lib", "libhekaton.so")]
    nfunc, total = run(files, "-v" in sys.argv, "--all" in sys.argv)
    byfn = defaultdict(list)
    for name, ins, P, S in total:
        byfn[name].append((ins, P, S))
    for name, hits in byfn.items():
        print("%s\n  %s" % (demangle(name)[:160], name[:120]))
        for ins, P, S in hits[:8]:
            regs = sorted({r for r in ins.reads})
            print("    0x%x  %-28s read under %s; reaching writes only under %s" % (
                ins.addr, (ins.mn + " " + ",".join(ins.ops))[:60], list(P), sorted(map(list, S))[:3]))
        if len(hits) > 8:
            print("    ... %d more" % (len(hits) - 8))
    if UNCLEAR and "-v" in sys.argv:
        print("EXEC regions followed only approximately in %d function(s)" % len(UNCLEAR))
        for name, why in UNCLEAR:
            print("  %s\n      %s%s" % (demangle(name)[:150], why[0], " (+%d)" % (len(why) - 1) if len(why) > 1 else ""))
    print("isa_lanecheck: %d function(s), %d half-defined VGPR read(s) in %d function(s)" % (nfunc, len(total), len(byfn)))
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
