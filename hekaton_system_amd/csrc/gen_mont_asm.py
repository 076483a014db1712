#!/usr/bin/env python3
"""Generates mont_asm.h: one inline-asm block per 8-limb Montgomery product for gfx950.

Product-scanning (FIPS) Montgomery multiplication with a 96-bit column accumulator held in three
scratch VGPRs: every 32x32 partial product is exactly
      v_mad_u64_u32 acc, vcc, x, y, acc        (64-bit multiply-accumulate, carry-out to VCC)
      v_addc_co_u32 acc2, vcc, 0, acc2, vcc    (carry into the third word)
so a product costs 2*N^2 = 128 pairs + N v_mul_lo_u32 + 3 moves per column — no register-pair
shuffling, which is where hipcc's code for the plain C++ CIOS loop spends more issue slots than on
the multiplies themselves (ubench: 537 VALU instructions per product vs ~340 here).
Modulus limbs and -p^-1 are s_mov'ed into clobbered SGPRs (VOP3 cannot take 32-bit literals on gfx9).
The m_j quotient digits and the result words share registers (m_j is dead when t_j is written).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_params import CURVES  # noqa: E402

ACC_LO, ACC_HI, ACC2 = "v4", "v5", "v6"     # scratch VGPRs (clobbered), pair must be even-aligned
ACC = "v[4:5]"
SBASE = 64                                   # s64.. hold the modulus limbs, then -p^-1


def _column_ops(L):
    """mac / shift / lo emitters for the 96-bit column accumulator.

    A column's low 64 bits live in an even-aligned VGPR pair (gfx950 requires that of 64-bit operands), its
    carries in a third register.  Two pairs alternate, A = v[4:5] with carries in v9 and B = v[8:9] with carries
    in v5: when a column ends, its middle word (the odd register of its pair) is the ONLY thing that has to move
    — into the even register of the other pair — because the carry word already sits in that pair's odd
    register.  So a column costs one v_mov; no register is ever cleared either: the very first product uses the
    constant 0 as its addend, and the first carry of every column WRITES the carry register (0 + 0 + carry)."""
    pairs = [("v[4:5]", "v4", "v5", "v9"), ("v[8:9]", "v8", "v9", "v5")]     # (pair, lo, mid, carry register)
    state = {"first_ever": True, "first_in_column": True, "k": 0}

    def mac(x, y):
        pair, _lo, _mid, car = pairs[state["k"]]
        addend = "0" if state["first_ever"] else pair
        L.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (pair, x, y, addend))
        if state["first_in_column"]:
            L.append("v_addc_co_u32 %s, vcc, 0, 0, vcc" % car)
        else:
            L.append("v_addc_co_u32 %s, vcc, 0, %s, vcc" % (car, car))
        state["first_ever"] = False
        state["first_in_column"] = False

    def shift():
        _pair, _lo, mid, _car = pairs[state["k"]]
        state["k"] ^= 1
        L.append("v_mov_b32 %s, %s" % (pairs[state["k"]][1], mid))
        state["first_in_column"] = True

    def lo():
        return pairs[state["k"]][1]

    return mac, shift, lo


ACC_CLOBBERS = ["v4", "v5", "v8", "v9"]


def gen_block(name, p, n):
    inv = (-pow(p, -1, 1 << 32)) % (1 << 32)
    limbs = [(p >> (32 * i)) & 0xFFFFFFFF for i in range(n)]
    # operand numbering: 0..n-1 outputs tm, n..2n-1 inputs a, 2n..3n-1 inputs b
    T = lambda i: "%%%d" % i
    A = lambda i: "%%%d" % (n + i)
    B = lambda i: "%%%d" % (2 * n + i)
    P = lambda i: "s%d" % (SBASE + i)
    SINV = "s%d" % (SBASE + n)
    L = []
    for i, l in enumerate(limbs):
        L.append("s_mov_b32 %s, 0x%08x" % (P(i), l))
    L.append("s_mov_b32 %s, 0x%08x" % (SINV, inv))
    mac, shift, lo = _column_ops(L)

    for i in range(n):
        for j in range(i):
            mac(A(j), B(i - j))
            mac(T(j), P(i - j))
        mac(A(i), B(0))
        L.append("v_mul_lo_u32 %s, %s, %s" % (T(i), lo(), SINV))
        mac(T(i), P(0))
        shift()
    for i in range(n, 2 * n):
        for j in range(i - n + 1, n):
            mac(A(j), B(i - j))
            mac(T(j), P(i - j))
        L.append("v_mov_b32 %s, %s" % (T(i - n), lo()))
        if i < 2 * n - 1:
            shift()
    body = "\\n\\t".join(L)
    outs = ", ".join('"=&v"(t%d)' % i for i in range(n))
    ins = ", ".join('"v"(a.v[%d])' % i for i in range(n)) + ", " + ", ".join('"v"(b.v[%d])' % i for i in range(n))
    clob = ", ".join('"%s"' % c for c in ["vcc"] + ACC_CLOBBERS + ["s%d" % (SBASE + i) for i in range(n + 1)])
    decl = "    u32 " + ", ".join("t%d" % i for i in range(n)) + ";"
    store = " ".join("r.v[%d] = t%d;" % (i, i) for i in range(n))
    return """
// %(name)s: r = a * b * R^-1 mod p  (result in [0, 2p), caller reduces once)
#define HK_MONT_ASM_%(name)s(r, a, b)                                                   \\
    do {                                                                                 \\
    %(decl)s                                                                             \\
        asm("%(body)s"                                                                   \\
            : %(outs)s                                                                   \\
            : %(ins)s                                                                    \\
            : %(clob)s);                                                                 \\
        %(store)s                                                                        \\
    } while (0)
""" % dict(name=name, decl=decl, body=body, outs=outs, ins=ins, clob=clob, store=store), len(L)


M_BASE = 10                                   # v10.. hold the quotient digits m_j of the tied (12-limb) form


def gen_block_tied(name, p, n):
    """Variant for fields whose 3n registers exceed the 30-operand limit of an asm statement: the a-limbs are
    read-write operands that receive the result (a_j is dead exactly when t_j is produced), the quotient
    digits live in clobbered physical VGPRs v8.., so only 2n operands are declared."""
    inv = (-pow(p, -1, 1 << 32)) % (1 << 32)
    limbs = [(p >> (32 * i)) & 0xFFFFFFFF for i in range(n)]
    TA = lambda i: "%%%d" % i                  # "+v": a_i in, t_i out
    B = lambda i: "%%%d" % (n + i)
    M = lambda i: "v%d" % (M_BASE + i)
    P = lambda i: "s%d" % (SBASE + i)
    SINV = "s%d" % (SBASE + n)
    L = []
    for i, l in enumerate(limbs):
        L.append("s_mov_b32 %s, 0x%08x" % (P(i), l))
    L.append("s_mov_b32 %s, 0x%08x" % (SINV, inv))
    mac, shift, lo = _column_ops(L)

    for i in range(n):
        for j in range(i):
            mac(TA(j), B(i - j))
            mac(M(j), P(i - j))
        mac(TA(i), B(0))
        L.append("v_mul_lo_u32 %s, %s, %s" % (M(i), lo(), SINV))
        mac(M(i), P(0))
        shift()
    for i in range(n, 2 * n):
        for j in range(i - n + 1, n):
            mac(TA(j), B(i - j))
            mac(M(j), P(i - j))
        L.append("v_mov_b32 %s, %s" % (TA(i - n), lo()))      # a_{i-n} is dead from this column on
        if i < 2 * n - 1:
            shift()
    body = "\\n\\t".join(L)
    outs = ", ".join('"+&v"(t%d)' % i for i in range(n))
    ins = ", ".join('"v"(b.v[%d])' % i for i in range(n))
    clob = ", ".join('"%s"' % c for c in ["vcc"] + ACC_CLOBBERS + ["v%d" % (M_BASE + i) for i in range(n)] +
                     ["s%d" % (SBASE + i) for i in range(n + 1)])
    decl = "    u32 " + ", ".join("t%d = a.v[%d]" % (i, i) for i in range(n)) + ";"
    store = " ".join("r.v[%d] = t%d;" % (i, i) for i in range(n))
    return """
// %(name)s: r = a * b * R^-1 mod p  (result in [0, 2p), caller reduces once) — tied-operand form
#define HK_MONT_ASM_%(name)s(r, a, b)                                                   \\
    do {                                                                                 \\
    %(decl)s                                                                             \\
        asm("%(body)s"                                                                   \\
            : %(outs)s                                                                   \\
            : %(ins)s                                                                    \\
            : %(clob)s);                                                                 \\
        %(store)s                                                                        \\
    } while (0)
""" % dict(name=name, decl=decl, body=body, outs=outs, ins=ins, clob=clob, store=store), len(L)


S_BASE_V = 10                                 # v8.. scratch for the trial subtraction of the lazy add


def gen_addsub(name, p, n):
    """Modular add / sub / double with the carry chain in VCC: 3n VALU instructions each, against ~6n for what
    hipcc makes of the portable u64 loops.  Lazy fields (4p <= R) work on [0, 2p) representatives and reduce
    against 2p; the others (BLS12-381 Fr) on canonical values against p (a + b < 2p < R still holds)."""
    R = 1 << (32 * n)
    bound = 2 * p if 4 * p <= R else p
    assert 2 * bound <= R
    p2 = [(bound >> (32 * i)) & 0xFFFFFFFF for i in range(n)]
    T = lambda i: "%%%d" % i                  # "+v": a_i in, result out
    B = lambda i: "%%%d" % (n + i)
    P = lambda i: "s%d" % (SBASE + i)
    S = lambda i: "v%d" % (S_BASE_V + i)
    ld = ["s_mov_b32 %s, 0x%08x" % (P(i), l) for i, l in enumerate(p2)]
    sclob = ["s%d" % (SBASE + i) for i in range(n)]

    def trial_sub(L):                         # r = t - 2p if that does not borrow, else t
        # a carry-chain instruction may read only one scalar operand and VCC is one, so the 2p limbs go through
        # the scratch VGPRs first (v_mov literal), then are overwritten by the difference
        for i in range(n):
            L.append("v_mov_b32 %s, 0x%08x" % (S(i), p2[i]))
        L.append("v_sub_co_u32 %s, vcc, %s, %s" % (S(0), T(0), S(0)))
        for i in range(1, n):
            L.append("v_subb_co_u32 %s, vcc, %s, %s, vcc" % (S(i), T(i), S(i)))
        for i in range(n):
            L.append("v_cndmask_b32 %s, %s, %s, vcc" % (T(i), S(i), T(i)))

    add = []
    add.append("v_add_co_u32 %s, vcc, %s, %s" % (T(0), T(0), B(0)))
    for i in range(1, n):
        add.append("v_addc_co_u32 %s, vcc, %s, %s, vcc" % (T(i), T(i), B(i)))
    trial_sub(add)
    dbl = []
    dbl.append("v_add_co_u32 %s, vcc, %s, %s" % (T(0), T(0), T(0)))
    for i in range(1, n):
        dbl.append("v_addc_co_u32 %s, vcc, %s, %s, vcc" % (T(i), T(i), T(i)))
    trial_sub(dbl)
    sub = list(ld)
    sub.append("v_sub_co_u32 %s, vcc, %s, %s" % (T(0), T(0), B(0)))
    for i in range(1, n):
        sub.append("v_subb_co_u32 %s, vcc, %s, %s, vcc" % (T(i), T(i), B(i)))
    sub.append("v_cndmask_b32 %s, 0, -1, vcc" % ACC_LO)                     # all ones iff the chain borrowed
    for i in range(n):
        sub.append("v_and_b32 %s, %s, %s" % (ACC_HI, P(i), ACC_LO))
        if i == 0:
            sub.append("v_add_co_u32 %s, vcc, %s, %s" % (T(0), T(0), ACC_HI))
        else:
            sub.append("v_addc_co_u32 %s, vcc, %s, %s, vcc" % (T(i), T(i), ACC_HI))
    decl = "    u32 " + ", ".join("t%d = a.v[%d]" % (i, i) for i in range(n)) + ";"
    store = " ".join("r.v[%d] = t%d;" % (i, i) for i in range(n))
    outs = ", ".join('"+&v"(t%d)' % i for i in range(n))
    ins = ", ".join('"v"(b.v[%d])' % i for i in range(n))

    def macro(kind, L, has_b, clob_v):
        nonlocal sclob
        clob = ", ".join('"%s"' % c for c in ["vcc"] + clob_v + sclob)
        return """
// %(name)s: r = a %(kind)s mod p on [0, %(bound)s) representatives
#define HK_%(KIND)s_ASM_%(name)s(r, a%(bparam)s)                                          \\
    do {                                                                                 \\
    %(decl)s                                                                             \\
        asm("%(body)s"                                                                   \\
            : %(outs)s                                                                   \\
            : %(ins)s                                                                    \\
            : %(clob)s);                                                                 \\
        %(store)s                                                                        \\
    } while (0)
""" % dict(name=name, bound="2p" if bound == 2 * p else "p", kind={"ADD": "+ b", "SUB": "- b", "DBL": "* 2", "RED": "(one conditional subtraction of the bound)", "CANON": "(one conditional subtraction of p: [0, 2p) -> [0, p))"}[kind], KIND=kind, bparam=", b" if has_b else "",
           decl=decl, body="\\n\\t".join(L), outs=outs, ins=ins if has_b else "", clob=clob, store=store)

    red = []
    trial_sub(red)
    canon = []
    if bound != p:                            # lazy field: memory form is canonical, one more subtraction of p
        p1 = [(p >> (32 * i)) & 0xFFFFFFFF for i in range(n)]
        for i in range(n):
            canon.append("v_mov_b32 %s, 0x%08x" % (S(i), p1[i]))
        canon.append("v_sub_co_u32 %s, vcc, %s, %s" % (S(0), T(0), S(0)))
        for i in range(1, n):
            canon.append("v_subb_co_u32 %s, vcc, %s, %s, vcc" % (S(i), T(i), S(i)))
        for i in range(n):
            canon.append("v_cndmask_b32 %s, %s, %s, vcc" % (T(i), S(i), T(i)))
    sv = [S(i) for i in range(n)]
    noscal = sclob
    sclob = []                                # add / dbl use literals only
    addm, dblm = macro("ADD", add, True, sv), macro("DBL", dbl, False, sv) + macro("RED", red, False, sv)
    if canon:
        dblm += macro("CANON", canon, False, sv)
    sclob = noscal
    return (addm + dblm + macro("SUB", sub, True, [ACC_LO, ACC_HI]),
            (len(add), len(dbl), len(sub)))


def main(path):
    out = ["/* GENERATED by gen_mont_asm.py — do not edit. */", "#pragma once", ""]
    for cname, c in CURVES.items():
        for fname, p, n in (("FR", c["r"], c["fr_n"]), ("FQ", c["q"], c["fq_n"])):
            if n == 8:
                blk, cnt = gen_block("%s_%s" % (cname, fname), p, n)
            else:
                blk, cnt = gen_block_tied("%s_%s" % (cname, fname), p, n)
            out.append("/* %d instructions */" % cnt)
            out.append(blk)
            if True:
                blk, cnts = gen_addsub("%s_%s" % (cname, fname), p, n)
                out.append("/* add / dbl / sub: %d / %d / %d instructions */" % cnts)
                out.append(blk)
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "mont_asm.h"))
