"""The code-object fence against the hipcc miscompile behind round 2's psi(0, 0) anomaly (DESIGN.md section 3b,
profiles/r03_psi_root_cause.txt): tools/isa_lanecheck.py must flag the miscompiled first form of k_points_psi4<Bls381FqP>
(its disassembly is the fixture tests/golden/isa/psi4_bls12_381_first_form.dis: the else arm's copy of limb 10 deleted,
v74 read uninitialised by the lanes holding (0, 0)) and must find nothing in the shipped library.  No GPU needed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lanecheck  # noqa: E402


def test_the_miscompiled_first_form_of_psi4_is_flagged():
    fixture = os.path.join(ROOT, "tests", "golden", "isa", "psi4_bls12_381_first_form.dis")
    nfunc, hits = isa_lanecheck.run([fixture])
    assert nfunc == 1 and len(hits) == 1
    name, ins, path, writes = hits[0]
    assert "k_points_psi4" in name and "Bls381FqP" in name
    assert ins.mn == "scratch_store_dwordx4" and 74 in ins.reads          # the store of limbs 8..11 of conj(y).c1
    # every reaching write of v74 sits one level below the read, in the THEN arm of one if
    assert all(len(w) == len(path) + 1 and w[-1][1] == "T" for w in writes)


def test_the_shipped_library_has_no_half_defined_register_read():
    lib = os.path.join(ROOT, "hekaton_system_amd", "lib", "libhekaton.so")
    assert os.path.exists(lib), "build the library first (python __graft_entry__.py)"
    nfunc, hits = isa_lanecheck.run([lib])
    assert nfunc > 150, "every device function of both curves must have been looked at"
    assert hits == [], [(h[0][:80], hex(h[1].addr)) for h in hits]


def test_covered_recognises_both_arms_and_enclosing_regions():
    c = isa_lanecheck.covered
    P = ((0, "T"),)
    assert c(P, {()})                                           # written under an enclosing region
    assert c(P, {P})
    assert not c(P, {P + ((2, "T"),)})                          # one arm only: the miscompile's shape
    assert c(P, {P + ((2, "T"),), P + ((2, "E"),)})             # both arms
    assert c(P, {P + ((2, "T"),), P + ((2, "E"), (3, "T")), P + ((2, "E"), (3, "E"))})
    assert not c(P, {P + ((2, "T"),), P + ((2, "E"), (3, "T"))})
    assert not c(P, {((1, "T"),)})                              # sibling region
