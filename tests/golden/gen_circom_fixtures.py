#!/usr/bin/env python3
"""Extracts the reference's own known-answer vectors for the circom file formats into binary fixtures.

The only golden vectors zhaowenlan1779/hekaton-system holds anywhere are the hex-literal `.r1cs` / `.wtns`
blobs and the text witness inside its circom-compat unit tests (circom-compat/src/lib.rs:548-737).  This script
(run once in the build container, where /root/reference is mounted) copies those DATA bytes — not source — to
    tests/golden/circom_sample.r1cs     lib.rs:549-606  (`sample`, and again `test_write` :647-704)
    tests/golden/circom_sample.wtns     lib.rs:724-733  (`wtns_bin_file`)
    tests/golden/circom_sample_witness.txt  lib.rs:608-615
so the tests can run on the GPU box, where the reference does not exist.
"""
import os
import re

SRC = "/root/reference/circom-compat/src/lib.rs"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    text = open(SRC).read()
    blobs = re.findall(r'hex_literal::hex!\(\s*"(.*?)"\s*\)', text, flags=re.S)
    assert len(blobs) == 3, len(blobs)
    r1cs_a = bytes.fromhex("".join(blobs[0].split()))
    r1cs_b = bytes.fromhex("".join(blobs[1].split()))
    wtns = bytes.fromhex("".join(blobs[2].split()))
    assert r1cs_a == r1cs_b and r1cs_a[:4] == b"r1cs" and wtns[:4] == b"wtns"
    wit = re.search(r'let witness_file = r#"(.*?)"#;', text, flags=re.S).group(1)
    open(os.path.join(HERE, "circom_sample.r1cs"), "wb").write(r1cs_a)
    open(os.path.join(HERE, "circom_sample.wtns"), "wb").write(wtns)
    open(os.path.join(HERE, "circom_sample_witness.txt"), "w").write(wit)
    print("r1cs", len(r1cs_a), "bytes; wtns", len(wtns), "bytes; witness", len(wit), "chars")


if __name__ == "__main__":
    main()
