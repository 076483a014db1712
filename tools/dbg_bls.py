import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from hekaton_system_amd import capi
from tests import golden_util as gu
ctx = capi.Context("bls12_381", 0)
for case in gu.load("msm.json")["bls12_381"]:
    print("case", case["group"], case["n"], flush=True)
    fn = ctx.msm_g1 if case["group"] == "g1" else ctx.msm_g2
    got = fn(gu.hb(case["bases"]), gu.hb(case["scalars_mont"]))
    print(" ok" if got.tobytes().hex() == case["expect"] else " MISMATCH", flush=True)
