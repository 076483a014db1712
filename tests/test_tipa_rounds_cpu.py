"""CPU: the round loop of the TIPP prover (tipa.gipa_rounds) in the EXPONENT domain - G1 / G2 vectors as their discrete
logarithms, a pairing value as the product of the logarithms, a GT product as a sum - so that the paired form (two rounds
per pass through the pairing pipeline: sixty quarter-by-quarter inner products, round k + 1's messages by bilinearity)
can be checked against the round-by-round form without a GPU: same messages, same challenges, same folded vectors."""
import random
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from hekaton_system_amd import tipa
from hekaton_system_amd.cp_groth16 import CURVE_PARAMS, FrCodec

R_MOD = CURVE_PARAMS["bn254"]["r"]


class ExpField:
    """GT in the exponent: an element is its logarithm."""
    one = 0

    @staticmethod
    def encode(x):
        return int(x).to_bytes(32, "little")

    @staticmethod
    def decode(b):
        return int.from_bytes(bytes(b), "little")

    @staticmethod
    def mul(a, b):
        return (a + b) % R_MOD


class View:
    def __init__(self, store, start, count):
        self.store, self.start, self.count = store, start, count

    def get(self):
        return self.store[self.start:self.start + self.count]


class ExpCtx:
    gt_bytes = 32

    def __init__(self):
        self.fc = FrCodec("bn254")
        self.pairing_calls = 0

    def pairing_pairs(self, lhs, rhs, pairs, n):
        self.pairing_calls += 1
        assert all(v.count == n for v in list(lhs) + list(rhs))
        rows = [sum(x * y for x, y in zip(lhs[a].get(), rhs[b].get())) % R_MOD for a, b in pairs]
        return np.frombuffer(b"".join(ExpField.encode(v) for v in rows), np.uint8).reshape(len(pairs), 32).copy()

    def gt_pow_prod(self, gts, scalars, group_len, in_gt=True):
        g = np.asarray(gts, np.uint8).reshape(-1, 32)
        ks = self.fc.dec(scalars)
        assert len(ks) == len(g) and len(g) % group_len == 0
        out = []
        for k0 in range(0, len(g), group_len):
            out.append(sum(ExpField.decode(g[k0 + j]) * ks[k0 + j] for j in range(group_len)) % R_MOD)
        return np.frombuffer(b"".join(ExpField.encode(v) for v in out), np.uint8).reshape(len(out), 32).copy()

    def points_fold_many(self, group, los, his, c, n, outs):
        for lo, hi, out in zip(los, his, outs):
            assert lo.count == hi.count == out.count == n
            out.store[out.start:out.start + n] = [(x + c * y) % R_MOD for x, y in zip(lo.get(), hi.get())]
        return outs


def _run(n, seed, paired):
    rnd = random.Random(seed)
    stores = [[rnd.randrange(R_MOD) for _ in range(n)] + [0] * n for _ in range(6)]
    win = lambda k, start, count: View(stores[k], start, count)
    ctx = ExpCtx()
    tr = tipa.Transcript(R_MOD)
    tr.absorb(b"instance", n.to_bytes(8, "little"))
    rounds, challenges, times = [], [], []
    with ThreadPoolExecutor(max_workers=4) as pool:
        pos = tipa.gipa_rounds(ctx, ExpField, ctx.fc, R_MOD, win, n, tr, pool.submit, rounds, challenges, times, paired)
    return rounds, challenges, [s[pos] for s in stores], ctx.pairing_calls, stores


@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64])
def test_paired_rounds_send_the_messages_of_the_round_by_round_form(n):
    r1, c1, fin1, calls1, st1 = _run(n, 5 + n, False)
    r2, c2, fin2, calls2, st2 = _run(n, 5 + n, True)
    logn = n.bit_length() - 1
    assert len(r1) == len(r2) == len(c1) == len(c2) == logn
    assert c1 == c2 and r1 == r2 and fin1 == fin2
    assert st1 == st2                                        # every folded vector along the way, too
    assert calls1 == logn and calls2 == logn // 2 + logn % 2
    # the round-by-round form is itself consistent: T . TL^c . TR^(1/c) is the inner product of the folded vectors
    a, v1, w1, b = (st1[k] for k in (0, 2, 4, 1))
    T = (sum(x * y for x, y in zip(a[:n], v1[:n])) + sum(x * y for x, y in zip(w1[:n], b[:n]))) % R_MOD
    pos, m = 0, n
    for rd, c in zip(r1, c1):
        T = (T + rd["TL"] * c + rd["TR"] * pow(c, -1, R_MOD)) % R_MOD
        pos, m = pos + m, m // 2
        assert T == (sum(x * y for x, y in zip(a[pos:pos + m], v1[pos:pos + m])) +
                     sum(x * y for x, y in zip(w1[pos:pos + m], b[pos:pos + m]))) % R_MOD


def test_a_round_pair_fits_the_batched_pairing_call():
    """One pass takes 5 vector pairs x 12 quarter combinations = 60 pairs out of 12 + 12 quarter vectors: inside the pair
    list hk_pairing_pairs accepts (PAIR_LIST_MAX) and the row table of its gather launch (GatherRows::MAX)."""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hekaton_system_amd", "csrc")
    drv = open(os.path.join(root, "msm_driver.cuh")).read()
    impl = open(os.path.join(root, "prove_impl.cuh")).read()
    pair_max = int(re.search(r"PAIR_LIST_MAX\s*=\s*(\d+)", drv).group(1))
    rows_max = int(re.search(r"struct GatherRows \{\s*enum \{ MAX = (\d+) \}", impl).group(1))
    assert len(tipa._NAME_PAIRS) * len(tipa._QUARTERS) == 60 <= pair_max
    assert 4 * (len(tipa._G1_VECS) + len(tipa._G2_VECS)) == 24 <= rows_max
    assert len(set(tipa._QUARTERS)) == 12 and all(0 <= i < 4 and 0 <= j < 4 for i, j in tipa._QUARTERS)
