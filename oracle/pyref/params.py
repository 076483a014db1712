"""Curve / field constants for the two instantiations the build ships.

TEST INFRASTRUCTURE (oracle): only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package.

PARITY UNPINNED: the reference (zhaowenlan1779/hekaton-system) holds no golden
vectors for the CP-Groth16 hot path and cannot be built here
(no Rust toolchain, arkworks crates un-vendored).  These
constants are the public BN254 / BLS12-381 parameters that ark-bn254 0.4 and
ark-bls12-381 0.4 instantiate (reference: mpi-snark/src/data_structures.rs:9
`use ark_bn254::{Bn254 as E, Fr}`; cp-groth16/src/prover.rs:15 generic over
`E: Pairing`).  Everything derived below is re-validated by self-checks
(`python -m oracle.pyref.params`).
"""

from dataclasses import dataclass


@dataclass(frozen=True)
class CurveParams:
    name: str
    cid: int                 # hk_curve enum value in include/hekaton.h
    r: int                   # scalar field modulus (Fr)
    q: int                   # base field modulus (Fq)
    fr_limbs64: int
    fq_limbs64: int
    two_adicity: int         # of r-1
    fr_generator: int        # multiplicative generator of Fr (ark `F::GENERATOR`)
    g1_b: int                # y^2 = x^3 + b
    g1_gen: tuple
    g2_b: tuple              # twist coefficient b' in Fq2 = Fq[u]/(u^2+1)
    g2_gen: tuple            # ((x.c0,x.c1),(y.c0,y.c1))

    @property
    def fr_bits(self):
        return self.r.bit_length()

    @property
    def fr_R(self):          # Montgomery radix used by ark MontBackend<_, N>: 2^(64 N)
        return 1 << (64 * self.fr_limbs64)

    @property
    def fq_R(self):
        return 1 << (64 * self.fq_limbs64)

    @property
    def two_adic_root(self):
        """ark `TWO_ADIC_ROOT_OF_UNITY` = GENERATOR^((r-1)/2^s)."""
        return pow(self.fr_generator, (self.r - 1) >> self.two_adicity, self.r)

    def root_of_unity(self, log_m):
        """ark-poly Radix2EvaluationDomain::group_gen for size 2^log_m
        (SURVEY.md Appendix A.2)."""
        assert log_m <= self.two_adicity
        return pow(self.two_adic_root, 1 << (self.two_adicity - log_m), self.r)


BN254 = CurveParams(
    name="bn254", cid=0,
    r=21888242871839275222246405745257275088548364400416034343698204186575808495617,
    q=21888242871839275222246405745257275088696311157297823662689037894645226208583,
    fr_limbs64=4, fq_limbs64=4, two_adicity=28, fr_generator=5,
    g1_b=3, g1_gen=(1, 2),
    g2_b=(19485874751759354771024239261021720505790618469301721065564631296452457478373,
          266929791119991161246907387137283842545076965332900288569378510910307636690),
    g2_gen=((10857046999023057135944570762232829481370756359578518086990519993285655852781,
             11559732032986387107991004021392285783925812861821192530917403151452391805634),
            (8495653923123431417604973247489272438418190587263600148770280649306958101930,
             4082367875863433681332203403145435568316851327593401208105741076214120093531)),
)

BLS12_381 = CurveParams(
    name="bls12_381", cid=1,
    r=52435875175126190479447740508185965837690552500527637822603658699938581184513,
    q=0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
    fr_limbs64=4, fq_limbs64=6, two_adicity=32, fr_generator=7,
    g1_b=4,
    g1_gen=(0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
            0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1),
    g2_b=(4, 4),
    g2_gen=((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
             0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
            (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
             0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be)),
)

CURVES = {"bn254": BN254, "bls12_381": BLS12_381}


def _is_probable_prime(n, rounds=16):
    import random
    if n < 4:
        return n in (2, 3)
    if n % 2 == 0:
        return False
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    rnd = random.Random(1234)
    for _ in range(rounds):
        a = rnd.randrange(2, n - 1)
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def self_check():
    from . import curve
    for cp in CURVES.values():
        assert _is_probable_prime(cp.r) and _is_probable_prime(cp.q)
        assert (cp.r - 1) % (1 << cp.two_adicity) == 0
        assert ((cp.r - 1) >> cp.two_adicity) % 2 == 1
        w = cp.two_adic_root
        assert pow(w, 1 << cp.two_adicity, cp.r) == 1
        assert pow(w, 1 << (cp.two_adicity - 1), cp.r) == cp.r - 1
        g1 = curve.G1(cp)
        g2 = curve.G2(cp)
        assert g1.on_curve(cp.g1_gen) and g2.on_curve(cp.g2_gen)
        assert g1.mul(cp.g1_gen, cp.r) is None
        assert g2.mul(cp.g2_gen, cp.r) is None
    # EIP-196 known answer: 2*G1 on BN254 (alt_bn128)
    g1 = curve.G1(BN254)
    assert g1.dbl((1, 2)) == (
        1368015179489954701390400359078579693043519447331113978918064868415326638035,
        9918110051302171585080402603319702774565515993150576347155970296011118125764)
    return True


if __name__ == "__main__":
    print("params self-check:", self_check())
