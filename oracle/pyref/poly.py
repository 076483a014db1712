"""Radix-2 evaluation domain over Fr with ark-poly 0.4 conventions (big-int Python).

TEST INFRASTRUCTURE (oracle) — see params.py header.  Restates
`ark_poly::Radix2EvaluationDomain::{fft,ifft}_in_place` and `get_coset`
(SURVEY.md Appendix A.2), as called from the reference through
`QAP::witness_map::<_, GeneralEvaluationDomain<_>>` at
cp-groth16/src/prover.rs:123: natural order in, natural order out, point i of
the domain is w^i, ifft scales by m^-1, coset FFT = scale coeff j by g^j then
FFT, coset iFFT = iFFT then scale coeff j by g^-j.
"""


def _bitrev(i, bits):
    return int(bin(i)[2:].zfill(bits)[::-1], 2) if bits else 0


def _ntt(vals, w, r):
    n = len(vals)
    assert n & (n - 1) == 0
    bits = n.bit_length() - 1
    a = [vals[_bitrev(i, bits)] for i in range(n)]
    length = 2
    while length <= n:
        wl = pow(w, n // length, r)
        half = length // 2
        for start in range(0, n, length):
            tw = 1
            for j in range(half):
                u = a[start + j]
                v = a[start + j + half] * tw % r
                a[start + j] = (u + v) % r
                a[start + j + half] = (u - v) % r
                tw = tw * wl % r
        length *= 2
    return a


class Domain:
    def __init__(self, cp, min_size):
        """ark `D::new(min_size)`: smallest power of two >= min_size; None-equivalent
        (ValueError) when it exceeds 2^TWO_ADICITY (-> PolynomialDegreeTooLarge)."""
        self.cp = cp
        self.r = cp.r
        size = 1
        log = 0
        while size < min_size:
            size *= 2
            log += 1
        if log > cp.two_adicity:
            raise ValueError("PolynomialDegreeTooLarge")
        self.size = size
        self.log_size = log
        self.group_gen = cp.root_of_unity(log)
        self.group_gen_inv = pow(self.group_gen, -1, self.r)
        self.size_inv = pow(size, -1, self.r)

    def _pad(self, v):
        assert len(v) <= self.size
        return list(v) + [0] * (self.size - len(v))

    def fft(self, coeffs):
        return _ntt(self._pad(coeffs), self.group_gen, self.r)

    def ifft(self, evals):
        out = _ntt(self._pad(evals), self.group_gen_inv, self.r)
        return [x * self.size_inv % self.r for x in out]

    def coset_fft(self, coeffs, g):
        c = self._pad(coeffs)
        p = 1
        for j in range(self.size):
            c[j] = c[j] * p % self.r
            p = p * g % self.r
        return _ntt(c, self.group_gen, self.r)

    def coset_ifft(self, evals, g):
        c = self.ifft(evals)
        gi = pow(g, -1, self.r)
        p = 1
        for j in range(self.size):
            c[j] = c[j] * p % self.r
            p = p * gi % self.r
        return c

    def evaluate_vanishing_polynomial(self, x):
        return (pow(x, self.size, self.r) - 1) % self.r

    def evaluate_all_lagrange_coefficients(self, tau):
        """L_i(tau) for the domain points w^i (ark-poly
        `evaluate_all_lagrange_coefficients`; tau assumed outside the domain)."""
        r = self.r
        z = self.evaluate_vanishing_polynomial(tau)
        assert z != 0
        out = []
        wi = 1
        for _ in range(self.size):
            # L_i(tau) = Z(tau) * w^i / (m * (tau - w^i))
            out.append(z * wi % r * pow(self.size * (tau - wi) % r, -1, r) % r)
            wi = wi * self.group_gen % r
        return out
