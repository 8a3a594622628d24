"""TEST INFRASTRUCTURE ONLY (see oracle/pasta.py): the Kimchi Poseidon sponge and the Fq-sponge of the reference,
restated so that the oracle can replay SRS::open end to end and be pinned on the reference's opening-proof
known-answer bytes (poly-commitment/tests/commitment.rs:388-440).

  ArithmeticSponge   poseidon/src/poseidon.rs:60-175, permutation.rs (full_round: sbox x^7, MDS, + round constants;
                     PlonkSpongeConstantsKimchi constants.rs:29-41: width 3, rate 2, 55 full rounds, no initial ARK)
  DefaultFqSponge    poseidon/src/sponge.rs:228-412 (absorb_g / absorb_fr / challenge / challenge_fq)
Parameters: tests/golden/poseidon_kimchi_params.json (made by tests/golden/make_poseidon_params.py)."""
import json
import os
from typing import List, Optional, Tuple

from . import pasta as P

_PARAMS = None


def params(name: str):
    global _PARAMS
    if _PARAMS is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "poseidon_kimchi_params.json")
        raw = json.load(open(path))
        _PARAMS = {k: {"mds": [[int(x) for x in row] for row in v["mds"]],
                       "rc": [[int(x) for x in row] for row in v["round_constants"]]} for k, v in raw.items() if k in ("fp", "fq")}
    return _PARAMS[name]


class ArithmeticSponge:
    RATE = 2

    def __init__(self, F: P.Field):
        self.F = F
        self.par = params("fp" if F is P.Fp else "fq")
        self.state = [0, 0, 0]
        self.mode, self.n = "absorbed", 0

    def permute(self):                                   # poseidon_block_cipher with PERM_HALF_ROUNDS_FULL = 0, no initial ARK
        p = self.F.p
        mds, rc = self.par["mds"], self.par["rc"]
        s = self.state
        for r in range(55):
            s = [pow(x, 7, p) for x in s]
            s = [(mds[i][0] * s[0] + mds[i][1] * s[1] + mds[i][2] * s[2] + rc[r][i]) % p for i in range(3)]
        self.state = s

    def absorb(self, xs):
        for x in xs:
            if self.mode == "absorbed":
                if self.n == self.RATE:
                    self.permute()
                    self.n = 1
                    self.state[0] = (self.state[0] + x) % self.F.p
                else:
                    self.state[self.n] = (self.state[self.n] + x) % self.F.p
                    self.n += 1
            else:
                self.state[0] = (self.state[0] + x) % self.F.p
                self.mode, self.n = "absorbed", 1

    def squeeze(self) -> int:
        if self.mode == "squeezed" and self.n < self.RATE:
            self.n += 1
            return self.state[self.n - 1]
        self.permute()
        self.mode, self.n = "squeezed", 1
        return self.state[0]


class DefaultFqSponge:
    """Sponge over the curve's base field; scalars are those of the curve's scalar field."""

    def __init__(self, curve: P.Curve):
        self.curve = curve
        self.sponge = ArithmeticSponge(curve.base)
        self.last_squeezed: List[int] = []

    def clone(self):
        c = DefaultFqSponge(self.curve)
        c.sponge.state = list(self.sponge.state); c.sponge.mode = self.sponge.mode; c.sponge.n = self.sponge.n
        c.last_squeezed = list(self.last_squeezed)
        return c

    def absorb_g(self, pts):
        self.last_squeezed = []
        for pt in pts:
            if pt is None:
                self.sponge.absorb([0]); self.sponge.absorb([0])
            else:
                self.sponge.absorb([pt[0]]); self.sponge.absorb([pt[1]])

    def absorb_fq(self, xs):
        self.last_squeezed = []
        for x in xs:
            self.sponge.absorb([x])

    def absorb_fr(self, xs):
        self.last_squeezed = []
        for x in xs:
            if self.curve.scalar.p < self.curve.base.p:
                self.sponge.absorb([x])
            else:
                self.sponge.absorb([x >> 1]); self.sponge.absorb([x & 1])

    def _squeeze_limbs(self, k: int) -> List[int]:
        while len(self.last_squeezed) < k:
            x = self.sponge.squeeze()
            self.last_squeezed += [x & (2**64 - 1), (x >> 64) & (2**64 - 1)]      # HIGH_ENTROPY_LIMBS = 2
        out, self.last_squeezed = self.last_squeezed[:k], self.last_squeezed[k:]
        return out

    def challenge(self) -> int:                         # 128 bits, as a scalar-field element
        lo, hi = self._squeeze_limbs(2)
        return lo | (hi << 64)

    def challenge_fq(self) -> int:
        self.last_squeezed = []
        return self.sponge.squeeze()
