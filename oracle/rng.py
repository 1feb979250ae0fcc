"""CPU restatement of the reference's parameter initialisation and shuffle -- TEST INFRASTRUCTURE.

What the reference tree itself holds (followed line by line):
  * randomNormal        tensor/tensor.nim:561-580   Box-Muller; ONE (x, y) draw yields the cosine value for the
                                                    current element and the sine value for the NEXT element of the
                                                    row-major fill; the pairing runs on across rows and blocks
  * FactorizationMachine.init  model/factorization_machine.nim:125-139   randomize(randomState); w = 0;
                                                    P = randomNormal([nOrders, nComponents, nFeatures + nAugments],
                                                    scale); intercept = 0
  * FFM init            model/field_aware_factorization_machine.nim:79-92   P = randomNormal([nFields, nFeatures, k])
  * shuffle in fit      optimizer/sgd.nim:297, adagrad.nim:167

What it calls from Nim's standard library (lib/pure/random.nim, Nim >= 1.0.6 per nimfm.nimble:10), which is NOT in
/root/reference and cannot be run here (no Nim toolchain): randomize, rand(1.0), rand(int), shuffle.  NimRand below
restates them from memory of Nim 1.0.x -- UNVERIFIED, "parity unpinned" (SURVEY.md Appendix B); everything above takes
the uniform stream as an argument, so the procedure is testable independently of the generator.
"""
import math

MASK = (1 << 64) - 1


class NimRand:
    """xoroshiro128+ as in Nim 1.0's random.nim (recalled, unverified)."""

    def __init__(self, seed=None):
        self.a0, self.a1 = 0x69B4C98CB8530805, 0xFED1DD3004688D68
        if seed is not None:
            self.randomize(seed)

    @staticmethod
    def _rotl(x, k):
        return ((x << k) | (x >> (64 - k))) & MASK

    def next(self):
        s0, s1 = self.a0, self.a1
        r = (s0 + s1) & MASK
        s1 ^= s0
        self.a0 = self._rotl(s0, 55) ^ s1 ^ ((s1 << 14) & MASK)
        self.a1 = self._rotl(s1, 36)
        return r

    def randomize(self, seed):  # initRand(seed)
        seed &= MASK
        self.a0, self.a1 = seed >> 16, seed & 0xFFFF
        self.next()

    def rand1(self):  # rand(1.0)
        import struct

        u = (0x3FF << 52) | (self.next() >> 12)
        return struct.unpack("<d", struct.pack("<Q", u))[0] - 1.0

    def rand_int(self, mx):  # rand(max: Natural): 0..max inclusive
        if mx == 0:
            return 0
        while True:
            x = self.next()
            if x <= MASK - (MASK % mx):
                return x % (mx + 1)

    def shuffle(self, x):
        for i in range(len(x) - 1, 0, -1):
            j = self.rand_int(i)
            x[i], x[j] = x[j], x[i]


def random_normal(shape, rand1, loc=0.0, scale=1.0):
    """tensor/tensor.nim:561-580, loop for loop.  rand1: callable returning the next rand(1.0).  -> nested lists."""
    out = [[[0.0] * shape[2] for _ in range(shape[1])] for _ in range(shape[0])]
    x = y = 0.0
    has_gauss = False
    for i in range(shape[0]):
        for j in range(shape[1]):
            for k in range(shape[2]):
                if not has_gauss:
                    x = rand1()
                    y = rand1()
                    z = math.sqrt(-2 * math.log(1.0 - x)) * math.cos(2 * math.pi * y)
                    has_gauss = True
                else:
                    z = math.sqrt(-2 * math.log(1.0 - x)) * math.sin(2 * math.pi * y)
                    has_gauss = False
                out[i][j][k] = loc + z * scale
    return out


def fm_init(random_state, n_orders, n_components, n_features, n_augments, scale=0.01):
    """model/factorization_machine.nim:125-139 -> (P as nested lists [o][s][j], w, intercept, the generator after)"""
    rng = NimRand(random_state)
    P = random_normal([n_orders, n_components, n_features + n_augments], rng.rand1, 0.0, scale)
    return P, [0.0] * n_features, 0.0, rng
