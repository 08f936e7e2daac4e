"""Off-circuit Poseidon random oracle of the reference, restated.  TEST INFRASTRUCTURE ONLY.

  PoseidonHash (sponge, padding, squeeze)           src/poseidon/poseidon_hash.rs:16-237  (adapted there from PSE snark-verifier)
  Spec::new(r_f, r_p)                               src/poseidon/spec.rs:14-16 -> `poseidon::Spec::new` of the THIRD-PARTY crate
      privacy-scaling-explorations/poseidon @ 807f8f55 (Cargo.toml:40-42), absent from /root/reference: its published
      algorithm is restated here -- round constants and the Cauchy MDS matrix from the Grain LFSR of the Poseidon paper
      (80-bit state: field type 1, s-box 0, field bits, t, R_F, R_P, 30 ones; 160 warm-up bits; self-shrinking output;
      constants by rejection sampling MSB-first, MDS entries 1 / (x_i + y_j) with x, y sampled without rejection), initial
      state (2^64, 0, ..).  The crate's "optimised" constants / sparse matrices are an equivalent rewriting of the plain
      permutation used here.

PINNED by the reference's own known answer `test_poseidon_hash` (src/poseidon/poseidon_hash.rs:248-266: pasta Fp, T = 3,
RATE = 2, R_F = 4, R_P = 3, absorb 0..4, squeeze 128 bits) -- tests/test_poseidon.py.
"""

PASTA_FP = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001


class _Grain:
    def __init__(self, num_bits, t, r_f, r_p):
        bits = []

        def app(n, v):
            for i in reversed(range(n)):
                bits.append((v >> i) & 1)
        app(2, 1); app(4, 0); app(12, num_bits); app(12, t); app(10, r_f); app(10, r_p); app(30, (1 << 30) - 1)
        self.s = bits
        for _ in range(160):
            self._new_bit()

    def _new_bit(self):
        s = self.s
        b = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        s.pop(0)
        s.append(b)
        return b

    def _next(self):                      # self-shrinking generator
        b = self._new_bit()
        while not b:
            self._new_bit()
            b = self._new_bit()
        return self._new_bit()

    def bits(self, n):                    # first bit = most significant
        v = 0
        for _ in range(n):
            v = (v << 1) | self._next()
        return v


def spec(p, t, r_f, r_p):
    """-> (round constants [r_f + r_p][t], mds [t][t]) as canonical ints"""
    nbits = p.bit_length()
    g = _Grain(nbits, t, r_f, r_p)

    def field():
        while True:
            v = g.bits(nbits)
            if v < p:
                return v
    rc = [[field() for _ in range(t)] for _ in range(r_f + r_p)]
    xs = [g.bits(nbits) % p for _ in range(t)]
    ys = [g.bits(nbits) % p for _ in range(t)]
    mds = [[pow((xs[i] + ys[j]) % p, p - 2, p) for j in range(t)] for i in range(t)]
    return rc, mds


def permute(state, rc, mds, p, r_f, r_p):
    t, half = len(state), r_f // 2
    for r in range(r_f + r_p):
        state = [(s + c) % p for s, c in zip(state, rc[r])]
        if r < half or r >= half + r_p:
            state = [pow(s, 5, p) for s in state]
        else:
            state[0] = pow(state[0], 5, p)
        state = [sum(mds[i][j] * state[j] for j in range(t)) % p for i in range(t)]
    return state


class PoseidonHash:                        # poseidon_hash.rs:155-237
    def __init__(self, p, t, rate, r_f, r_p):
        assert rate == t - 1
        self.p, self.t, self.rate, self.r_f, self.r_p = p, t, rate, r_f, r_p
        self.rc, self.mds = spec(p, t, r_f, r_p)
        self.buf = []

    def absorb_field(self, v):
        self.buf.append(v % self.p)
        return self

    def absorb_field_iter(self, vs):
        for v in vs:
            self.absorb_field(v)
        return self

    def absorb_point(self, xy):            # :126-141: (x, y), the identity as (0, 0)
        x, y = xy if xy is not None else (0, 0)
        return self.absorb_field(x).absorb_field(y)

    def squeeze(self, num_bits):           # output(), :190-212: the buffer is kept, the state restarts
        p, rate = self.p, self.rate
        state = [1 << 64] + [0] * (self.t - 1)                    # poseidon::State::default()
        chunks = [self.buf[i:i + rate] for i in range(0, len(self.buf), rate)]
        if len(self.buf) % rate == 0:
            chunks.append([])
        for ch in chunks:                                          # pre_round, :40-65
            for i, v in enumerate(ch):
                state[1 + i] = (state[1 + i] + v) % p
            if len(ch) < rate:
                state[1 + len(ch)] = (state[1 + len(ch)] + 1) % p
            state = permute(state, self.rc, self.mds, p, self.r_f, self.r_p)
        return state[1] & ((1 << num_bits) - 1)
