"""LDS bank-conflict model of the NTT's lazy 9-word tile (csrc/ntt.hip: lazy::get / put): for every access pattern of k_ntt_pass_lazy /
k_ntt_last_lazy -- the two load phases (bit-reversed rows), the closing loop, the four elements of every radix-4 butterfly group -- the
number of lanes of a wavefront that meet on one of the 64 four-byte banks, per padding rule.  `e>>6` (one word per 64 elements) is what the
kernels use since r05.  usage: python tools/ntt_lds_banks.py"""
RB, COLS = 8, 8
rows = 1 << RB


def bitrev(x, b):
    return int(format(x, "0%db" % b)[::-1], 2)


def patterns():
    pats = {}
    pats["load_pass"] = [[bitrev((w * 64 + l) // COLS, RB) * COLS + (w * 64 + l) % COLS for l in range(64)] for w in range(4)]
    pats["load_last"] = [[bitrev((w * 64 + l) % rows, RB) * COLS + (w * 64 + l) // rows for l in range(64)] for w in range(4)]
    pats["close"] = [[w * 64 + l for l in range(64)] for w in range(2)]
    for K in range(4):
        s = 2 * K
        h = 1 << s
        for q in range(4):
            pats["stage%d_e%d" % (K, q)] = [[((((w * 64 + l) // COLS) >> s) << (s + 2) | (((w * 64 + l) // COLS) & (h - 1))) * COLS + q * h * COLS + (w * 64 + l) % COLS
                                             for l in range(64)] for w in range(4)]
    return pats


def cost(pad):
    tot = {}
    for name, waves in patterns().items():
        c = 0
        for lanes in waves:
            for l in range(9):
                banks = {}
                for e in lanes:
                    b = (9 * e + pad(e) + l) % 64
                    banks[b] = banks.get(b, 0) + 1
                c += max(banks.values())
        tot[name] = c / (len(waves) * 9)
    return tot


if __name__ == "__main__":
    for n, f in {"none": lambda e: 0, "e>>3": lambda e: e >> 3, "e>>4": lambda e: e >> 4, "e>>5": lambda e: e >> 5, "e>>6": lambda e: e >> 6,
                 "e>>7": lambda e: e >> 7}.items():
        t = cost(f)
        st = [v for k, v in t.items() if k.startswith("stage")]
        print(n.ljust(6), "load_pass %.1f  load_last %.1f  close %.1f  butterfly groups avg %.2f max %.1f   (lanes per bank, 1.0 = conflict-free)"
              % (t["load_pass"], t["load_last"], t["close"], sum(st) / len(st), max(st)))
