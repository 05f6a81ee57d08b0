"""Second, independent restatement of the integer coder in pure Python big-int arithmetic (small cases only).
Written from the reference source text (src/main.rs:261-318, 353-449, 500-548) to cross-check the C oracle:
the reference itself has no KATs for these functions, so the two restatements pin each other."""
M64 = (1 << 64) - 1
M128 = (1 << 128) - 1
FULL, HALF, QUARTER = (1 << 32) - 1, 1 << 31, 1 << 30


class PyEncoder:
    def __init__(self):
        self.low, self.high, self.carry_run = 0, FULL, 0
        self.bits = []

    def _put(self, bit):
        self.bits.append(bit & 1)
        while self.carry_run > 0:
            self.bits.append((~bit) & 1)
            self.carry_run -= 1

    def encode_counts(self, c_lo, c_hi, total):
        rng = (self.high - self.low + 1) & M64
        new_low = (self.low + (rng * c_lo) // total) & M128
        new_high = (self.low + (rng * c_hi) // total - 1) & M128
        self.low, self.high = new_low & FULL, new_high & FULL
        while True:
            if self.high < HALF:
                self._put(0)
            elif self.low >= HALF:
                self._put(1); self.low -= HALF; self.high -= HALF
            elif self.low >= QUARTER and self.high < 3 * QUARTER:
                self.carry_run += 1; self.low -= QUARTER; self.high -= QUARTER
            else:
                break
            self.low = (self.low << 1) & FULL
            self.high = ((self.high << 1) & FULL) | 1

    def finish(self):
        self.carry_run += 1
        self._put(0 if self.low < QUARTER else 1)
        while len(self.bits) % 8:
            self.bits.append(0)
        out = bytearray()
        for i in range(0, len(self.bits), 8):
            b = 0
            for x in self.bits[i:i + 8]:
                b = (b << 1) | x
            out.append(b)
        return bytes(out)


class PyDecoder:
    def __init__(self, data):
        self.data, self.pos = data, 0
        self.low, self.high, self.code = 0, FULL, 0
        for _ in range(32):
            self.code = (self.code << 1) | self._bit()

    def _bit(self):
        if self.pos >= 8 * len(self.data):
            return 1
        b = (self.data[self.pos >> 3] >> (7 - (self.pos & 7))) & 1
        self.pos += 1
        return b

    def decode_symbol_counts(self, cdf, total):
        rng = (self.high - self.low + 1) & M64
        value = ((((self.code - self.low + 1) & M64) * total - 1) & M128) // rng
        value &= 0xFFFFFFFF
        lo, hi = 0, len(cdf) - 1
        while lo + 1 < hi:
            mid = (lo + hi) // 2
            if cdf[mid] <= value:
                lo = mid
            else:
                hi = mid
        s = lo
        new_low = (self.low + (rng * cdf[s]) // total) & M128
        new_high = (self.low + (rng * cdf[s + 1]) // total - 1) & M128
        self.low, self.high = new_low & M64, new_high & M64
        while True:
            if self.high < HALF:
                pass
            elif self.low >= HALF:
                self.low -= HALF; self.high -= HALF; self.code -= HALF
            elif self.low >= QUARTER and self.high < 3 * QUARTER:
                self.low -= QUARTER; self.high -= QUARTER; self.code -= QUARTER
            else:
                break
            self.low = (self.low << 1) & FULL
            self.high = ((self.high << 1) & FULL) | 1
            self.code = ((self.code << 1) & FULL) | self._bit()
        return s
