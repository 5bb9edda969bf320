"""Julia's `Random.seed!(k); randn(m, n)` restated -- TEST INFRASTRUCTURE ONLY (the rule of oracle/fos_oracle.py's header:
only tests/, smoke() and the cpu_baseline leg may import anything under oracle/).

Why: the reference's whole-solve known answers (test/testDRandGAPA.jl:2-18, README.md:21-26) are literal optima of a
problem whose data is `Random.seed!(2); A = randn(40, 50); b = randn(40, 1)`.  Julia is not in this image, but its
generator is a published algorithm: `MersenneTwister` is dSFMT-19937 (Saito & Matsumoto, dSFMT 2.2.3, seeded with
`dsfmt_init_by_array(UInt32[seed])`) read through a 1002-double cache, and `randn` is the 256-layer ziggurat of
Julia's stdlib Random (normal.jl, tables of randmtzig).  Restating both gives the reference's inputs, and the two optima
the test file holds pin the restatement itself: the pre-1.5 stream reproduces 12.38418747141913 to 3e-15 (relative) and
the >= 1.5 stream (array `randn!` filling the array straight from the generator first) reproduces 10.945929126466417
within the test's own `≈` (that literal was pasted from a solve at eps = 1e-8: it is 9.5e-9 above the exact optimum).
A single wrong draw moves the optimum in the second digit.

Two consumers of the stream exist in Julia's history and both are here:
  * randn_scalar_fill  -- Julia < 1.5: `for i in eachindex(A) A[i] = randn(rng)`.
  * randn_array_fill   -- Julia 1.5 / 1.6 (still MersenneTwister as the default generator): `randn!(rng, A::Array{Float64})`
    first fills A with raw doubles in [1, 2) (`rand!(rng, A, CloseOpen12())`: `dsfmt_fill_array_close1_open2!` straight
    from the state for all but the last 2 (3 if odd) entries when that is >= 382 values, the scalar cache path
    otherwise), then maps each entry through the ziggurat, drawing the rare rejections' extra numbers from the cache.
"""
from __future__ import annotations

import math
import struct

import numpy as np

_M64 = (1 << 64) - 1
_M32 = 0xFFFFFFFF
# dSFMT-19937 parameters (dSFMT-params19937.h)
_N = 191
_POS1 = 117
_SL1 = 19
_SR = 12
_MSK1 = 0x000ffafffffffb3f
_MSK2 = 0x000ffdfffc90fffd
_FIX1 = 0x90014964b32f4329
_FIX2 = 0x3b8d12ac548a7c7a
_PCV1 = 0x3d84e1ac0dc82880
_PCV2 = 0x0000000000000001
_LOW_MASK = 0x000FFFFFFFFFFFFF
_HIGH_CONST = 0x3FF0000000000000
_CACHE = 1002                        # MT_CACHE_F: doubles Julia's MersenneTwister keeps ahead
_MIN_ARRAY = 382                     # dsfmt_get_min_array_size()


def _recursion(a, b, lung):
    """dSFMT's do_recursion on one 128-bit word (two u64 halves)."""
    t0, t1 = a
    l0, l1 = lung
    n0 = ((t0 << _SL1) & _M64) ^ (l1 >> 32) ^ ((l1 << 32) & _M64) ^ b[0]
    n1 = ((t1 << _SL1) & _M64) ^ (l0 >> 32) ^ ((l0 << 32) & _M64) ^ b[1]
    return ((n0 >> _SR) ^ (n0 & _MSK1) ^ t0, (n1 >> _SR) ^ (n1 & _MSK2) ^ t1), (n0, n1)


class JuliaMersenneTwister:
    """`MersenneTwister(seed)` for 0 <= seed < 2^32 (Julia 0.7 .. 1.6 seeding: the key is the seed's UInt32 digits)."""

    def __init__(self, seed: int):
        if not 0 <= seed <= _M32:
            raise ValueError("one UInt32 word of seed only")
        self._init_by_array([seed])
        self._vals: list[int] = []
        self._idx = _CACHE               # cache empty after seeding

    def _init_by_array(self, key):
        size = (_N + 1) * 4              # the state seen as 32-bit words (little endian)
        lag = 11
        mid = (size - lag) // 2
        p = [0x8b8b8b8b] * size
        klen = len(key)
        count = max(klen + 1, size)

        def f1(x):
            return ((x ^ (x >> 27)) * 1664525) & _M32

        def f2(x):
            return ((x ^ (x >> 27)) * 1566083941) & _M32

        r = f1(p[0] ^ p[mid % size] ^ p[(size - 1) % size])
        p[mid % size] = (p[mid % size] + r) & _M32
        r = (r + klen) & _M32
        p[(mid + lag) % size] = (p[(mid + lag) % size] + r) & _M32
        p[0] = r
        count -= 1
        i = 1
        for j in range(count):
            r = f1(p[i] ^ p[(i + mid) % size] ^ p[(i + size - 1) % size])
            p[(i + mid) % size] = (p[(i + mid) % size] + r) & _M32
            r = (r + (key[j] if j < klen else 0) + i) & _M32
            p[(i + mid + lag) % size] = (p[(i + mid + lag) % size] + r) & _M32
            p[i] = r
            i = (i + 1) % size
        for _ in range(size):
            r = f2((p[i] + p[(i + mid) % size] + p[(i + size - 1) % size]) & _M32)
            p[(i + mid) % size] ^= r
            r = (r - i) & _M32
            p[(i + mid + lag) % size] ^= r
            p[i] = r
            i = (i + 1) % size
        u64 = [p[2 * k] | (p[2 * k + 1] << 32) for k in range(size // 2)]
        for k in range(2 * _N):          # initial_mask: every state word is a double in [1, 2)
            u64[k] = (u64[k] & _LOW_MASK) | _HIGH_CONST
        inner = ((u64[2 * _N] ^ _FIX1) & _PCV1) ^ ((u64[2 * _N + 1] ^ _FIX2) & _PCV2)   # period_certification
        sh = 32
        while sh:
            inner ^= inner >> sh
            sh >>= 1
        if not inner & 1:
            u64[2 * _N + 1] ^= 1
        self._status = [(u64[2 * k], u64[2 * k + 1]) for k in range(_N + 1)]

    def fill_raw(self, ndoubles: int) -> list[int]:
        """dsfmt_fill_array_close1_open2: `ndoubles` (even, >= 382) values straight from the state, as bit patterns."""
        assert ndoubles % 2 == 0 and ndoubles >= _MIN_ARRAY
        size = ndoubles // 2
        st = self._status
        arr = [None] * size
        lung = st[_N]
        i = 0
        while i < _N - _POS1:
            arr[i], lung = _recursion(st[i], st[i + _POS1], lung)
            i += 1
        while i < _N:
            arr[i], lung = _recursion(st[i], arr[i + _POS1 - _N], lung)
            i += 1
        while i < size - _N:
            arr[i], lung = _recursion(arr[i - _N], arr[i + _POS1 - _N], lung)
            i += 1
        j = 0
        while j < 2 * _N - size:
            st[j] = arr[j + size - _N]
            j += 1
        while i < size:
            arr[i], lung = _recursion(arr[i - _N], arr[i + _POS1 - _N], lung)
            st[j] = arr[i]
            i += 1
            j += 1
        st[_N] = lung
        return [w for pair in arr for w in pair]

    def raw(self) -> int:
        """Bit pattern of the next cached double in [1, 2) (`rand_inbounds(r, Close1Open2())` after `reserve_1`)."""
        if self._idx >= _CACHE:
            self._vals = self.fill_raw(_CACHE)
            self._idx = 0
        v = self._vals[self._idx]
        self._idx += 1
        return v

    def rand(self) -> float:
        """`rand(rng)`: CloseOpen01 = the [1, 2) double minus one."""
        return struct.unpack("<d", struct.pack("<Q", self.raw()))[0] - 1.0


# ---- ziggurat tables (randmtzig's create_ziggurat_tables; Julia's normal.jl holds the same numbers as literals)
_NOR_R = 3.6541528853610088
_NOR_INV_R = 0.27366123732975828
_NMANT = 2251799813685248.0          # 2^51
# area of one layer: r f(r) + the tail's mass (randmtzig prints it to 12 digits; the tables were made from the full value)
_AREA = _NOR_R * math.exp(-0.5 * _NOR_R * _NOR_R) + math.sqrt(math.pi / 2) * math.erfc(_NOR_R / math.sqrt(2))


def _ziggurat_tables():
    ki = [0] * 256
    wi = [0.0] * 256
    fi = [0.0] * 256
    x1 = _NOR_R
    wi[255] = x1 / _NMANT
    fi[255] = math.exp(-0.5 * x1 * x1)
    ki[0] = int(x1 * fi[255] / _AREA * _NMANT)
    wi[0] = _AREA / fi[255] / _NMANT
    fi[0] = 1.0
    for i in range(254, 0, -1):
        x = math.sqrt(-2.0 * math.log(_AREA / x1 + fi[i + 1]))
        ki[i + 1] = int(x / x1 * _NMANT)
        wi[i] = x / _NMANT
        fi[i] = math.exp(-0.5 * x * x)
        x1 = x
    ki[1] = 0
    return ki, wi, fi


_KI, _WI, _FI = _ziggurat_tables()


def _randn_from_bits(rng: JuliaMersenneTwister, bits: int) -> float:
    """normal.jl `_randn(rng, r::UInt64)` with `randn_unlikely` (0-based tables here)."""
    while True:
        r = bits & _LOW_MASK
        rabs = r >> 1                    # one bit for the sign
        idx = rabs & 0xFF
        x = (-rabs if r & 1 else rabs) * _WI[idx]
        if rabs < _KI[idx]:
            return x                     # 99.3 % of the draws
        if idx == 0:                     # the tail
            while True:
                xx = -_NOR_INV_R * math.log(rng.rand())
                yy = -math.log(rng.rand())
                if yy + yy > xx * xx:
                    return -_NOR_R - xx if (rabs >> 8) & 1 else _NOR_R + xx
        if (_FI[idx - 1] - _FI[idx]) * rng.rand() + _FI[idx] < math.exp(-0.5 * x * x):
            return x                     # the wedge
        bits = rng.raw()                 # `return randn(rng)`


def randn_scalar_fill(rng: JuliaMersenneTwister, *dims: int) -> np.ndarray:
    """`randn(rng, dims...)` as Julia < 1.5 filled it: one scalar `randn(rng)` per entry, column-major."""
    n = int(np.prod(dims))
    return np.array([_randn_from_bits(rng, rng.raw()) for _ in range(n)]).reshape(dims, order="F")


def randn_array_fill(rng: JuliaMersenneTwister, *dims: int) -> np.ndarray:
    """`randn(rng, dims...)` as Julia 1.5 / 1.6 filled it (`randn!(::MersenneTwister, ::Array{Float64})`)."""
    n = int(np.prod(dims))
    if n < 13:
        return randn_scalar_fill(rng, *dims)
    n2 = (n - 2) // 2 * 2
    if n2 < _MIN_ARRAY:
        bits = [rng.raw() for _ in range(n)]
    else:
        bits = rng.fill_raw(n2) + [rng.raw() for _ in range(n - n2)]
    return np.array([_randn_from_bits(rng, b) for b in bits]).reshape(dims, order="F")


def readme_nnls_data(julia: str):
    """test/testDRandGAPA.jl:2-5: `Random.seed!(2); A = randn(40, 50); b = randn(40, 1)` and the optimum the file holds
    for that Julia generation (:11-17)."""
    rng = JuliaMersenneTwister(2)
    fill = {"pre1.5": randn_scalar_fill, "1.5": randn_array_fill}[julia]
    A = fill(rng, 40, 50)
    b = fill(rng, 40, 1)
    opt = {"pre1.5": 12.38418747141913, "1.5": 10.945929126466417}[julia]
    return A, b[:, 0], opt


def feasibility_test_data():
    """test/testfeasibility.jl:2-7: `Random.seed!(2); xsol1 = randn(100); A = randn(50, 100); b = A*xsol1`, Julia >= 1.5 draw
    (the file has no version switch and its assertions -- DR reaches the intersection to 1e-12, AP / GAP / FISTA end
    :Indeterminate -- hold for that draw only: the pre-1.5 draw gives an EMPTY intersection, distance 6.18)."""
    rng = JuliaMersenneTwister(2)
    xsol = randn_array_fill(rng, 100)
    A = randn_array_fill(rng, 50, 100)
    return xsol, A
