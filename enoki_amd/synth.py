"""Synthetic inputs generated ON THE DEVICE with the array ops themselves: a counter-based
integer hash that host (tests/conftest.py:hash_u32) and device evaluate identically, so a shard can create its
index range [begin, begin+n) without any host data."""
import enoki_amd.hip as ek


def hash_u32(begin, n, seed):
    """h(i, seed) for i in [begin, begin+n):  v = i + seed*0x9E3779B9; v ^= v>>16; v *= 0x7feb352d; v ^= v>>15;
    v *= 0x846ca68b; v ^= v>>16   (all arithmetic modulo 2^32)"""
    U = ek.UInt32
    v = U.arange(n) + U((begin + seed * 0x9E3779B9) & 0xFFFFFFFF)
    v = v ^ (v >> U(16))
    v = v * U(0x7FEB352D)
    v = v ^ (v >> U(15))
    v = v * U(0x846CA68B)
    v = v ^ (v >> U(16))
    return v


def uniform_pm1(begin, n, seed):
    """f32 uniform in [-1, 1): 2u - 1 with u = (h >> 8) * 2^-24 (exact in f32)"""
    h = hash_u32(begin, n, seed)
    u = ek.Float32(h >> ek.UInt32(8)) * ek.Float32(2.0 ** -24)
    return ek.fmadd(u, ek.Float32(2.0), ek.Float32(-1.0))


def index_mod(begin, n, seed, k):
    """uniform indices in [0, k)"""
    return hash_u32(begin, n, seed) % ek.UInt32(k)
