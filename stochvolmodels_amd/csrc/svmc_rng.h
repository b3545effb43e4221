// svmc_rng.h -- counter-based randoms for the gfx950 kernels.
//
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11) keyed by the seed and indexed by
// (global path id, chain-global step id, stream, call id): a lane owns a path, so it derives every
// increment it needs in registers, no generator state lives in memory and the result is independent of
// how paths are sharded over GPUs.  Replaces the reference's serial MT19937+polar draw of two
// [nb_steps, nb_path] arrays (pricers/logsv_pricer.py:1025-1026, pricers/heston_pricer.py:369-370).
// The CPU twin used by the parity tests is oracle/svmc_oracle.c (svo_draw_normals).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svmc {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&r)[4])
{
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += W0;
        k1 += W1;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

// 52 random mantissa bits -> u = m * 2^-52 + 2^-53 in (0,1); both operations are exact in fp64.
__device__ __forceinline__ double u52(uint32_t lo, uint32_t hi)
{
    const uint64_t bits = ((static_cast<uint64_t>(hi) << 32) | lo) >> 12;
    return __longlong_as_double(0x3FF0000000000000ll | static_cast<long long>(bits)) - (1.0 - 0x1.0p-53);
}

// stream 0: Box-Muller pair.  R = sqrt(-2 ln u1), (w0, w1) = R (cos 2 pi u2, sin 2 pi u2)
__device__ __forceinline__ void draw_normals(uint64_t seed, uint32_t c3, uint64_t path, uint32_t step,
                                             double &w0, double &w1)
{
    uint32_t r[4];
    philox4x32_10(static_cast<uint32_t>(path), static_cast<uint32_t>(path >> 32), step, c3,
                  static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), r);
    const double u1 = u52(r[0], r[1]);
    const double u2 = u52(r[2], r[3]);
    const double R = sqrt(-2.0 * log(u1));
    double s, c;
    sincospi(2.0 * u2, &s, &c);
    w0 = R * c;
    w1 = R * s;
}

// stream 1: one uniform in (0,1)
__device__ __forceinline__ double draw_uniform(uint64_t seed, uint32_t c3, uint64_t path, uint32_t step)
{
    uint32_t r[4];
    philox4x32_10(static_cast<uint32_t>(path), static_cast<uint32_t>(path >> 32), step, c3 | 1u,
                  static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), r);
    return u52(r[0], r[1]);
}

}  // namespace svmc
