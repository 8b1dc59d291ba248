// bh_init.hpp -- on-device initial conditions.  Replaces initializeGpu (project.cu:304-341) and its
// three kernels (initializeCurandStatesGpu / initializeMassesGpu / initializeVectorsGpu, :219-296).
//
// The reference keeps one 48-byte cuRAND XORWOW state per body and seeds it from time(NULL)
// (:323), so its runs are not reproducible.  Here the generator is counter based (Philox-4x32-10,
// Salmon et al. 2011): the random numbers of body i are a pure function of (seed, i), there is no
// state array, and a (seed, n) pair always gives the same bodies on any grid shape.
// generateRandomGpu's rule is kept (:84-97): a range with both bounds positive is sampled
// log-uniformly, any other range linearly.
#pragma once

#include "bh_prims.hpp"

namespace bh {

struct Philox {
    uint32_t c[4];
    __device__ static void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1)
    {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    // four 32-bit words for (counter, stream) under a 64-bit seed
    __device__ static void draw(uint64_t seed, uint64_t counter, uint32_t stream, uint32_t (&out)[4])
    {
        uint32_t c[4] = {(uint32_t)counter, (uint32_t)(counter >> 32), stream, 0x9E3779B9u};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) { round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
    }
};

// uniform double in [0, 1) with 53 random bits
__device__ __forceinline__ double u01(uint32_t hi, uint32_t lo)
{
    return (double)(((uint64_t)hi << 21) ^ (uint64_t)(lo >> 11)) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ double sample_range(double u, double lower, double upper)
{
    if (lower > 0 && upper > 0)                               // project.cu:86-89
        return pow(10.0, log10(lower) + u * (log10(upper) - log10(lower)));
    return u * (upper - lower) + lower;                       // project.cu:90-95
}

// kind 0: the reference's box distribution (masses in [lm, hm], positions in [lp, hp]^2,
//         velocities in [lv, hv]^2)
// kind 1: projected Plummer sphere (BASELINE config 3): scale lp (= a), truncation radius hp,
//         equal masses hm (lm ignored), zero velocities
template <typename Real2, typename Real>
__global__ __launch_bounds__(kBlock) void init_bodies_kernel(Real2 *__restrict__ pos, Real2 *__restrict__ vel,
                                                              Real *__restrict__ mass, int64_t n, uint64_t seed,
                                                              int kind, double lm, double hm, double lp,
                                                              double hp, double lv, double hv)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint32_t r[4], q[4];
    Philox::draw(seed, (uint64_t)i, 0u, r);
    Philox::draw(seed, (uint64_t)i, 1u, q);
    if (kind == 0) {
        uint32_t w[4];
        Philox::draw(seed, (uint64_t)i, 2u, w);
        mass[i] = (Real)sample_range(u01(r[0], r[1]), lm, hm);
        pos[i] = Real2{(Real)sample_range(u01(r[2], r[3]), lp, hp), (Real)sample_range(u01(q[0], q[1]), lp, hp)};
        vel[i] = Real2{(Real)sample_range(u01(q[2], q[3]), lv, hv), (Real)sample_range(u01(w[0], w[1]), lv, hv)};
    } else {
        // radius by inversion of the Plummer mass profile, rejected beyond the truncation radius by
        // drawing again from further streams (expected 1.015 draws)
        double rad = 0.0, cz = 0.0, phi = 0.0;
        for (uint32_t t = 0; t < 64; ++t) {
            uint32_t w[4];
            Philox::draw(seed, (uint64_t)i, 16u + t, w);
            double u = u01(w[0], w[1]);
            if (u < 1e-300) u = 1e-300;
            rad = lp / sqrt(pow(u, -2.0 / 3.0) - 1.0);
            cz = 2.0 * u01(w[2], w[3]) - 1.0;
            phi = 6.283185307179586 * u01(r[0] + t, r[1]);
            if (rad <= hp) break;
        }
        const double s = sqrt(1.0 - cz * cz) * rad;
        mass[i] = (Real)hm;
        pos[i] = Real2{(Real)(s * cos(phi)), (Real)(s * sin(phi))};
        vel[i] = Real2{(Real)0, (Real)0};
    }
}

}  // namespace bh
