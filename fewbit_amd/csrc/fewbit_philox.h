// fewbit_philox.h -- Philox4x32 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11), the counter-based
// generator behind everything this library draws from a 64-bit seed: the sign words and Gaussian stream seeds of the dense sketches
// (fewbit_sketch.hip; counter word 3 = 0 / 2) and the sampled rows of the cosine transform (fewbit_dct.hip; counter word 3 = 3).
// Host and device evaluate the same function (fewbit_hip_philox4x32 exposes it to the tests).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace fewbit_hip {
namespace sketch {

constexpr int kPhiloxRounds = 10;

struct Key { uint32_t k0, k1; };

template <int ROUNDS = kPhiloxRounds>
__host__ __device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, Key key, uint32_t (&out)[4]) {
    uint32_t k0 = key.k0, k1 = key.k1;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0, p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
        const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0, n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1;
        c1 = static_cast<uint32_t>(p1);
        c3 = static_cast<uint32_t>(p0);
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

}  // namespace sketch
}  // namespace fewbit_hip
