// Stand-alone use of the C-ABI (include/fewbit_hip.h) with nothing but the HIP runtime: no torch, no python.
// What a non-PyTorch host (the reference's C++ launchers, or any FFI) would do:
//   quantize_forward  : y = gelu(x), packed 3-bit codes -> state
//   quantize_backward : gx = levels[code] * gy
// and a check against a scalar restatement on the host.  Build: make -C fewbit_amd/csrc demo
#include <fewbit_hip.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define HIP_OK(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_));          \
            return 2;                                                                  \
        }                                                                              \
    } while (0)

int main() {
    const size_t n = 1000003;                       // ragged on purpose
    const float borders[7] = {-2.41658115f, -0.71000803f, -0.32584056f, 1.06942185e-04f, 0.32605717f, 0.71024084f, 2.41447878f};
    float levels[8];
    for (int j = 0; j < 8; ++j) levels[j] = 0.125f * static_cast<float>(j);   // any table will do for the demo

    std::vector<float> x(n), gy(n), y(n), gx(n);
    uint32_t seed = 12345u;
    for (size_t i = 0; i < n; ++i) {
        seed = seed * 1664525u + 1013904223u;
        x[i] = (static_cast<float>(seed >> 8) / 16777216.0f - 0.5f) * 8.0f;
        seed = seed * 1664525u + 1013904223u;
        gy[i] = static_cast<float>(seed >> 8) / 16777216.0f - 0.5f;
    }
    const int nbits = fewbit_hip_bitwidth(8);
    const size_t nstate = fewbit_hip_state_nbytes(n, nbits);
    std::vector<uint8_t> state(nstate);

    float *dx, *dy, *dgy, *dgx, *db, *dl;
    uint8_t *dstate;
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    HIP_OK(hipMalloc(&dx, n * 4));
    HIP_OK(hipMalloc(&dy, n * 4));
    HIP_OK(hipMalloc(&dgy, n * 4));
    HIP_OK(hipMalloc(&dgx, n * 4));
    HIP_OK(hipMalloc(&db, sizeof borders));
    HIP_OK(hipMalloc(&dl, sizeof levels));
    HIP_OK(hipMalloc(&dstate, nstate));
    HIP_OK(hipMemcpyAsync(dx, x.data(), n * 4, hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(dgy, gy.data(), n * 4, hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(db, borders, sizeof borders, hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(dl, levels, sizeof levels, hipMemcpyHostToDevice, stream));

    int rc = fewbit_hip_quantize_forward(FEWBIT_GELU, FEWBIT_F32, dx, dy, dstate, n, db, 7, 0.0, 0.0, stream);
    if (rc == FEWBIT_OK) rc = fewbit_hip_quantize_backward(FEWBIT_F32, dgy, dstate, dgx, n, dl, 8, stream);
    if (rc != FEWBIT_OK) {
        std::fprintf(stderr, "fewbit_hip error %d: %s\n", rc, fewbit_hip_last_error());
        return 3;
    }
    HIP_OK(hipMemcpyAsync(y.data(), dy, n * 4, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(gx.data(), dgx, n * 4, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(state.data(), dstate, nstate, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));

    // host restatement: code = #{b < x}; LSB-first 3-bit stream; gx = levels[code] * gy
    size_t bad_code = 0, bad_gx = 0, bad_y = 0;
    for (size_t i = 0; i < n; ++i) {
        uint32_t code = 0;
        for (int j = 0; j < 7; ++j) code += !(borders[j] >= x[i]);
        const size_t bit = 3 * i;
        uint32_t got = (state[bit >> 3] | (bit / 8 + 1 < nstate ? state[(bit >> 3) + 1] << 8 : 0)) >> (bit & 7);
        bad_code += (got & 7u) != code;
        bad_gx += gx[i] != levels[code] * gy[i];
        const double ref = 0.5 * x[i] * (1.0 + std::erf(x[i] * 0.70710678118654752440));
        bad_y += std::fabs(y[i] - ref) > 4e-7 * std::fmax(1.0, std::fabs(ref));
    }
    std::printf("fewbit C-ABI v%d demo: n=%zu state=%zu bytes, mismatches: codes %zu, gradients %zu, forward %zu\n",
                fewbit_hip_abi_version(), n, nstate, bad_code, bad_gx, bad_y);
    return (bad_code || bad_gx || bad_y) ? 1 : 0;
}
