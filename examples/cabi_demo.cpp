// Stand-alone use of the C-ABI (include/fewbit_hip.h) with nothing but the HIP runtime: no torch, no python.
// What a non-PyTorch host (the reference's C++ launchers, or any FFI) would do:
//   quantize_forward  : y = gelu(x), packed 3-bit codes -> state
//   quantize_backward : gx = levels[code] * gy
//   sketch            : out = S . m with the Rademacher matrix S(seed) that exists only inside the kernel; the host rebuilds
//                       S from the ABI's own host Philox (fewbit_hip_philox4x32) and the published bit order
//   sampled_dct       : out = scale * DCT-II_ortho(m along its rows)[idx] (the reference's 'dct' estimator)
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

    // ---- random projection: out[p x f] = S[p x r] . m[r x f], S = S(seed) never in memory (fp32 m is rounded to bf16 inside) ----
    const size_t rows = 700, feats = 40, proj = 24;
    const uint64_t sk_seed = 0x0123456789abcdefull;
    std::vector<float> m(rows * feats), out(proj * feats);
    for (size_t i = 0; i < m.size(); ++i) {
        seed = seed * 1664525u + 1013904223u;
        m[i] = static_cast<float>(static_cast<int>(seed >> 24) - 128) / 64.0f;      // exactly representable in bf16
    }
    float *dm, *dout;
    void *dws = nullptr;
    const size_t ws_bytes = fewbit_hip_sketch_workspace(FEWBIT_SKETCH_RADEMACHER, FEWBIT_F32, rows, feats, proj);
    HIP_OK(hipMalloc(&dm, m.size() * 4));
    HIP_OK(hipMalloc(&dout, out.size() * 4));
    if (ws_bytes) HIP_OK(hipMalloc(&dws, ws_bytes));
    HIP_OK(hipMemcpyAsync(dm, m.data(), m.size() * 4, hipMemcpyHostToDevice, stream));
    rc = fewbit_hip_sketch(FEWBIT_SKETCH_RADEMACHER, FEWBIT_F32, dm, rows, feats, feats, proj, sk_seed, 1.0, dout, dws, ws_bytes, stream);
    if (rc != FEWBIT_OK) {
        std::fprintf(stderr, "fewbit_hip_sketch error %d: %s\n", rc, fewbit_hip_last_error());
        return 3;
    }
    HIP_OK(hipMemcpyAsync(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    // S[i][r]: word s/4 of philox(i, 2*(r/256) + h, 0, 0), bit (j odd ? 31 : 15) - (4*(s%4) + j/2); s = (r%256)/16, h = (r/8)%2, j = r%8
    const uint32_t key[2] = {static_cast<uint32_t>(sk_seed), static_cast<uint32_t>(sk_seed >> 32)};
    size_t bad_sketch = 0;
    for (size_t i = 0; i < proj; ++i) {
        std::vector<double> want(feats, 0.0);
        for (size_t r = 0; r < rows; ++r) {
            const uint32_t s = static_cast<uint32_t>((r % 256) / 16), h = static_cast<uint32_t>((r / 8) % 2), j = static_cast<uint32_t>(r % 8);
            const uint32_t ctr[4] = {static_cast<uint32_t>(i), static_cast<uint32_t>(2 * (r / 256) + h), 0u, 0u};
            uint32_t w[4];
            fewbit_hip_philox4x32(ctr, key, w);
            const double sign = ((w[s / 4] >> (((j & 1) ? 31 : 15) - (4 * (s % 4) + j / 2))) & 1u) ? -1.0 : 1.0;
            for (size_t f = 0; f < feats; ++f) want[f] += sign * m[r * feats + f];
        }
        for (size_t f = 0; f < feats; ++f) bad_sketch += std::fabs(out[i * feats + f] - want[f]) > 1e-3;
    }
    std::printf("  sketch %zu x %zu . %zu x %zu (Rademacher, seed %llx): mismatches %zu\n", proj, rows, rows, feats,
                static_cast<unsigned long long>(sk_seed), bad_sketch);
    // ---- the same product with its seed in DEVICE memory (what a launch recorded in a hipGraph uses): the seed kernel derives
    //      seed = mix(base, counter++) on the device; the host evaluates the same function and must get the same product ----
    uint64_t *dwords;                                   // [0] the counter, [1] the seed of this call
    const uint64_t base = 42, counter0 = 6;
    HIP_OK(hipMalloc(&dwords, 16));
    HIP_OK(hipMemcpyAsync(dwords, &counter0, 8, hipMemcpyHostToDevice, stream));
    std::vector<float> out_dev(out.size()), out_val(out.size());
    rc = fewbit_hip_sketch_next_seed(dwords, base, dwords + 1, stream);
    if (rc == FEWBIT_OK)
        rc = fewbit_hip_sketch_device_seed(FEWBIT_SKETCH_GAUSSIAN, FEWBIT_F32, dm, rows, feats, feats, proj, dwords + 1, 0.5, dout, dws, ws_bytes, stream);
    if (rc != FEWBIT_OK) {
        std::fprintf(stderr, "device-seed sketch error %d: %s\n", rc, fewbit_hip_last_error());
        return 3;
    }
    HIP_OK(hipMemcpyAsync(out_dev.data(), dout, out.size() * 4, hipMemcpyDeviceToHost, stream));
    uint64_t counter1 = 0;
    HIP_OK(hipMemcpyAsync(&counter1, dwords, 8, hipMemcpyDeviceToHost, stream));
    rc = fewbit_hip_sketch(FEWBIT_SKETCH_GAUSSIAN, FEWBIT_F32, dm, rows, feats, feats, proj, fewbit_hip_sketch_mix_seed(base, counter0), 0.5, dout, dws,
                           ws_bytes, stream);
    if (rc != FEWBIT_OK) return 3;
    HIP_OK(hipMemcpyAsync(out_val.data(), dout, out.size() * 4, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    size_t bad_seed = counter1 != counter0 + 1;
    for (size_t i = 0; i < out.size(); ++i) bad_seed += out_dev[i] != out_val[i];
    std::printf("  seed in device memory (counter %llu -> %llu): differences from the seed by value %zu\n",
                static_cast<unsigned long long>(counter0), static_cast<unsigned long long>(counter1), bad_seed);
    // ---- the sampled cosine transform (the reference's 'dct' estimator): out[j] = 2 * DCT-II_ortho(m along its rows)[idx[j]], against
    //      the cosine sum itself in double precision on the host ----
    const size_t drows = 256, dfeats = 6, dproj = 5;
    const int64_t picks[dproj] = {0, 1, 128, 255, 77};
    std::vector<float> dmat(drows * dfeats), dgot(dproj * dfeats);
    for (size_t i = 0; i < dmat.size(); ++i) {
        seed = seed * 1664525u + 1013904223u;
        dmat[i] = static_cast<float>(seed >> 8) / 16777216.0f - 0.5f;
    }
    const size_t dct_ws_bytes = fewbit_hip_sampled_dct_workspace(FEWBIT_F32, drows, dfeats, dproj);
    float *ddm, *ddout;
    int64_t *didx;
    void *ddws;
    HIP_OK(hipMalloc(&ddm, dmat.size() * 4));
    HIP_OK(hipMalloc(&ddout, dgot.size() * 4));
    HIP_OK(hipMalloc(&didx, sizeof picks));
    HIP_OK(hipMalloc(&ddws, dct_ws_bytes));
    HIP_OK(hipMemcpyAsync(ddm, dmat.data(), dmat.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(didx, picks, sizeof picks, hipMemcpyHostToDevice, stream));
    rc = fewbit_hip_sampled_dct(FEWBIT_F32, ddm, drows, dfeats, dfeats, didx, dproj, 2.0, ddout, ddws, dct_ws_bytes, stream);
    if (rc != FEWBIT_OK) {
        std::fprintf(stderr, "sampled_dct error %d: %s\n", rc, fewbit_hip_last_error());
        return 3;
    }
    HIP_OK(hipMemcpyAsync(dgot.data(), ddout, dgot.size() * 4, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    size_t bad_dct = 0;
    const double pi = 3.14159265358979323846;
    for (size_t j = 0; j < dproj; ++j) {
        for (size_t f = 0; f < dfeats; ++f) {
            double acc = 0.0;
            for (size_t r = 0; r < drows; ++r) acc += dmat[r * dfeats + f] * std::cos(pi * static_cast<double>(picks[j]) * (2.0 * r + 1.0) / (2.0 * drows));
            const double want = 2.0 * acc * (picks[j] == 0 ? std::sqrt(1.0 / drows) : std::sqrt(2.0 / drows));
            bad_dct += std::fabs(dgot[j * dfeats + f] - want) > 1e-5;
        }
    }
    std::printf("  sampled DCT of %zu x %zu, %zu rows picked (workspace %zu bytes): mismatches %zu\n", drows, dfeats, dproj, dct_ws_bytes, bad_dct);
    // ---- the same with the rows a function of a seed: equals the explicit call on fewbit_hip_sampled_rows(seed), bit for bit ----
    int64_t seeded_picks[dproj];
    std::vector<float> dseeded(dproj * dfeats);
    const uint64_t dct_seed = 0x5eed0f00dull;
    rc = fewbit_hip_sampled_rows(dct_seed, drows, dproj, seeded_picks);
    HIP_OK(hipMemcpyAsync(didx, seeded_picks, sizeof seeded_picks, hipMemcpyHostToDevice, stream));
    if (rc == FEWBIT_OK) rc = fewbit_hip_sampled_dct(FEWBIT_F32, ddm, drows, dfeats, dfeats, didx, dproj, 2.0, ddout, ddws, dct_ws_bytes, stream);
    HIP_OK(hipMemcpyAsync(dgot.data(), ddout, dgot.size() * 4, hipMemcpyDeviceToHost, stream));
    if (rc == FEWBIT_OK) rc = fewbit_hip_sampled_dct_seeded(FEWBIT_F32, ddm, drows, dfeats, dfeats, dct_seed, nullptr, dproj, 2.0, ddout, ddws, dct_ws_bytes, stream);
    if (rc != FEWBIT_OK) {
        std::fprintf(stderr, "sampled_dct_seeded error %d: %s\n", rc, fewbit_hip_last_error());
        return 3;
    }
    HIP_OK(hipMemcpyAsync(dseeded.data(), ddout, dseeded.size() * 4, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    size_t bad_rows = 0;
    for (size_t e = 0; e < dseeded.size(); ++e) bad_rows += std::memcmp(&dseeded[e], &dgot[e], 4) != 0;
    std::printf("  rows of seed %#llx: %lld %lld %lld %lld %lld; seeded call against the explicit one: mismatches %zu\n", static_cast<unsigned long long>(dct_seed),
                static_cast<long long>(seeded_picks[0]), static_cast<long long>(seeded_picks[1]), static_cast<long long>(seeded_picks[2]),
                static_cast<long long>(seeded_picks[3]), static_cast<long long>(seeded_picks[4]), bad_rows);
    return (bad_code || bad_gx || bad_y || bad_sketch || bad_seed || bad_dct || bad_rows) ? 1 : 0;
}
