// What can a streaming kernel reach on MI355X at the SIZE of the headline config (32 MiB in, 32 MiB out), cache-cold
// and cache-warm?  Copy / read-only / write-only kernels in the launch shapes this repo uses or could use.
//   hipcc --offload-arch=gfx950 -O3 scratch/stream_bench.hip -o scratch/stream_bench && scratch/stream_bench [MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4 *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT> __device__ __forceinline__ void st(u32x4 *p, u32x4 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// persistent waves, round-robin tiles of U x 1 KiB per wave, two register buffers (the structure of fewbit's kernels)
template <int U, bool NTL, bool NTS, int OP>   // OP 0 copy, 1 read-only, 2 write-only
__global__ __launch_bounds__(256) void persistent(const u32x4 *src, u32x4 *dst, size_t nvec, unsigned *sink) {
    const int lane = threadIdx.x & 63;
    const size_t nwaves = (size_t)gridDim.x * 4, wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t ntiles = nvec / (64 * U);
    u32x4 A[U], B[U];
    unsigned acc = 0;
    size_t t = wave;
    if (t >= ntiles) return;
    const size_t last = ntiles - 1;
    auto load = [&](size_t tt, u32x4 (&buf)[U]) {
        if (OP == 2) { for (int u = 0; u < U; ++u) buf[u] = u32x4{(unsigned)tt, 1, 2, 3}; return; }
#pragma unroll
        for (int u = 0; u < U; ++u) buf[u] = ld<NTL>(src + (tt * U + u) * 64 + lane);
    };
    auto process = [&](size_t tt, const u32x4 (&buf)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (OP == 1) acc += buf[u].x ^ buf[u].y ^ buf[u].z ^ buf[u].w;
            else st<NTS>(dst + (tt * U + u) * 64 + lane, buf[u]);
        }
    };
    load(t, A);
    for (;;) {
        const size_t t1 = t + nwaves;
        load(t1 < last ? t1 : last, B);
        process(t, A);
        if (t1 >= ntiles) break;
        const size_t t2 = t1 + nwaves;
        load(t2 < last ? t2 : last, A);
        process(t1, B);
        if (t2 >= ntiles) break;
        t = t2;
    }
    if (OP == 1 && acc == 0x12345678u) *sink = acc;
}

// one-shot grid: every thread moves V vectors, block-contiguous (the classic copy kernel)
template <int V, bool NTL, bool NTS> __global__ __launch_bounds__(256) void oneshot(const u32x4 *src, u32x4 *dst, size_t nvec) {
    const size_t base = (size_t)blockIdx.x * 256 * V + threadIdx.x;
    u32x4 r[V];
#pragma unroll
    for (int v = 0; v < V; ++v) r[v] = ld<NTL>(src + base + v * 256);
#pragma unroll
    for (int v = 0; v < V; ++v) st<NTS>(dst + base + v * 256, r[v]);
}

// an empty kernel in the same grid: what a dependent launch costs by itself (the floor under any tiny tensor)
__global__ __launch_bounds__(256) void nothing(const u32x4 *, u32x4 *, size_t, unsigned *) {}
void run_nothing(const u32x4 *s, u32x4 *d, size_t nvec, unsigned *sink, int bpc, hipStream_t st_) {
    hipLaunchKernelGGL(nothing, dim3(bpc), dim3(256), 0, st_, s, d, nvec, sink);
}

struct Case { const char *name; void (*launch)(const u32x4 *, u32x4 *, size_t, unsigned *, int, hipStream_t); int bytes_factor; };

template <int U, bool NTL, bool NTS, int OP> void run_persistent(const u32x4 *s, u32x4 *d, size_t nvec, unsigned *sink, int bpc, hipStream_t st_) {
    hipLaunchKernelGGL((persistent<U, NTL, NTS, OP>), dim3(256 * bpc), dim3(256), 0, st_, s, d, nvec, sink);
}
template <int V, bool NTL, bool NTS> void run_oneshot(const u32x4 *s, u32x4 *d, size_t nvec, unsigned *, int, hipStream_t st_) {
    hipLaunchKernelGGL((oneshot<V, NTL, NTS>), dim3((unsigned)(nvec / (256 * V))), dim3(256), 0, st_, s, d, nvec);
}

int main(int argc, char **argv) {
    const size_t mib = argc > 1 ? atoi(argv[1]) : 32;
    const size_t bytes = mib << 20, nvec = bytes / 16;
    const int nsets = (int)((size_t)(3) * 1024 / (2 * mib)) + 2;   // ~3 GiB rotated
    std::vector<u32x4 *> S(nsets), D(nsets);
    for (int i = 0; i < nsets; ++i) {
        CHECK(hipMalloc(&S[i], bytes)); CHECK(hipMalloc(&D[i], bytes));
        CHECK(hipMemset(S[i], i + 1, bytes)); CHECK(hipMemset(D[i], 0, bytes));
    }
    unsigned *sink; CHECK(hipMalloc(&sink, 4));
    hipStream_t st_; CHECK(hipStreamCreate(&st_));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct Row { const char *name; void (*fn)(const u32x4 *, u32x4 *, size_t, unsigned *, int, hipStream_t); int bpc; double traffic; };
    const double both = 2.0 * bytes, one = 1.0 * bytes;
    std::vector<Row> rows = {
        {"empty kernel, 512 blocks (time only)", run_nothing, 512, both},
        {"empty kernel, 2048 blocks (time only)", run_nothing, 2048, both},
        {"persistent U1 8blk/CU copy", run_persistent<1, false, false, 0>, 8, both},
        {"persistent U1 8blk/CU copy nt-store", run_persistent<1, false, true, 0>, 8, both},
        {"persistent U1 8blk/CU copy nt-load nt-store", run_persistent<1, true, true, 0>, 8, both},
        {"persistent U2 8blk/CU copy nt-store", run_persistent<2, false, true, 0>, 8, both},
        {"persistent U2 4blk/CU copy nt-store", run_persistent<2, false, true, 0>, 4, both},
        {"persistent U4 4blk/CU copy nt-store", run_persistent<4, false, true, 0>, 4, both},
        {"persistent U4 2blk/CU copy nt-store", run_persistent<4, false, true, 0>, 2, both},
        {"oneshot V1 copy", run_oneshot<1, false, false>, 0, both},
        {"oneshot V1 copy nt-store", run_oneshot<1, false, true>, 0, both},
        {"oneshot V4 copy nt-store", run_oneshot<4, false, true>, 0, both},
        {"oneshot V8 copy nt-store", run_oneshot<8, false, true>, 0, both},
        {"persistent U1 8blk/CU read-only", run_persistent<1, false, false, 1>, 8, one},
        {"persistent U2 8blk/CU read-only", run_persistent<2, false, false, 1>, 8, one},
        {"persistent U1 8blk/CU read-only nt", run_persistent<1, true, false, 1>, 8, one},
        {"persistent U1 8blk/CU write-only", run_persistent<1, false, false, 2>, 8, one},
        {"persistent U1 8blk/CU write-only nt", run_persistent<1, false, true, 2>, 8, one},
        {"persistent U2 8blk/CU write-only nt", run_persistent<2, false, true, 2>, 8, one},
        {"persistent U4 4blk/CU write-only nt", run_persistent<4, false, true, 2>, 4, one},
        {"persistent U4 2blk/CU write-only nt", run_persistent<4, false, true, 2>, 2, one},
        {"persistent U4 4blk/CU write-only", run_persistent<4, false, false, 2>, 4, one},
        {"persistent U4 4blk/CU read-only", run_persistent<4, false, false, 1>, 4, one},
    };
    printf("%zu MiB per buffer, %d rotating sets; us per launch (GB/s, %% of 8 TB/s)\n", mib, nsets);
    for (auto &r : rows) {
        double res[2];
        for (int cold = 0; cold < 2; ++cold) {
            const int reps = cold ? 3 * nsets : 300;
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                for (int i = 0; i < (cold ? nsets : 5); ++i) r.fn(S[cold ? i : 0], D[cold ? i : 0], nvec, sink, r.bpc, st_);
                CHECK(hipEventRecord(e0, st_));
                for (int i = 0; i < reps; ++i) { const int k = cold ? i % nsets : 0; r.fn(S[k], D[k], nvec, sink, r.bpc, st_); }
                CHECK(hipEventRecord(e1, st_));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms * 1e3 / reps < best) best = ms * 1e3 / reps;
            }
            res[cold] = best;
        }
        printf("%-46s warm %7.2f us (%6.0f GB/s %5.1f%%) | cold %7.2f us (%6.0f GB/s %5.1f%%)\n", r.name, res[0], r.traffic / res[0] / 1e3,
               r.traffic / res[0] / 8e4, res[1], r.traffic / res[1] / 1e3, r.traffic / res[1] / 8e4);
    }
    return 0;
}
