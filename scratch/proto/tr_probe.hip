// probe: what does ds_read_b64_tr_b16 return?  LDS holds element index; lane l of a 16-lane group supplies the address of
// 4 contiguous elements: row (l%16)/4, columns 4*(l%4).. of a [4][16] block whose rows are `stride` elements apart
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(uint16_t* out, int stride) {
    __shared__ uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = static_cast<uint16_t>(i);
    __syncthreads();
    const int l = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int elem = (l / 4) * stride + (l % 4) * 4 + g * 16;       // group g: columns 16g..16g+15
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + elem));
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = static_cast<uint16_t>(v[e]);
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int stride : {64, 256}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("stride %d (element index = row*stride + col)\n", stride);
        for (int lane = 0; lane < 64; ++lane) {
            printf(" lane %2d:", lane);
            for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[lane * 4 + e] / stride, h[lane * 4 + e] % stride);
            if (lane % 4 == 3) printf("\n");
        }
    }
    return 0;
}
