// oob_probe.hip -- do buffer stores that the bounds check of their descriptor drops (offset beyond num_records: the lane kernel's
// "kNoRow") still reach the memory side?  Counted with  rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -- ./oob_probe
// Each kernel writes the same 1 GiB once with 16-byte streaming buffer stores (a wave instruction = 8 pieces of 128 bytes, the lane
// kernel's tile store); the variants add stores that are dropped:
//   k_plain      nothing dropped                       k_extra_oob   one more, wholly out-of-range store per store
//   k_half_oob   every store has half of its 128-byte pieces out of range (the other half is written by a second store)
//   k_piece64    64-byte pieces (4 lanes each), 16 per instruction, the two halves of a 128-byte line by different instructions
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kNoRow = 0x80000000u;
constexpr int AUX = 2;                                   // nt

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(double *base, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000); }

// the 1 GiB is cut into 1 MiB windows, one per workgroup pass; inside a window a wave instruction covers 8 x 128 B that lie 128 KiB apart
// (like eight chains' rows), the next instruction the next 128 B of each
template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int windows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int w = blockIdx.x; w < windows; w += gridDim.x) {
        double *base = out + (size_t)w * (1 << 17);                      // 1 MiB
        const __amdgpu_buffer_rsrc_t rs = rsrc(base, 1u << 20);
        // wave `wave` writes bytes [wave * 32 KiB, +32 KiB) of each of the eight 128-KiB streams
        for (int it = 0; it < 256; ++it) {
            u32x4 v = {(uint32_t)it, (uint32_t)lane, 0u, 0x3ff00000u};
            if (MODE == 3) {                                             // 64-byte pieces: 16 streams of 64 KiB, 4 lanes each
                const uint32_t off = (uint32_t)(lane >> 2) * (64u << 10) + (uint32_t)wave * (16u << 10) + (uint32_t)it * 64u + (uint32_t)(lane & 3) * 16u;
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, AUX);
                continue;
            }
            const uint32_t off = (uint32_t)(lane >> 3) * (128u << 10) + (uint32_t)wave * (32u << 10) + (uint32_t)it * 128u + (uint32_t)(lane & 7) * 16u;
            if (MODE == 0) __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, AUX);
            if (MODE == 1) { __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, AUX); __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)kNoRow, 0, AUX); }
            if (MODE == 2) {
                const bool lo = ((lane >> 3) & 1) == 0;
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(lo ? off : kNoRow), 0, AUX);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(lo ? kNoRow : off), 0, AUX);
            }
        }
    }
}

int main() {
    double *out;
    const size_t bytes = 1ull << 30;
    if (hipMalloc(&out, bytes) != hipSuccess) return 1;
    const int windows = (int)(bytes >> 20);
    hipLaunchKernelGGL(k<0>, dim3(512), dim3(256), 0, 0, out, windows);
    hipLaunchKernelGGL(k<1>, dim3(512), dim3(256), 0, 0, out, windows);
    hipLaunchKernelGGL(k<2>, dim3(512), dim3(256), 0, 0, out, windows);
    hipLaunchKernelGGL(k<3>, dim3(512), dim3(256), 0, 0, out, windows);
    if (hipDeviceSynchronize() != hipSuccess) { printf("failed\n"); return 1; }
    printf("ok: each kernel wrote %zu bytes = %zu 64-byte lines\n", bytes, bytes / 64);
    return 0;
}
