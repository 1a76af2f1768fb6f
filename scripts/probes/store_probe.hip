// store_probe.hip -- what 16 KB of row stores per event cost a wave-per-chain kernel at three waves per SIMD (config 4's shape:
// 10^4 chains x 199 events x two rows of 1000 doubles), beside W x 16 fp64 FMAs of arithmetic per event.
//   MODE 0  no stores                        MODE 1  32 x global_store_dwordx2 nt per event (what di_unit_kernel does)
//   MODE 2  16 x global_store_dwordx4 nt     MODE 3  as 1, plain stores
//   MODE 4  a writer wave per block: three waves compute and hand their rows over through LDS, the fourth stores them
//   MODE 5  as 1, and eight loads (a column of P) whose result is awaited before the stores are issued (vmcnt coupling)
//   MODE 6  16 x plain dwordx4               MODE 7  as 3, the whole X row before the V row      MODE 8  as 6, X row then V row
// hipcc --offload-arch=gfx950 -O3 store_probe.hip -o store_probe_all ; ./store_probe_all [chains]   (every mode on the same arrays)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#ifndef MODE
#define MODE 1
#endif
#ifndef PROBE_D
#define PROBE_D 1000       // doubles per row: 1000 = config 4 (rows of 62.5 L2 lines: every other row starts in the middle of one); 1008 = rows of 63 whole lines
#endif
constexpr int WAVE = 64, KC = 16, D = PROBE_D, E = 199;

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

template <int M>
__global__ __launch_bounds__(256, 3) void probe(double *X, double *V, const double *P, int W, int chains, double *sink) {
    extern __shared__ double lds[];                       // 48 KB per block: three blocks per CU = three waves per SIMD
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    double a[KC], b[KC];
#pragma unroll
    for (int s = 0; s < KC; ++s) { a[s] = 1.0 + 1e-9 * (lane + s); b[s] = 1.0 - 1e-12 * lane; }
    const double c0 = 1e-7;
    if constexpr (M == 4) {
        // waves 0..2: chains 3 blockIdx + w; wave 3: their writer
        volatile int *flag = (volatile int *)(lds + 3 * 2 * KC * WAVE);      // ready[3], taken[3]
        if (threadIdx.x < 6) flag[threadIdx.x] = 0;
        __syncthreads();
        if (w < 3) {
            const int chain = blockIdx.x * 3 + w;
            if (chain >= chains) return;                   // (the writer learns it from the chain count)
            double *row = lds + w * 2 * KC * WAVE;
            for (int e = 1; e <= E; ++e) {
                for (int r = 0; r < W; ++r) {
#pragma unroll
                    for (int s = 0; s < KC; ++s) a[s] = fma(a[s], b[s], c0);
                }
                while (flag[3 + w] != e - 1) __builtin_amdgcn_s_sleep(1);     // the writer has copied the previous rows
#pragma unroll
                for (int s = 0; s < KC; ++s) { row[s * WAVE + lane] = a[s]; row[(KC + s) * WAVE + lane] = b[s]; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) flag[w] = e;
            }
        } else {
            int done[3] = {0, 0, 0};
            int live = 0;
            for (int q = 0; q < 3; ++q) if (blockIdx.x * 3 + q < chains) ++live;
            int finished = 0;
            while (finished < live) {
                for (int q = 0; q < 3; ++q) {
                    if (blockIdx.x * 3 + q >= chains || done[q] == E) continue;
                    const int e = flag[q];
                    if (e == done[q]) continue;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    const double *row = lds + q * 2 * KC * WAVE;
                    double xr[KC], vr[KC];
#pragma unroll
                    for (int s = 0; s < KC; ++s) { xr[s] = row[s * WAVE + lane]; vr[s] = row[(KC + s) * WAVE + lane]; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) flag[3 + q] = e;
                    const size_t off = ((size_t)(blockIdx.x * 3 + q) * (E + 1) + e) * D;
#pragma unroll
                    for (int s = 0; s < KC; ++s) {
                        const int i = s * WAVE + lane;
                        if (i < D) { __builtin_nontemporal_store(xr[s], X + off + i); __builtin_nontemporal_store(vr[s], V + off + i); }
                    }
                    done[q] = e;
                    if (e == E) ++finished;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        return;
    } else {
        const int chain = blockIdx.x * 4 + w;
        if (chain >= chains) return;
        for (int e = 1; e <= E; ++e) {
            for (int r = 0; r < W; ++r) {
#pragma unroll
                for (int s = 0; s < KC; ++s) a[s] = fma(a[s], b[s], c0);
            }
            const size_t off = ((size_t)chain * (E + 1) + e) * D;
            if constexpr (M == 5) {
                const double *col = P + (size_t)((chain * 7 + e * 13) % D) * D;
                double pc[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) pc[s] = col[s * WAVE + lane];
#pragma unroll
                for (int s = 0; s < 8; ++s) b[s] = fma(1e-30, pc[s], b[s]);
            }
            if constexpr (M == 1 || M == 5) {
#pragma unroll
                for (int s = 0; s < KC; ++s) {
                    const int i = s * WAVE + lane;
                    if (i < D) { __builtin_nontemporal_store(a[s], X + off + i); __builtin_nontemporal_store(b[s], V + off + i); }
                }
            } else if constexpr (M == 3) {
#pragma unroll
                for (int s = 0; s < KC; ++s) {
                    const int i = s * WAVE + lane;
                    if (i < D) { X[off + i] = a[s]; V[off + i] = b[s]; }
                }
            } else if constexpr (M == 7) {                     // the X row, then the V row (plain stores)
#pragma unroll
                for (int s = 0; s < KC; ++s) { const int i = s * WAVE + lane; if (i < D) X[off + i] = a[s]; }
#pragma unroll
                for (int s = 0; s < KC; ++s) { const int i = s * WAVE + lane; if (i < D) V[off + i] = b[s]; }
            } else if constexpr (M == 6) {                     // sixteen 16-byte plain stores
                typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int s = 0; s < KC / 2; ++s) {
                    const int i = s * 2 * WAVE + 2 * lane;
                    if (i < D) {
                        d2 xv = {a[2 * s], a[2 * s + 1]}, vv = {b[2 * s], b[2 * s + 1]};
                        *(d2 *)(X + off + i) = xv;
                        *(d2 *)(V + off + i) = vv;
                    }
                }
            } else if constexpr (M == 8) {                     // as 6, the X row then the V row
                typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int s = 0; s < KC / 2; ++s) { const int i = s * 2 * WAVE + 2 * lane; if (i < D) { d2 xv = {a[2 * s], a[2 * s + 1]}; *(d2 *)(X + off + i) = xv; } }
#pragma unroll
                for (int s = 0; s < KC / 2; ++s) { const int i = s * 2 * WAVE + 2 * lane; if (i < D) { d2 vv = {b[2 * s], b[2 * s + 1]}; *(d2 *)(V + off + i) = vv; } }
            } else if constexpr (M == 2) {
                // sixteen 16-byte stores per lane: the same bytes (pairs of coordinates; which value lands where does not matter here)
                typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int s = 0; s < KC / 2; ++s) {
                    const int i = s * 2 * WAVE + 2 * lane;
                    if (i < D) {
                        d2 xv = {a[2 * s], a[2 * s + 1]}, vv = {b[2 * s], b[2 * s + 1]};
                        __builtin_nontemporal_store(xv, (d2 *)(X + off + i));
                        __builtin_nontemporal_store(vv, (d2 *)(V + off + i));
                    }
                }
            }
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int s = 0; s < KC; ++s) acc += a[s] + b[s];
    if (acc == 123.456) sink[0] = acc;
}

template <int M> static float run(double *X, double *V, double *P, double *sink, int W, int chains, size_t shmem) {
    const int blocks = (M == 4) ? (chains + 2) / 3 : (chains + 3) / 4;
    hipFuncSetAttribute((const void *)probe<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipEvent_t t0, t1;
    hipEventCreate(&t0); hipEventCreate(&t1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(t0, 0);
        hipLaunchKernelGGL(probe<M>, dim3(blocks), dim3(256), shmem, 0, X, V, P, W, chains, sink);
        hipEventRecord(t1, 0);
        if (hipEventSynchronize(t1) != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); exit(1); }
        float ms; hipEventElapsedTime(&ms, t0, t1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (M == 0) ? 0.0 : (double)chains * E * D * 16.0;
    printf("MODE %d  W %3d  LDS %2zu KB/block: %.3f ms  (%.2f TB/s of rows)\n", M, W, shmem / 1024, best, bytes / best / 1e9);
    fflush(stdout);
    return best;
}

// every mode on the SAME two arrays (the physical placement of an allocation moves the write rate by ~10 % on this part)
int main(int argc, char **argv) {
    const int chains = argc > 1 ? atoi(argv[1]) : 10000;
    const size_t n = (size_t)chains * (E + 1) * D;
    double *X, *V, *P, *sink;
    if (hipMalloc(&X, n * 8) != hipSuccess || hipMalloc(&V, n * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&P, (size_t)D * D * 8 + 4096); hipMemset(P, 0, (size_t)D * D * 8 + 4096);
    hipMalloc(&sink, 8);
    const size_t three = 48 * 1024, two = 64 * 1024;       // blocks per CU = waves per SIMD: 3, 2
    for (int round = 0; round < 2; ++round)
        for (int W : {0, 30, 60}) {
            run<0>(X, V, P, sink, W, chains, three);
            run<1>(X, V, P, sink, W, chains, three);
            run<3>(X, V, P, sink, W, chains, three);
            run<7>(X, V, P, sink, W, chains, three);
            run<2>(X, V, P, sink, W, chains, three);
            run<6>(X, V, P, sink, W, chains, three);
            run<8>(X, V, P, sink, W, chains, three);
            run<5>(X, V, P, sink, W, chains, three);
            run<4>(X, V, P, sink, W, chains, three);
            run<3>(X, V, P, sink, W, chains, two);
            run<6>(X, V, P, sink, W, chains, two);
        }
    return 0;
}
