// dgemm_mfma.hip -- the dense-precision target's matrix product on the fp64 matrix cores, hand-written for gfx950:
//     Out (M x N) = A (M x K) * B (K x N),   all column-major,   A = the precision matrix P (d x d),  B = [x | v] of all chains.
// grad-U = P x is the one GEMM-shaped operation of the path (north star: "MFMA only if grad-U is a dense mat-vec"); the batched dense
// forms of dense_batched.hip multiply P with the state columns of ALL chains at once -- every 256 events (Zig-Zag, incremental form) or
// once per event (BPS, Boomerang).  v_mfma_f64_16x16x4_f64: D(16x16) += A(16x4) B(4x16); lane l of the wave holds A[l % 16][l / 16],
// B[l / 16][l % 16] and the four accumulator rows D[4 (l / 16) + r][l % 16].
//
// Tiling: one workgroup (four waves) computes a 128 x 64 tile, each wave a 64 x 32 quarter = 4 x 2 accumulator tiles; three workgroups per
// CU (three waves per SIMD take turns at the matrix core); the K loop advances in blocks of 8: A- and B-slabs staged through LDS (double
// buffered, the next block's global loads in flight during the current block's MFMAs), k-major with padded rows so that the four k-rows
// of a fragment read fall into two disjoint halves of the 64 LDS banks; the fragments of the next k-step are read before the current
// step's MFMAs issue.  An fp64 MFMA occupies its SIMD for 64 cycles: eight independent accumulators per wave keep it issuing back to back.  Workgroups are numbered so that each XCD works on a contiguous range of tiles (M fastest): the B slab of a tile
// column is fetched into ONE L2.  Edges: M and N by clamped loads + masked stores, K by zero-filled A rows.
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdint.h>
#include <stdlib.h>

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int BN = 64, BK = 8;
constexpr int LDB = BN + 16;          // doubles per k-row of the A / B slab in LDS: a row stride of 128 (mod 256) bytes puts the
                                                     // four k-rows of a fragment read into two disjoint halves of the 64 banks

template <int BM>
__global__ __launch_bounds__(256, 3) void dgemm_mfma_kernel(int M, int N, int K, const double *__restrict__ A, int lda,
                                                           const double *__restrict__ B, int ldb, double *__restrict__ C, int ldc,
                                                           int tiles_m, int tiles_n, const double *__restrict__ zeros) {
    constexpr int LDA = BM + 16;                      // doubles per k-row of the A slab in LDS (LDB: of the B slab)
    constexpr int WM = BM / 2, WN = BN / 2;           // the wave's quarter of the tile: WM x 32 = MT x 2 accumulator tiles of 16 x 16
    constexpr int MT = WM / 16, NT = WN / 16;
    constexpr int PR = BM / 2, KPP = 256 / PR, AQ = BK / KPP;   // A slab: row pairs per k-row, k-rows per pass of the 256 threads, passes
    __shared__ double As[2][BK * LDA];               // 2 x 9 KB
    __shared__ double Bs[2][BK * LDB];               // 2 x 5 KB
    // XCD-aware tile numbering: workgroup b runs on XCD b % 8; give every XCD a contiguous run of tiles (M fastest), so that the B slab of a
    // tile column and the rows of A it meets are fetched into ONE L2
    const int n_tiles = tiles_m * tiles_n;
    const int b = blockIdx.x;
    const int per = (n_tiles + 7) / 8;
    const int tile = (b % 8) * per + b / 8;
    if (tile >= n_tiles || b / 8 >= per) return;
    const int tm = tile % tiles_m, tn = tile / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;          // the wave's quarter: rows wm * 64 .., columns wn * 32 ..

    // ---- global -> register staging of one K-block.  A slab BM x BK: k-rows of 128 contiguous doubles; thread t loads the pair
    // i = 2 (t % 64) + {0, 1} of k-rows t / 64 + 4 q, q = 0..3 (a wave reads one whole 1 KB row per instruction).  B slab BK x BN: columns of 16
    // contiguous doubles = one 128-byte line; thread t loads the k-pair 2 (t % 8) + {0, 1} of columns t / 8 + 32 q, q = 0, 1 (a wave reads
    // eight whole lines per instruction).  Both as 16-byte loads (lda, ldb even and the bases 16-byte aligned: checked by the host).
    // Branch-free: out-of-range rows / columns / k are CLAMPED to the last valid aligned pair (M, K even: checked by the host), never skipped --
    // a conditional load would put an s_waitcnt vmcnt(0) between the loads of a block and expose the memory latency once per block.
    // K edge: rows of A beyond K are read from a pair of zeros in global memory -- the ADDRESS is selected, not the value, so that no
    // instruction has to wait for the load before the block's MFMAs -- and make the padded k-steps contribute nothing.
    // (Plain scalars and macros, no arrays behind lambdas: the compiler moved such an array into LDS and waited for every load.)
    const int ia = m0 + 2 * (tid % PR);
    const double *a_ptr = A + ((ia + 1 < M) ? ia : M - 2);
    const int j0 = n0 + (tid >> 2);
    const double *b_ptr0 = B + (size_t)((j0 < N) ? j0 : N - 1) * ldb;
    const int ka = tid / PR, kp = 2 * (tid & 3);
    double2 ra0, ra1 = {0.0, 0.0}, rb0;
#define GEMM_LOAD_A(dst, q, k0) { const int k_ = (k0) + ka + KPP * (q); dst = *(const double2 *)((k_ < K) ? a_ptr + (size_t)k_ * lda : zeros); }
#define GEMM_LOAD_BLOCK(k0) { GEMM_LOAD_A(ra0, 0, k0) if constexpr (AQ == 2) GEMM_LOAD_A(ra1, 1, k0)                                  \
        const int kc_ = ((k0) + kp + 1 < K) ? (k0) + kp : K - 2;   /* (finite values from valid rows: the matching rows of A are zero) */ \
        rb0 = *(const double2 *)(b_ptr0 + kc_); }
    double *as_w = &As[0][ka * LDA + 2 * (tid % PR)], *bs_w = &Bs[0][kp * LDB + (tid >> 2)];
#define GEMM_STORE_BLOCK(buf) { double *aw_ = as_w + (buf) * (BK * LDA), *bw_ = bs_w + (buf) * (BK * LDB);                              \
        *(double2 *)(aw_) = ra0; if constexpr (AQ == 2) *(double2 *)(aw_ + KPP * LDA) = ra1; bw_[0] = rb0.x; bw_[LDB] = rb0.y; }

    double4_t acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (double4_t){0.0, 0.0, 0.0, 0.0};

    const int nkb = (K + BK - 1) / BK;
    GEMM_LOAD_BLOCK(0)
    GEMM_STORE_BLOCK(0)
    __syncthreads();
    const int fr = lane >> 4, fc = lane & 15;         // fragment coordinates of this lane: k-row fr, row (A) / column (B) fc
    const double *ap = &As[0][fr * LDA + wm * WM + fc], *bp = &Bs[0][fr * LDB + wn * WN + fc];
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nkb) GEMM_LOAD_BLOCK((kb + 1) * BK)   // in flight during this block's MFMAs
        const double *ab = ap + buf * (BK * LDA), *bb0 = bp + buf * (BK * LDB);
        double a[2][MT], bf[2][NT];                    // fragments of two k-steps: the next step's reads are issued before this step's MFMAs
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) a[0][mi] = ab[mi * 16];
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) bf[0][ni] = bb0[ni * 16];
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < BK / 4) {
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) a[nxt][mi] = ab[(ks + 1) * 4 * LDA + mi * 16];
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) bf[nxt][ni] = bb0[(ks + 1) * 4 * LDB + ni * 16];
            }
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int ni = 0; ni < NT; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][mi], bf[cur][ni], acc[mi][ni], 0, 0, 0);
        }
        if (kb + 1 < nkb) {
            GEMM_STORE_BLOCK(buf ^ 1)                 // the other buffer: nobody reads it during this block
            __syncthreads();
        }
    }
    // ---- epilogue.  v_mfma_f64_16x16x4_f64 leaves D[4 r + (lane / 16)][lane % 16] in accumulator register r (the fp64 form interleaves the
    // four rows a lane holds with the lane quarters -- unlike the fp32 16x16x4 form, where a lane holds four CONSECUTIVE rows)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const int j = n0 + wn * WN + ni * 16 + fc;
            if (j < N) {
                double *dst = C + (size_t)j * ldc;
                const double4_t v = acc[mi][ni];
                const int i = m0 + wm * WM + mi * 16 + fr;
                if (i < M) dst[i] = v.x;
                if (i + 4 < M) dst[i + 4] = v.y;
                if (i + 8 < M) dst[i + 8] = v.z;
                if (i + 12 < M) dst[i + 12] = v.w;
            }
        }
    }
}

}  // namespace

// Out = A * B on `stream`.  Requirements: column-major; M, K, lda, ldb even and A, B 16-byte aligned (the engine's workspaces are; an odd
// dimension goes to rocBLAS).  Returns 0 or the hipError_t of the launch.
// pdmp_dgemm_mfma_polite: the same product launched so that at most `max_wg` of its workgroups fit on a CU (extra, unused dynamic LDS): a kernel
// of another stream -- the step kernel of the other half-ensemble, dense_batched.hip -- finds registers and LDS beside it.
static thread_local int g_polite_wg = 0;      // (per calling thread: two engine handles on two host threads may run products at once)
extern "C" int pdmp_dgemm_mfma(int M, int N, int K, const double *A, int lda, const double *B, int ldb, double *C, int ldc, void *stream);
extern "C" int pdmp_dgemm_mfma_polite(int max_wg, int M, int N, int K, const double *A, int lda, const double *B, int ldb, double *C, int ldc, void *stream) {
    g_polite_wg = max_wg;
    const int rc = pdmp_dgemm_mfma(M, N, K, A, lda, B, ldb, C, ldc, stream);
    g_polite_wg = 0;
    return rc;
}
extern "C" int pdmp_dgemm_mfma(int M, int N, int K, const double *A, int lda, const double *B, int ldb, double *C, int ldc, void *stream) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if ((M & 1) || (K & 1) || (lda & 1) || (ldb & 1) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return (int)hipErrorInvalidValue;
    static double *zeros[16] = {nullptr};              // 16 bytes of zeros per device, allocated on first use, never freed
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return (int)hipErrorInvalidDevice;
    {
        static std::mutex zmu;
        std::lock_guard<std::mutex> lock(zmu);
        if (!zeros[dev]) {
            double *z = nullptr;
            if (hipMalloc(&z, 64) != hipSuccess) return (int)hipErrorOutOfMemory;
            if (hipMemset(z, 0, 64) != hipSuccess) { (void)hipFree(z); return (int)hipErrorUnknown; }
            zeros[dev] = z;
        }
    }
    // 128 x 64 tiles unless they would leave the last round of tiles mostly alone on the machine (a lone workgroup per CU does not fill the
    // matrix core): then 64 x 64 tiles, twice as many.  PDMP_DGEMM_BM = 64 | 128 pins the shape (A/B runs).
    const int tiles_n = (N + BN - 1) / BN;
    const int t128 = ((M + 127) / 128) * tiles_n;
    static const int pin = getenv("PDMP_DGEMM_BM") ? atoi(getenv("PDMP_DGEMM_BM")) : 0;
    const bool big = pin ? (pin == 128) : (t128 >= 2048);
    const int bm = big ? 128 : 64;
    const int tiles_m = (M + bm - 1) / bm;
    const int n_tiles = tiles_m * tiles_n;
    const int per = (n_tiles + 7) / 8;
    // LDS per workgroup: 28 KB (128-row tiles) / 20 KB (64-row tiles) of 160 KB per CU; a cap of w workgroups per CU = padding to just over 160 / (w + 1) KB
    size_t pad = 0;
    if (g_polite_wg > 0) { const size_t own = big ? 28672 : 20480, want = (size_t)(160 * 1024) / (size_t)(g_polite_wg + 1) + 1024; pad = want > own ? want - own : 0; }
    if (big) hipLaunchKernelGGL((dgemm_mfma_kernel<128>), dim3((unsigned)(per * 8)), dim3(256), pad, (hipStream_t)stream, M, N, K, A, lda, B, ldb, C, ldc, tiles_m, tiles_n, zeros[dev]);
    else hipLaunchKernelGGL((dgemm_mfma_kernel<64>), dim3((unsigned)(per * 8)), dim3(256), pad, (hipStream_t)stream, M, N, K, A, lda, B, ldb, C, ldc, tiles_m, tiles_n, zeros[dev]);
    return (int)hipGetLastError();
}
