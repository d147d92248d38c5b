// Streaming skeletons of the strided (axis-0) tile kernels: what the memory system gives a tile shape before any
// solver arithmetic is added.  A workgroup owns LINES adjacent k-columns x all n rows (stride sx) of one j; a thread
// owns M rows of LPT adjacent columns (8*LPT-byte accesses).  NJ adds the two j-neighbour reads of the fused kernel
// (L2 hits when the neighbouring tiles run next door).  One barrier in the middle stands for the separator exchange.
//   hipcc -O3 --offload-arch=gfx950 scripts/stream_probe.hip -o scripts/_build/stream_probe && scripts/_build/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ long xcd_chunk_tile(long b, long ntiles)
{
    const long q = ntiles >> 3, rem = ntiles & 7;
    const long x = b & 7, idx = b >> 3;
    return x * q + (x < rem ? x : rem) + idx;
}

template <int LPT> struct Vec;
template <> struct Vec<1> { typedef double T; };
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef double d4_t __attribute__((ext_vector_type(4)));
template <> struct Vec<2> { typedef d2_t T; };
template <> struct Vec<4> { typedef d4_t T; };

template <int LPT> __device__ __forceinline__ void acc(typename Vec<LPT>::T &a, const typename Vec<LPT>::T &b, double w) { a += w * b; }

__device__ __forceinline__ double first(double v) { return v; }
__device__ __forceinline__ double first(d2_t v) { return v.x; }
__device__ __forceinline__ double first(d4_t v) { return v.x; }

template <int M, int LPT, int LINES, bool NJ, int THREADS, int OCC>
__global__ __launch_bounds__(THREADS, OCC) void k_tile(const double *__restrict__ in, double *__restrict__ out, int n, int ny, int nz,
                                                       long sx, long ntiles, int tiles_inner)
{
    typedef typename Vec<LPT>::T V;
    __shared__ double sm[THREADS];
    constexpr int LANES = LINES / LPT;                  // lanes across the tile's k-extent
    const int tid = threadIdx.x;
    const long tile = xcd_chunk_tile(blockIdx.x, ntiles);
    const int j = (int)(tile / tiles_inner), ti = (int)(tile - (long)j * tiles_inner);
    const int kk = tid % LANES, sg = tid / LANES;
    const long base = (long)j * nz + (long)ti * LINES + (long)kk * LPT + (long)sg * M * sx;
    V d[M];
#pragma unroll
    for (int r = 0; r < M; ++r) d[r] = *reinterpret_cast<const V *>(in + base + (long)r * sx);
    if (NJ) {
        const long jm = (j > 0) ? -(long)nz : 0, jp = (j < ny - 1) ? (long)nz : 0;
#pragma unroll
        for (int r = 0; r < M; ++r) {
            const V a = *reinterpret_cast<const V *>(in + base + (long)r * sx + jm);
            const V b = *reinterpret_cast<const V *>(in + base + (long)r * sx + jp);
            acc<LPT>(d[r], a, 0.25);
            acc<LPT>(d[r], b, 0.25);
        }
    }
    // stand-in for the separator exchange
    sm[tid] = first(d[M - 1]);
    __syncthreads();
    const double x = sm[(tid + LANES) % THREADS];
#pragma unroll
    for (int r = 0; r < M; ++r) {
        d[r] += 1e-9 * x;
        __builtin_nontemporal_store(d[r], reinterpret_cast<V *>(out + base + (long)r * sx));
    }
}

// a thread owns M rows of TWO adjacent j-rows (same k): the inner j-neighbours are its own registers, the outer two are loaded
template <int M, int LINES, int THREADS, int OCC>
__global__ __launch_bounds__(THREADS, OCC) void k_tile_jpair(const double *__restrict__ in, double *__restrict__ out, int n, int ny, int nz,
                                                             long sx, long ntiles, int tiles_inner)
{
    __shared__ double sm[THREADS];
    const int tid = threadIdx.x;
    const long tile = xcd_chunk_tile(blockIdx.x, ntiles);
    const int j = 2 * (int)(tile / tiles_inner), ti = (int)(tile % tiles_inner);
    const int kk = tid % LINES, sg = tid / LINES;
    const long base = (long)j * nz + (long)ti * LINES + kk + (long)sg * M * sx;
    double d0[M], d1[M];
#pragma unroll
    for (int r = 0; r < M; ++r) { d0[r] = in[base + (long)r * sx]; d1[r] = in[base + (long)r * sx + nz]; }
    const long jm = (j > 0) ? -(long)nz : 0, jp = (j + 1 < ny - 1) ? 2L * nz : nz;
#pragma unroll
    for (int r = 0; r < M; ++r) {
        const double a = in[base + (long)r * sx + jm], b = in[base + (long)r * sx + jp];
        const double c0 = d0[r], c1 = d1[r];
        d0[r] = c0 + 0.25 * (a + c1);
        d1[r] = c1 + 0.25 * (c0 + b);
    }
    sm[tid] = d0[M - 1];
    __syncthreads();
    const double x = sm[(tid + LINES) % THREADS];
#pragma unroll
    for (int r = 0; r < M; ++r) {
        __builtin_nontemporal_store(d0[r] + 1e-9 * x, out + base + (long)r * sx);
        __builtin_nontemporal_store(d1[r] + 1e-9 * x, out + base + (long)r * sx + nz);
    }
}

template <int M, int LINES, int THREADS, int OCC>
static int run_jpair(const char *name, const double *in, double *out, int n, long sx)
{
    static_assert(THREADS == LINES * (512 / M), "threads");
    const int tiles_inner = n / LINES;
    const long ntiles = (long)(n / 2) * tiles_inner;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int it = 0; it < 12; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_tile_jpair<M, LINES, THREADS, OCC>), dim3((unsigned)ntiles), dim3(THREADS), 0, 0, in, out, n, n, n, sx, ntiles, tiles_inner);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2) ts.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    const double med = ts[ts.size() / 2], cells = (double)n * n * n;
    printf("%-44s  min %.3f  med %.3f ms   %.2f TB/s on 16 B/cell\n", name, ts[0], med, 16.0 * cells / (med * 1e-3) / 1e12);
    fflush(stdout);
    return 0;
}

// read-only / write-only / plain-store variants of the 16-line tile
template <int MODE>
__global__ __launch_bounds__(512, 2) void k_tile_rw(const double *__restrict__ in, double *__restrict__ out, int n, int nz, long sx,
                                                    long ntiles, int tiles_inner)
{
    constexpr int M = 16, LINES = 16;
    const int tid = threadIdx.x;
    const long tile = xcd_chunk_tile(blockIdx.x, ntiles);
    const int j = (int)(tile / tiles_inner), ti = (int)(tile % tiles_inner);
    const int kk = tid % LINES, sg = tid / LINES;
    const long base = (long)j * nz + (long)ti * LINES + kk + (long)sg * M * sx;
    double d[M];
#pragma unroll
    for (int r = 0; r < M; ++r) d[r] = (MODE == 2) ? (double)(tid + r) : in[base + (long)r * sx];
    if (MODE == 1) {
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < M; ++r) s += d[r];
        if (s == 12345.678) out[base] = s;
    } else {
#pragma unroll
        for (int r = 0; r < M; ++r) {
            if (MODE == 3) out[base + (long)r * sx] = d[r];
            else __builtin_nontemporal_store(d[r], out + base + (long)r * sx);
        }
    }
}
template <int MODE>
static int run_rw(const char *name, const double *in, double *out, int n, long sx, double bytes)
{
    const int tiles_inner = n / 16;
    const long ntiles = (long)n * tiles_inner;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int it = 0; it < 12; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_tile_rw<MODE>), dim3((unsigned)ntiles), dim3(512), 0, 0, in, out, n, n, sx, ntiles, tiles_inner);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double med = ts[ts.size() / 2], cells = (double)n * n * n;
    printf("%-44s  min %.3f  med %.3f ms   %.2f TB/s on %.0f B/cell\n", name, ts[0], med, bytes * cells / (med * 1e-3) / 1e12, bytes);
    fflush(stdout);
    return 0;
}

template <int M, int LPT, int LINES, bool NJ, int THREADS, int OCC>
static int run(const char *name, const double *in, double *out, int n, long sx, size_t dyn_lds = 0)
{
    static_assert(THREADS == (LINES / LPT) * (512 / M), "threads");
    const int tiles_inner = n / LINES;
    const long ntiles = (long)n * tiles_inner;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int it = 0; it < 12; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_tile<M, LPT, LINES, NJ, THREADS, OCC>), dim3((unsigned)ntiles), dim3(THREADS), dyn_lds, 0, in, out, n, n, n, sx, ntiles, tiles_inner);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2) ts.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    const double med = ts[ts.size() / 2], cells = (double)n * n * n;
    printf("%-44s  min %.3f  med %.3f ms   %.2f TB/s on 16 B/cell\n", name, ts[0], med, 16.0 * cells / (med * 1e-3) / 1e12);
    fflush(stdout);
    return 0;
}

int main()
{
    const int n = 512;
    const long sx = (long)n * n + 256;
    const size_t N = (size_t)n * sx + 4096;
    double *a, *b;
    CK(hipMalloc(&a, N * 8)); CK(hipMalloc(&b, N * 8));
    CK(hipMemset(a, 0, N * 8)); CK(hipMemset(b, 0, N * 8));
#define RUN(M, LPT, LINES, NJ, THREADS, OCC) if (run<M, LPT, LINES, NJ, THREADS, OCC>("M=" #M " lpt=" #LPT " lines=" #LINES " nj=" #NJ " thr=" #THREADS " occ=" #OCC, a, b, n, sx)) return 1
    if (run_rw<1>("16-line tile, read only", a, b, n, sx, 8.0)) return 1;
    if (run_rw<2>("16-line tile, write only (nt)", a, b, n, sx, 8.0)) return 1;
    if (run_rw<0>("16-line tile, read + nt write", a, b, n, sx, 16.0)) return 1;
    if (run_rw<3>("16-line tile, read + plain write", a, b, n, sx, 16.0)) return 1;
    if (run_jpair<16, 16, 512, 2>("jpair M=16 lines=16 thr=512 occ=2", a, b, n, sx)) return 1;
    if (run_jpair<16, 16, 512, 1>("jpair M=16 lines=16 thr=512 occ=1", a, b, n, sx)) return 1;
    if (run_jpair<8, 16, 1024, 1>("jpair M=8 lines=16 thr=1024 occ=1", a, b, n, sx)) return 1;
    if (run_jpair<16, 8, 256, 2>("jpair M=16 lines=8 thr=256 occ=2", a, b, n, sx)) return 1;
    // the same skeleton held to 2 and 1 workgroups per CU by dynamic LDS (the real fused kernel runs 2 per CU at 128 VGPRs)
    if (run<16, 1, 16, true, 512, 2>("M=16 lines=16 nj=true, 2 workgroups per CU", a, b, n, sx, 72 * 1024)) return 1;
    if (run<16, 1, 16, true, 512, 2>("M=16 lines=16 nj=true, 1 workgroup per CU", a, b, n, sx, 150 * 1024)) return 1;
    if (run<16, 1, 16, true, 512, 2>("M=16 lines=16 nj=true, 3 workgroups per CU", a, b, n, sx, 50 * 1024)) return 1;
    if (run<16, 1, 16, false, 512, 2>("M=16 lines=16 nj=false, 2 workgroups per CU", a, b, n, sx, 72 * 1024)) return 1;
    RUN(16, 1, 16, false, 512, 2);
    RUN(16, 1, 16, true, 512, 2);
    RUN(32, 1, 32, false, 512, 2);
    RUN(32, 1, 32, true, 512, 2);
    RUN(16, 1, 32, false, 1024, 1);
    RUN(16, 1, 32, true, 1024, 1);
    RUN(16, 2, 32, false, 512, 2);
    RUN(16, 2, 32, true, 512, 2);
    RUN(16, 2, 32, false, 512, 1);
    RUN(16, 2, 32, true, 512, 1);
    RUN(16, 2, 16, false, 256, 2);
    RUN(16, 2, 16, true, 256, 2);
    RUN(16, 2, 16, true, 256, 4);
    RUN(32, 2, 32, false, 256, 2);
    RUN(32, 2, 32, true, 256, 2);
    RUN(16, 4, 64, false, 512, 1);
    RUN(16, 4, 64, true, 512, 1);
    RUN(16, 4, 32, false, 256, 2);
    RUN(16, 4, 32, true, 256, 2);
    RUN(8, 2, 32, false, 1024, 1);
    RUN(8, 2, 32, true, 1024, 1);
    RUN(8, 4, 64, true, 1024, 1);
    RUN(8, 4, 32, true, 512, 2);
    return 0;
}
