// Go / no-go gate for a BRICKED field layout: streaming skeletons -- the loads of a tile,
// one barrier, nt stores, no solver arithmetic -- of the three sweep access patterns at 512^3 in the shipped padded-plane
// layout and in a layout of bricks of BI planes x 16 k-cells of one j-row:
//     element (i, j, k) at (((i / BI) * ny + j) * (nz / 16) + k / 16) * (BI * 16) + (i % BI) * 16 + k % 16
// (BI = 4: 512-byte bricks, BI = 16: 2 KiB bricks).  Gate: go on to real kernels only if the fused-pattern skeleton (axis-0
// tile + the two j-neighbour reads) reaches <= 0.45 ms and neither the axis-1 nor the axis-2 pattern loses more than 3 %.
//   hipcc -O3 --offload-arch=gfx950 scripts/brick_probe.hip -o scripts/_build/brick_probe && scripts/_build/brick_probe
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

struct Dim { int n, ny, nz; long sx; };

// ---- addressing: BI = 0 padded planes (the shipped layout), BI > 0 bricks ------------------------------------------
template <int BI>
__device__ __forceinline__ long addr(const Dim &D, int i, int j, int k)
{
    if (BI == 0) return (long)i * D.sx + (long)j * D.nz + k;
    const int nkb = D.nz >> 4;
    return ((((long)(i / BI) * D.ny + j) * nkb + (k >> 4)) * (BI * 16)) + (i % BI) * 16 + (k & 15);
}

// ---- axis-0 tile (the fused kernel's shape): 16 k-lines x all rows i of one j; thread (sg, kk) owns M = 16 consecutive rows
template <int BI, bool NJ>
__global__ __launch_bounds__(512, 2) void k_axis0(const double *__restrict__ in, double *__restrict__ out, Dim D, long ntiles)
{
    __shared__ double sm[512];
    constexpr int M = 16;
    const int tid = threadIdx.x;
    const long tile = xcd_chunk_tile(blockIdx.x, ntiles);
    const int tiles_inner = D.nz / 16;
    const int j = (int)(tile / tiles_inner), kb = (int)(tile % tiles_inner);
    const int kk = tid & 15, sg = tid >> 4;
    double d[M];
#pragma unroll
    for (int r = 0; r < M; ++r) d[r] = in[addr<BI>(D, sg * M + r, j, kb * 16 + kk)];
    if (NJ) {
        const int jm = j > 0 ? j - 1 : j, jp = j < D.ny - 1 ? j + 1 : j;
#pragma unroll
        for (int r = 0; r < M; ++r)
            d[r] += 0.25 * (in[addr<BI>(D, sg * M + r, jm, kb * 16 + kk)] + in[addr<BI>(D, sg * M + r, jp, kb * 16 + kk)]);
    }
    sm[tid] = d[M - 1];
    __syncthreads();
    const double x = sm[(tid + 16) & 511];
#pragma unroll
    for (int r = 0; r < M; ++r) __builtin_nontemporal_store(d[r] + 1e-9 * x, out + addr<BI>(D, sg * M + r, j, kb * 16 + kk));
}

// the same tile with the lanes of a wave covering WHOLE bricks per instruction (BI = 4: lane = (ii, kk), a thread's rows are
// ii + 4 m -- not consecutive, a real kernel would have to transpose through LDS): the upper bound of what bricks can give
template <bool NJ>
__global__ __launch_bounds__(512, 2) void k_axis0_wholebrick(const double *__restrict__ in, double *__restrict__ out, Dim D, long ntiles)
{
    __shared__ double sm[512];
    constexpr int M = 16, BI = 4;
    const int tid = threadIdx.x;
    const long tile = xcd_chunk_tile(blockIdx.x, ntiles);
    const int tiles_inner = D.nz / 16;
    const int j = (int)(tile / tiles_inner), kb = (int)(tile % tiles_inner);
    const int kk = tid & 15, ii = (tid >> 4) & 3, wv = tid >> 6;        // wave wv owns bricks wv*16 .. wv*16+15 of the line
    double d[M];
#pragma unroll
    for (int r = 0; r < M; ++r) d[r] = in[addr<BI>(D, (wv * M + r) * BI + ii, j, kb * 16 + kk)];
    if (NJ) {
        const int jm = j > 0 ? j - 1 : j, jp = j < D.ny - 1 ? j + 1 : j;
#pragma unroll
        for (int r = 0; r < M; ++r)
            d[r] += 0.25 * (in[addr<BI>(D, (wv * M + r) * BI + ii, jm, kb * 16 + kk)] + in[addr<BI>(D, (wv * M + r) * BI + ii, jp, kb * 16 + kk)]);
    }
    sm[tid] = d[M - 1];
    __syncthreads();
    const double x = sm[(tid + 16) & 511];
#pragma unroll
    for (int r = 0; r < M; ++r) __builtin_nontemporal_store(d[r] + 1e-9 * x, out + addr<BI>(D, (wv * M + r) * BI + ii, j, kb * 16 + kk));
}

// ---- axis-1 tile: IT planes x 16 k-lines x all rows j; thread (sg, ii, kk) owns M = 32 consecutive rows j.
// shipped kernel: IT = 1 (256 threads); bricks: IT = 4 planes of one brick layer (1024 threads, 512-byte pieces per row)
template <int BI, int IT>
__global__ __launch_bounds__(256 * IT, IT == 1 ? 4 : 1) void k_axis1(const double *__restrict__ in, double *__restrict__ out, Dim D, long ntiles)
{
    __shared__ double sm[256 * IT];
    constexpr int M = 32;
    const int tid = threadIdx.x;
    const long tile = xcd_chunk_tile(blockIdx.x, ntiles);
    const int tiles_inner = D.nz / 16;
    const int ig = (int)(tile / tiles_inner), kb = (int)(tile % tiles_inner);
    const int kk = tid & 15, ii = (tid >> 4) % IT, sg = tid / (16 * IT);
    const int i = ig * IT + ii;
    double d[M];
#pragma unroll
    for (int r = 0; r < M; ++r) d[r] = in[addr<BI>(D, i, sg * M + r, kb * 16 + kk)];
    sm[tid] = d[M - 1];
    __syncthreads();
    const double x = sm[(tid + 16) % (256 * IT)];
#pragma unroll
    for (int r = 0; r < M; ++r) __builtin_nontemporal_store(d[r] + 1e-9 * x, out + addr<BI>(D, i, sg * M + r, kb * 16 + kk));
}

// ---- axis-2 (contiguous) pattern: a wave owns two lines of 512 cells; per instruction every lane moves 16 bytes.
// shipped layout: the two lines are 2 x 4 KiB contiguous; bricks (BI = 4): the two lines (ii, ii + 1) are 32 pieces of 256 bytes
template <int BI>
__global__ __launch_bounds__(256, 4) void k_axis2(const double *__restrict__ in, double *__restrict__ out, Dim D, long npairs)
{
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    const long pair = (long)blockIdx.x * 4 + (threadIdx.x >> 6);       // pair of lines (i even, i + 1) of one j
    if (pair >= npairs) return;
    const int j = (int)(pair % D.ny), i = 2 * (int)(pair / D.ny);
    d2_t v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        // 1 KiB per instruction: shipped layout = contiguous run of line i + (q >> 2); bricks = 4 pieces of 256 bytes
        long a;
        if (BI == 0) a = addr<0>(D, i + (q >> 2), j, (q & 3) * 128 + lane * 2);
        else { const int piece = q * 4 + (lane >> 4), w = lane & 15; a = addr<BI>(D, i + (w >> 3), j, piece * 16 + (w & 7) * 2); }
        v[q] = *reinterpret_cast<const d2_t *>(in + a);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        long a;
        if (BI == 0) a = addr<0>(D, i + (q >> 2), j, (q & 3) * 128 + lane * 2);
        else { const int piece = q * 4 + (lane >> 4), w = lane & 15; a = addr<BI>(D, i + (w >> 3), j, piece * 16 + (w & 7) * 2); }
        __builtin_nontemporal_store(v[q] * 1.000000001, reinterpret_cast<d2_t *>(out + a));
    }
}

template <class F>
static int timeit(const char *name, F launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int it = 0; it < 14; ++it) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 4) ts.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    const double med = ts[ts.size() / 2];
    printf("%-78s  min %.3f  med %.3f ms   %.2f TB/s on 16 B/cell\n", name, ts[0], med, 16.0 * 134217728.0 / (med * 1e-3) / 1e12);
    fflush(stdout);
    return 0;
}

int main()
{
    const int n = 512;
    Dim D; D.n = n; D.ny = n; D.nz = n; D.sx = (long)n * n + 256;
    const size_t N = (size_t)n * D.sx + 4096;
    double *a, *b;
    CK(hipMalloc(&a, N * 8)); CK(hipMalloc(&b, N * 8));
    CK(hipMemset(a, 0, N * 8)); CK(hipMemset(b, 0, N * 8));
    const long t0 = (long)n * (n / 16);
#define L0(K, ...) [&] { hipLaunchKernelGGL((K<__VA_ARGS__>), dim3((unsigned)t0), dim3(512), 0, 0, a, b, D, t0); }
    if (timeit("axis-0 tile, padded planes (shipped), read + write", L0(k_axis0, 0, false))) return 1;
    if (timeit("axis-0 tile, padded planes (shipped), + j-neighbours  [fused pattern]", L0(k_axis0, 0, true))) return 1;
    if (timeit("axis-0 tile, bricks of 4 planes, read + write", L0(k_axis0, 4, false))) return 1;
    if (timeit("axis-0 tile, bricks of 4 planes, + j-neighbours  [fused pattern]", L0(k_axis0, 4, true))) return 1;
    if (timeit("axis-0 tile, bricks of 16 planes, read + write", L0(k_axis0, 16, false))) return 1;
    if (timeit("axis-0 tile, bricks of 16 planes, + j-neighbours  [fused pattern]", L0(k_axis0, 16, true))) return 1;
    if (timeit("axis-0 tile, bricks of 4, whole brick per instruction, read + write", [&] { hipLaunchKernelGGL((k_axis0_wholebrick<false>), dim3((unsigned)t0), dim3(512), 0, 0, a, b, D, t0); })) return 1;
    if (timeit("axis-0 tile, bricks of 4, whole brick per instruction, + j-neighbours", [&] { hipLaunchKernelGGL((k_axis0_wholebrick<true>), dim3((unsigned)t0), dim3(512), 0, 0, a, b, D, t0); })) return 1;
    const long t1 = (long)n * (n / 16), t1b = (long)(n / 4) * (n / 16);
    if (timeit("axis-1 tile, padded planes (shipped: 1 plane x 16 k, 256 threads)", [&] { hipLaunchKernelGGL((k_axis1<0, 1>), dim3((unsigned)t1), dim3(256), 0, 0, a, b, D, t1); })) return 1;
    if (timeit("axis-1 tile, bricks of 4 planes (4 planes x 16 k, 1024 threads)", [&] { hipLaunchKernelGGL((k_axis1<4, 4>), dim3((unsigned)t1b), dim3(1024), 0, 0, a, b, D, t1b); })) return 1;
    if (timeit("axis-1 tile, bricks of 4 planes (1 plane x 16 k, 256 threads)", [&] { hipLaunchKernelGGL((k_axis1<4, 1>), dim3((unsigned)t1), dim3(256), 0, 0, a, b, D, t1); })) return 1;
    if (timeit("axis-1 tile, bricks of 16 planes (4 planes x 16 k, 1024 threads)", [&] { hipLaunchKernelGGL((k_axis1<16, 4>), dim3((unsigned)t1b), dim3(1024), 0, 0, a, b, D, t1b); })) return 1;
    const long np = (long)(n / 2) * n;
    if (timeit("axis-2 lines, padded planes (shipped: 2 x 4 KiB contiguous per wave)", [&] { hipLaunchKernelGGL((k_axis2<0>), dim3((unsigned)((np + 3) / 4)), dim3(256), 0, 0, a, b, D, np); })) return 1;
    if (timeit("axis-2 lines, bricks of 4 planes (32 pieces of 256 bytes per wave)", [&] { hipLaunchKernelGGL((k_axis2<4>), dim3((unsigned)((np + 3) / 4)), dim3(256), 0, 0, a, b, D, np); })) return 1;
    return 0;
}
