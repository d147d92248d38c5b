// adi_morph.hip -- the callers either side of the hot path, on the device:
//   * 6-connectivity voxel morphology of the mask pipeline: dilate6 / erode6 (waam_from_stl_v7_mm.py:73-96) and the
//     flood fill of the outside air (:106-134, with the padding the reference's comment intends: D8 in DESIGN.md);
//   * frame packing for output: fp64 field (nx, ny, nz) in the padded-plane layout -> big-endian float32 in VTK point
//     order (x fastest), what a legacy-VTK BINARY file holds (the reference writes ASCII cell by cell in Python,
//     vtk_writer.py:4-30, waam_from_stl_v7_mm.py:191-216).
// Masks here are dense uint8 (nx, ny, nz), C order.  Byte work, HBM/L2-bound; nothing here is on the step's path.
#include "adi_common.hpp"

namespace adi {

template <bool ERODE>
__global__ __launch_bounds__(256) void k_morph6(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int nx, int ny,
                                                int nz)
{
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long plane = (long)ny * nz, n = plane * nx;
    if (q >= n) return;
    const int i = (int)(q / plane);
    const long r = q - (long)i * plane;
    const int j = (int)(r / nz), k = (int)(r - (long)j * nz);
    const bool c = in[q] != 0;
    if (ERODE) {
        // b[1:-1,1:-1,1:-1] = a & its six neighbours; the box boundary stays 0 (:84-96)
        bool v = false;
        if (i > 0 && i < nx - 1 && j > 0 && j < ny - 1 && k > 0 && k < nz - 1)
            v = c && in[q - plane] && in[q + plane] && in[q - nz] && in[q + nz] && in[q - 1] && in[q + 1];
        out[q] = v ? 1 : 0;
    } else {
        bool v = c;                                            // b = a | shifted copies, inside the box (:73-82)
        if (i > 0) v = v || in[q - plane];
        if (i < nx - 1) v = v || in[q + plane];
        if (j > 0) v = v || in[q - nz];
        if (j < ny - 1) v = v || in[q + nz];
        if (k > 0) v = v || in[q - 1];
        if (k < nz - 1) v = v || in[q + 1];
        out[q] = v ? 1 : 0;
    }
}

// seeds of the flood fill: air cells on the boundary of the box (they touch the padding layer, which is air)
__global__ __launch_bounds__(256) void k_flood_seed(const uint8_t *__restrict__ solid, uint8_t *__restrict__ outside,
                                                    int nx, int ny, int nz)
{
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long plane = (long)ny * nz, n = plane * nx;
    if (q >= n) return;
    const int i = (int)(q / plane);
    const long r = q - (long)i * plane;
    const int j = (int)(r / nz), k = (int)(r - (long)j * nz);
    const bool edge = i == 0 || i == nx - 1 || j == 0 || j == ny - 1 || k == 0 || k == nz - 1;
    outside[q] = (edge && solid[q] == 0) ? 1 : 0;
}

// one thread per grid line along `axis`: a forward and a backward scan carry "outside" through runs of air cells, so a
// round of three sweeps moves the front across whole lines instead of one cell per pass (the reference dilates once per
// iteration, up to nx+ny+nz+10 iterations); the fixed point -- air connected to the boundary -- is the same set.
__global__ __launch_bounds__(256) void k_flood_sweep(const uint8_t *__restrict__ solid, uint8_t *__restrict__ outside,
                                                     int nx, int ny, int nz, int axis, int *__restrict__ changed)
{
    const long lid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long plane = (long)ny * nz;
    long base, stride, nlines;
    int n;
    if (axis == 0) { nlines = plane; n = nx; stride = plane; base = lid; }
    else if (axis == 1) { nlines = (long)nx * nz; n = ny; stride = nz; base = (lid / nz) * plane + (lid % nz); }
    else { nlines = (long)nx * ny; n = nz; stride = 1; base = lid * nz; }
    if (lid >= nlines) return;
    bool any = false, carry = false;
    for (int r = 0; r < n; ++r) {
        const long p = base + (long)r * stride;
        const bool o = outside[p] != 0;
        const bool v = o || (carry && solid[p] == 0);
        if (v && !o) { outside[p] = 1; any = true; }
        carry = v;
    }
    carry = false;
    for (int r = n - 1; r >= 0; --r) {
        const long p = base + (long)r * stride;
        const bool o = outside[p] != 0;
        const bool v = o || (carry && solid[p] == 0);
        if (v && !o) { outside[p] = 1; any = true; }
        carry = v;
    }
    if (any) *changed = 1;
}

// fp64 (nx, ny, nz) with plane stride sx -> float32 big-endian, element (i, j, k) at (k*ny + j)*nx + i.
// 32 x 32 (i, k) tiles through LDS: reads run along k, writes along i, both coalesced.
__global__ __launch_bounds__(256) void k_pack_frame_f32be(const double *__restrict__ T, int nx, int ny, int nz, long sx,
                                                          unsigned *__restrict__ out)
{
    __shared__ float tile[32][33];
    const int j = blockIdx.z;
    const int i0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int a = ty; a < 32; a += 8) {
        const int i = i0 + a, k = k0 + tx;
        if (i < nx && k < nz) tile[a][tx] = (float)T[(long)i * sx + (long)j * nz + k];
    }
    __syncthreads();
    for (int a = ty; a < 32; a += 8) {
        const int k = k0 + a, i = i0 + tx;
        if (i < nx && k < nz)
            out[((long)k * ny + j) * nx + i] = __builtin_bswap32(__float_as_uint(tile[tx][a]));
    }
}

}  // namespace adi

using namespace adi;

extern "C" {

int adi_morph6(int op, const uint8_t *d_in, uint8_t *d_out, int nx, int ny, int nz, void *stream)
{
    ADI_REQUIRE(op == 0 || op == 1, "adi_morph6: op must be 0 (dilate6) or 1 (erode6)");
    ADI_REQUIRE(d_in && d_out && d_in != d_out, "adi_morph6: null or aliased argument");
    ADI_REQUIRE(nx > 0 && ny > 0 && nz > 0, "adi_morph6: bad grid %d x %d x %d", nx, ny, nz);
    const long n = (long)nx * ny * nz;
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (op == 0) hipLaunchKernelGGL(k_morph6<false>, dim3(grid), dim3(256), 0, as_stream(stream), d_in, d_out, nx, ny, nz);
    else hipLaunchKernelGGL(k_morph6<true>, dim3(grid), dim3(256), 0, as_stream(stream), d_in, d_out, nx, ny, nz);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_flood_outside(const uint8_t *d_solid, uint8_t *d_outside, int nx, int ny, int nz, int *d_flag, int *rounds,
                      void *stream)
{
    ADI_REQUIRE(d_solid && d_outside && d_flag && d_solid != d_outside, "adi_flood_outside: null or aliased argument");
    ADI_REQUIRE(nx > 0 && ny > 0 && nz > 0, "adi_flood_outside: bad grid %d x %d x %d", nx, ny, nz);
    hipStream_t st = as_stream(stream);
    const long n = (long)nx * ny * nz;
    hipLaunchKernelGGL(k_flood_seed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_solid, d_outside, nx, ny, nz);
    const long nl[3] = {(long)ny * nz, (long)nx * nz, (long)nx * ny};
    int r = 0;
    const int max_rounds = nx + ny + nz + 10;                 // the reference's own bound (:123-124) is generous here too
    for (; r < max_rounds; ++r) {
        ADI_HIP_TRY(hipMemsetAsync(d_flag, 0, sizeof(int), st));
        for (int ax = 0; ax < 3; ++ax)
            hipLaunchKernelGGL(k_flood_sweep, dim3((unsigned)((nl[ax] + 255) / 256)), dim3(256), 0, st, d_solid, d_outside,
                               nx, ny, nz, ax, d_flag);
        int h = 0;
        ADI_HIP_TRY(hipMemcpyAsync(&h, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
        ADI_HIP_TRY(hipStreamSynchronize(st));
        if (!h) break;
    }
    if (rounds) *rounds = r + 1;
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

int adi_pack_frame_f32be(const double *d_T, int nx, int ny, int nz, long plane_stride, uint32_t *d_out, void *stream)
{
    ADI_REQUIRE(d_T && d_out, "adi_pack_frame_f32be: null argument");
    ADI_REQUIRE(nx > 0 && ny > 0 && nz > 0 && ny <= 65535, "adi_pack_frame_f32be: bad grid %d x %d x %d", nx, ny, nz);
    const long sx = plane_stride ? plane_stride : (long)ny * nz;
    ADI_REQUIRE(sx >= (long)ny * nz, "adi_pack_frame_f32be: plane_stride %ld < ny*nz", sx);
    hipLaunchKernelGGL(k_pack_frame_f32be, dim3((nz + 31) / 32, (nx + 31) / 32, ny), dim3(256), 0, as_stream(stream), d_T, nx,
                       ny, nz, sx, d_out);
    ADI_CHECK_LAUNCH();
    return ADI_OK;
}

}  // extern "C"
