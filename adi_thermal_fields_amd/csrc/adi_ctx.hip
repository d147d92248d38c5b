// adi_ctx.hip -- error reporting, device queries and the context API of libadi_hip.so
// (library-owned device memory, host arrays at the boundary).  See include/adi_hip.h.
#include <string.h>

#include <new>

#include "adi_common.hpp"

namespace adi {

char *err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int set_err(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace adi

using namespace adi;

struct adi_ctx {
    int nx, ny, nz, device;
    double dx;
    long sx;             // padded plane stride (elements)
    size_t N;            // allocated elements per field = nx * sx
    uint8_t *mask, *flags, *dir_mask;
    double *T[2];        // ping-pong state
    double *tmp[2];      // stage scratch
    double *coeff[3], *qflux[3], *dir_val;
    void *work;
    size_t work_bytes;
    int cur;             // index of the current state buffer
    int variant;         // ADI_SWEEP_* chosen at build_coeffs time
    bool have_mask, have_packs, have_T;
    bool all_solid;      // every cell in the mask: passed to adi_step as the box hint (bit 1 of `sparse`)
    int promise;         // no-fallback promise (bit 2 of `sparse`): -1 not known for this mask / these packs, 0 no, 1 yes
    double fconsts[12];  // per-face scalar coefficients / fluxes of the packs (adi_face_constants) ...
    bool fconsts_ok;     // ... valid for all three axes: passed to adi_step as h_face_consts
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    float last_ms;
};

static void ctx_free(adi_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    void *ptrs[] = {c->mask, c->flags, c->dir_mask, c->T[0], c->T[1], c->tmp[0], c->tmp[1], c->coeff[0], c->coeff[1],
                    c->coeff[2], c->qflux[0], c->qflux[1], c->qflux[2], c->dir_val, c->work};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// dense host array <-> padded-plane device array
static hipError_t copy_planes(adi_ctx *c, void *dst, const void *src, size_t elem, bool to_device)
{
    const size_t row = (size_t)c->ny * c->nz * elem;
    const size_t dpitch = to_device ? (size_t)c->sx * elem : row;
    const size_t spitch = to_device ? row : (size_t)c->sx * elem;
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, row, c->nx, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost,
                            c->stream);
}

extern "C" {

int adi_abi_version(void) { return ADI_ABI_VERSION; }

const char *adi_last_error(void) { return err_buf(); }

int adi_device_count(int *count)
{
    ADI_REQUIRE(count, "adi_device_count: null argument");
    ADI_HIP_TRY(hipGetDeviceCount(count));
    return ADI_OK;
}

int adi_device_info(int device, char *name, int *cu_count, size_t *hbm_bytes, size_t *lds_per_cu)
{
    hipDeviceProp_t p;
    ADI_HIP_TRY(hipGetDeviceProperties(&p, device));
    if (name) {
        snprintf(name, 256, "%s (%s)", p.name, p.gcnArchName);
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    if (lds_per_cu) *lds_per_cu = p.maxSharedMemoryPerMultiProcessor;
    return ADI_OK;
}

int adi_ctx_create(int nx, int ny, int nz, double dx, int device, adi_ctx **out)
{
    ADI_REQUIRE(out, "adi_ctx_create: null output");
    ADI_REQUIRE(nx > 0 && ny > 0 && nz > 0 && dx > 0.0, "adi_ctx_create: bad grid %d x %d x %d, dx=%g", nx, ny, nz, dx);
    ADI_HIP_TRY(hipSetDevice(device));
    adi_ctx *c = new (std::nothrow) adi_ctx();
    if (!c) return set_err(ADI_ERR_HIP, "adi_ctx_create: out of host memory");
    memset(c, 0, sizeof(*c));
    c->nx = nx; c->ny = ny; c->nz = nz; c->dx = dx; c->device = device;
    c->sx = adi_recommended_plane_stride(ny, nz);
    c->N = (size_t)nx * c->sx;
    const size_t fb = c->N * sizeof(double);
#define CTX_ALLOC(ptr, bytes)                                                          \
    do {                                                                               \
        hipError_t e_ = hipMalloc((void **)&(ptr), (bytes));                           \
        if (e_ != hipSuccess) {                                                        \
            ctx_free(c);                                                               \
            return set_err(ADI_ERR_HIP, "hipMalloc(%zu) failed: %s", (size_t)(bytes), hipGetErrorString(e_)); \
        }                                                                              \
    } while (0)
    CTX_ALLOC(c->mask, c->N);
    CTX_ALLOC(c->flags, c->N);
    CTX_ALLOC(c->dir_mask, c->N);
    CTX_ALLOC(c->dir_val, fb);
    for (int i = 0; i < 2; ++i) { CTX_ALLOC(c->T[i], fb); CTX_ALLOC(c->tmp[i], fb); }
    for (int a = 0; a < 3; ++a) { CTX_ALLOC(c->coeff[a], fb); CTX_ALLOC(c->qflux[a], fb); }
    size_t wb = 0;
    for (int a = 0; a < 3; ++a) {
        size_t b = 0;
        adi_sweep_workspace_bytes(a, nx, ny, nz, c->sx, &b);
        if (b > wb) wb = b;
    }
    if (wb) CTX_ALLOC(c->work, wb);
    c->work_bytes = wb;
#undef CTX_ALLOC
    if (hipStreamCreate(&c->stream) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess ||
        hipEventCreate(&c->ev1) != hipSuccess) {
        ctx_free(c);
        return set_err(ADI_ERR_HIP, "adi_ctx_create: stream/event creation failed");
    }
    *out = c;
    return ADI_OK;
}

int adi_ctx_destroy(adi_ctx *ctx)
{
    ctx_free(ctx);
    return ADI_OK;
}

int adi_ctx_set_mask(adi_ctx *c, const uint8_t *h_mask)
{
    ADI_REQUIRE(c && h_mask, "adi_ctx_set_mask: null argument");
    ADI_HIP_TRY(hipSetDevice(c->device));
    ADI_HIP_TRY(hipMemsetAsync(c->mask, 0, c->N, c->stream));
    ADI_HIP_TRY(copy_planes(c, c->mask, h_mask, 1, true));
    int rc = adi_build_nbr_flags(c->mask, c->nx, c->ny, c->nz, c->sx, c->flags, c->stream);
    if (rc != ADI_OK) return rc;
    ADI_HIP_TRY(hipStreamSynchronize(c->stream));
    bool all = true;
    const size_t cells = (size_t)c->nx * c->ny * c->nz;
    for (size_t i = 0; i < cells && all; ++i) all = h_mask[i] != 0;
    c->all_solid = all;
    c->have_mask = true;
    c->promise = -1;
    c->have_packs = false;  // packs depend on the mask: rebuild before the next step
    return ADI_OK;
}

int adi_ctx_build_coeffs(adi_ctx *c, double rho, double cp, const int *h_mode, const double *h_scalar,
                         const double *const *h_h_field, const int *q_mode, const double *q_scalar,
                         const double *const *h_q_field, const uint8_t *h_dir_mask, const double *h_dir_val)
{
    ADI_REQUIRE(c && h_mode && h_scalar && q_mode && q_scalar, "adi_ctx_build_coeffs: null argument");
    if (!c->have_mask) return set_err(ADI_ERR_STATE, "adi_ctx_build_coeffs: set the mask first");
    ADI_HIP_TRY(hipSetDevice(c->device));
    const size_t fb = c->N * sizeof(double);
    // per-voxel h/q fields are staged through the two stage-scratch buffers one face pair at a time;
    // to keep this simple and exact we upload each field into a temporary device buffer
    const double *dh[6] = {0}, *dq[6] = {0};
    double *staged[12] = {0};
    int ns = 0, rc = ADI_OK;
    auto stage = [&](const double *host) -> const double * {
        double *d = nullptr;
        if (hipMalloc((void **)&d, fb) != hipSuccess) return nullptr;
        staged[ns++] = d;
        if (copy_planes(c, d, host, sizeof(double), true) != hipSuccess) return nullptr;
        return d;
    };
    for (int f = 0; f < 6 && rc == ADI_OK; ++f) {
        if (h_mode[f] == ADI_FACE_FIELD) {
            if (!h_h_field || !h_h_field[f]) rc = set_err(ADI_ERR_ARG, "adi_ctx_build_coeffs: missing h field %d", f);
            else if (!(dh[f] = stage(h_h_field[f]))) rc = set_err(ADI_ERR_HIP, "adi_ctx_build_coeffs: staging failed");
        }
        if (rc == ADI_OK && q_mode[f] == ADI_FACE_FIELD) {
            if (!h_q_field || !h_q_field[f]) rc = set_err(ADI_ERR_ARG, "adi_ctx_build_coeffs: missing q field %d", f);
            else if (!(dq[f] = stage(h_q_field[f]))) rc = set_err(ADI_ERR_HIP, "adi_ctx_build_coeffs: staging failed");
        }
    }
    if (rc == ADI_OK)
        rc = adi_build_coeffs(c->mask, c->nx, c->ny, c->nz, c->sx, c->dx, rho, cp, h_mode, h_scalar, dh, q_mode, q_scalar, dq,
                              c->coeff, c->qflux, c->stream);
    if (rc == ADI_OK) {
        int valid[3];
        rc = adi_face_constants(c->dx, rho, cp, h_mode, h_scalar, q_mode, q_scalar, c->fconsts, valid);
        c->fconsts_ok = rc == ADI_OK && valid[0] && valid[1] && valid[2];
    }
    bool has_dir = false, has_q = false;
    if (rc == ADI_OK) {
        for (int f = 0; f < 6; ++f) has_q = has_q || (q_mode[f] != ADI_FACE_NONE);
        if (h_dir_mask) {
            const size_t nd = (size_t)c->nx * c->ny * c->nz;
            for (size_t p = 0; p < nd && !has_dir; ++p) has_dir = h_dir_mask[p] != 0;
        }
        if (has_dir) {
            if (hipMemsetAsync(c->dir_mask, 0, c->N, c->stream) != hipSuccess ||
                copy_planes(c, c->dir_mask, h_dir_mask, 1, true) != hipSuccess)
                rc = set_err(ADI_ERR_HIP, "adi_ctx_build_coeffs: dir_mask upload failed");
            if (rc == ADI_OK) {
                hipError_t e = h_dir_val ? copy_planes(c, c->dir_val, h_dir_val, sizeof(double), true)
                                         : hipMemsetAsync(c->dir_val, 0, fb, c->stream);  // dir_value None -> 0 (:75-76)
                if (e != hipSuccess) rc = set_err(ADI_ERR_HIP, "adi_ctx_build_coeffs: dir_val upload failed");
            }
        }
    }
    hipError_t es = hipStreamSynchronize(c->stream);
    for (int i = 0; i < ns; ++i) (void)hipFree(staged[i]);
    if (rc != ADI_OK) return rc;
    if (es != hipSuccess) return set_err(ADI_ERR_HIP, "adi_ctx_build_coeffs: %s", hipGetErrorString(es));
    c->variant = has_dir ? (has_q ? ADI_SWEEP_GENERAL : ADI_SWEEP_NO_Q) : (has_q ? ADI_SWEEP_NO_DIR : ADI_SWEEP_LEAN);
    c->have_packs = true;
    c->promise = -1;
    return ADI_OK;
}

int adi_ctx_upload_T(adi_ctx *c, const double *h_T)
{
    ADI_REQUIRE(c && h_T, "adi_ctx_upload_T: null argument");
    ADI_HIP_TRY(hipSetDevice(c->device));
    ADI_HIP_TRY(copy_planes(c, c->T[c->cur], h_T, sizeof(double), true));
    ADI_HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_T = true;
    return ADI_OK;
}

int adi_ctx_download_T(adi_ctx *c, double *h_T)
{
    ADI_REQUIRE(c && h_T, "adi_ctx_download_T: null argument");
    if (!c->have_T) return set_err(ADI_ERR_STATE, "adi_ctx_download_T: no field uploaded");
    ADI_HIP_TRY(hipSetDevice(c->device));
    ADI_HIP_TRY(copy_planes(c, h_T, c->T[c->cur], sizeof(double), false));
    ADI_HIP_TRY(hipStreamSynchronize(c->stream));
    return ADI_OK;
}

int adi_ctx_download_pack(adi_ctx *c, int axis, double *h_coeff, double *h_qflux)
{
    ADI_REQUIRE(c && axis >= 0 && axis < 3, "adi_ctx_download_pack: bad argument");
    if (!c->have_packs) return set_err(ADI_ERR_STATE, "adi_ctx_download_pack: packs not built");
    ADI_HIP_TRY(hipSetDevice(c->device));
    if (h_coeff) ADI_HIP_TRY(copy_planes(c, h_coeff, c->coeff[axis], sizeof(double), false));
    if (h_qflux) ADI_HIP_TRY(copy_planes(c, h_qflux, c->qflux[axis], sizeof(double), false));
    ADI_HIP_TRY(hipStreamSynchronize(c->stream));
    return ADI_OK;
}

int adi_ctx_step(adi_ctx *c, double rho, double cp, double k, double dt, double theta, double Tinf, int nsteps)
{
    ADI_REQUIRE(c && nsteps >= 0, "adi_ctx_step: bad argument");
    if (!c->have_mask || !c->have_packs || !c->have_T)
        return set_err(ADI_ERR_STATE, "adi_ctx_step: needs mask, packs (rebuilt after every mask change) and a field");
    ADI_HIP_TRY(hipSetDevice(c->device));
    ADI_HIP_TRY(hipEventRecord(c->ev0, c->stream));
    for (int s = 0; s < nsteps; ++s) {
        const int nxt = c->cur ^ 1;
        const int sp = 1 | (c->all_solid ? 2 : 0);
        int rc;
        if (c->promise < 0 && c->work != nullptr && c->work_bytes >= sizeof(unsigned)) {
            // first step after a mask / pack change: count the units each sweep's FAST kernel queues (a function of the mask
            // and the packs, not of the field); if there are none, the later steps skip the queue reset and the fallback launch
            unsigned q[3];
            rc = adi_step_queued(c->T[c->cur], c->T[nxt], c->tmp[0], c->tmp[1], c->flags, c->coeff, c->dir_mask, c->dir_val,
                                 c->qflux, c->variant, sp, c->nx, c->ny, c->nz, c->sx, c->dx, rho, cp, k, dt, theta, Tinf,
                                 c->fconsts_ok ? c->fconsts : nullptr, c->work, c->work_bytes, c->stream, q);
            if (rc != ADI_OK) return rc;
            ADI_HIP_TRY(hipStreamSynchronize(c->stream));
            c->promise = (q[0] == 0 && q[1] == 0 && q[2] == 0) ? 1 : 0;
        } else {
            rc = adi_step(c->T[c->cur], c->T[nxt], c->tmp[0], c->tmp[1], c->flags, c->coeff, c->dir_mask, c->dir_val, c->qflux,
                          c->variant, sp | (c->promise == 1 ? 4 : 0), c->nx, c->ny, c->nz, c->sx, c->dx, rho, cp, k, dt, theta, Tinf,
                          c->fconsts_ok ? c->fconsts : nullptr, c->work, c->work_bytes, c->stream);
            if (rc != ADI_OK) return rc;
        }
        c->cur = nxt;
    }
    ADI_HIP_TRY(hipEventRecord(c->ev1, c->stream));
    ADI_HIP_TRY(hipStreamSynchronize(c->stream));
    ADI_HIP_TRY(hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
    return ADI_OK;
}

int adi_copy_planes(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t plane_bytes, size_t nplanes,
                    int to_device, void *stream)
{
    ADI_REQUIRE(dst && src, "adi_copy_planes: null argument");
    ADI_REQUIRE(dst_pitch >= plane_bytes && src_pitch >= plane_bytes, "adi_copy_planes: pitch smaller than a plane");
    if (plane_bytes == 0 || nplanes == 0) return ADI_OK;
    const hipMemcpyKind kind = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
    if (dst_pitch == plane_bytes && src_pitch == plane_bytes)
        ADI_HIP_TRY(hipMemcpyAsync(dst, src, plane_bytes * nplanes, kind, as_stream(stream)));
    else
        ADI_HIP_TRY(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, plane_bytes, nplanes, kind, as_stream(stream)));
    return ADI_OK;
}

int adi_ctx_last_step_ms(adi_ctx *c, float *ms)
{
    ADI_REQUIRE(c && ms, "adi_ctx_last_step_ms: null argument");
    *ms = c->last_ms;
    return ADI_OK;
}

}  // extern "C"
