// C-ABI plumbing: version, thread-local error string, host-pointer convenience entry points.
#include "zm_common.h"

#include <cstring>

namespace zm {

char* last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// RAII device staging buffer for the *_host_* entry points.
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
};

}  // namespace zm

extern "C" int zm_version(void) { return 100; }

extern "C" const char* zm_last_error(void) { return zm::last_error_buf(); }

extern "C" int zm_lqr_backward_host_f64(const double* A, const double* B, const double* Q, const double* R, double* L,
                                        int64_t batch, int T, int n, int m) {
    if (batch == 0) return ZM_OK;   /* empty batch: nothing to do (pointers of empty arrays may be NULL) */
    if (!A || !B || !Q || !R || !L) return zm::set_error(ZM_EINVAL, "zm_lqr_backward_host_f64: null pointer");
    if (batch < 0 || T < 1 || n < 1 || m < 1) return zm::set_error(ZM_EINVAL, "zm_lqr_backward_host_f64: bad size");
    if (!zm_lqr_backward_supported(n, m, 8))
        return zm::set_error(ZM_EUNSUPPORTED, "zm_lqr_backward_host_f64: (n=%d, m=%d) not covered", n, m);
    if (batch == 0) return ZM_OK;
    const size_t steps = (size_t)batch * T;
    const size_t bA = steps * n * n * sizeof(double), bB = steps * n * m * sizeof(double);
    const size_t bR = steps * m * m * sizeof(double), bL = bB;
    zm::DevBuf dA, dB, dQ, dR, dL;
    ZM_HIP_CHECK(dA.alloc(bA));
    ZM_HIP_CHECK(dB.alloc(bB));
    ZM_HIP_CHECK(dQ.alloc(bA));
    ZM_HIP_CHECK(dR.alloc(bR));
    ZM_HIP_CHECK(dL.alloc(bL));
    ZM_HIP_CHECK(hipMemcpy(dA.p, A, bA, hipMemcpyHostToDevice));
    ZM_HIP_CHECK(hipMemcpy(dB.p, B, bB, hipMemcpyHostToDevice));
    ZM_HIP_CHECK(hipMemcpy(dQ.p, Q, bA, hipMemcpyHostToDevice));
    ZM_HIP_CHECK(hipMemcpy(dR.p, R, bR, hipMemcpyHostToDevice));
    int rc = zm_lqr_backward_f64((const double*)dA.p, (const double*)dB.p, (const double*)dQ.p, (const double*)dR.p,
                                 (double*)dL.p, batch, T, n, m, nullptr);
    if (rc != ZM_OK) return rc;
    ZM_HIP_CHECK(hipDeviceSynchronize());
    ZM_HIP_CHECK(hipMemcpy(L, dL.p, bL, hipMemcpyDeviceToHost));
    return ZM_OK;
}
