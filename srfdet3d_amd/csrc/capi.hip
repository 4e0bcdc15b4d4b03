// capi.hip -- ABI bookkeeping: version, error strings, device probe.
#include "common.hpp"

extern "C" int srf_abi_version(void) { return 1; }

// 0 = production; 1 = developer build (-DSRF_DEV: timing-ablation kernels with wrong outputs by design, stamp hooks).  The
// Python loader refuses a library that answers 1 unless the caller asked for the developer library by name.
extern "C" int srf_build_flavour(void)
{
#ifdef SRF_DEV
    return 1;
#else
    return 0;
#endif
}

extern "C" const char *srf_error_string(int code)
{
    switch (code) {
    case SRF_OK:
        return "ok";
    case SRF_EINVAL:
        return "invalid argument";
    case SRF_EWORKSPACE:
        return "workspace too small";
    case SRF_EUNSUPPORTED:
        return "unsupported shape";
    default:
        if (code <= SRF_EHIP_BASE) return hipGetErrorString((hipError_t)(SRF_EHIP_BASE - code));
        return "unknown error";
    }
}

extern "C" int srf_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return SRF_EHIP_BASE - (int)e;
    return n;
}
