// pme_fft.hip — see pme_fft.h
#include "pme_fft.h"

#include <hipfft/hipfft.h>

#include <string>

#include "common.h"
#include "../../include/mollyhip.h"

namespace mhip {

static void fft_check(hipfftResult r, const char* what) {
    if (r != HIPFFT_SUCCESS) throw ApiError{MHIP_ERR_HIP, std::string("hipFFT: ") + what + " failed (code " + std::to_string((int)r) + ")"};
}

void FftPlan3d::create(int nx, int ny, int nz, bool double_precision) {
    destroy();
    dbl = double_precision;
    // each handle is stored as soon as it exists, so that a failure of the second plan (work areas of a 2^30-point mesh run to gigabytes)
    // leaves nothing behind that destroy() cannot see
    hipfftHandle a, b;
    fft_check(hipfftPlan3d(&a, nx, ny, nz, dbl ? HIPFFT_D2Z : HIPFFT_R2C), "plan (real to complex)");
    r2c = (void*)a;
    const hipfftResult rb = hipfftPlan3d(&b, nx, ny, nz, dbl ? HIPFFT_Z2D : HIPFFT_C2R);
    if (rb != HIPFFT_SUCCESS) { destroy(); fft_check(rb, "plan (complex to real)"); }
    c2r = (void*)b;
}

void FftPlan3d::destroy() {
    if (r2c) (void)hipfftDestroy((hipfftHandle)r2c);
    if (c2r) (void)hipfftDestroy((hipfftHandle)c2r);
    r2c = c2r = nullptr;
}

void FftPlan3d::forward(hipStream_t s, void* real_in, void* complex_out) {
    fft_check(hipfftSetStream((hipfftHandle)r2c, s), "set stream");
    if (dbl) fft_check(hipfftExecD2Z((hipfftHandle)r2c, (hipfftDoubleReal*)real_in, (hipfftDoubleComplex*)complex_out), "forward transform");
    else fft_check(hipfftExecR2C((hipfftHandle)r2c, (hipfftReal*)real_in, (hipfftComplex*)complex_out), "forward transform");
}

void FftPlan3d::backward(hipStream_t s, void* complex_in, void* real_out) {
    fft_check(hipfftSetStream((hipfftHandle)c2r, s), "set stream");
    if (dbl) fft_check(hipfftExecZ2D((hipfftHandle)c2r, (hipfftDoubleComplex*)complex_in, (hipfftDoubleReal*)real_out), "backward transform");
    else fft_check(hipfftExecC2R((hipfftHandle)c2r, (hipfftComplex*)complex_in, (hipfftReal*)real_out), "backward transform");
}

}  // namespace mhip
