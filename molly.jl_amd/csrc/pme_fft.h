// pme_fft.h — library 3-D real ↔ complex FFTs (hipFFT) for PME meshes too long for the direct-DFT passes of pme.h (more than 512 points on an axis):
// plan_fft! / plan_bfft! of the reference (ewald.jl:405-411), unnormalised forward e^{−2πi jk/n} and backward e^{+2πi jk/n}, on the half spectrum
// [x][y][kz <= nz/2] the rest of the reciprocal space uses.  The library's headers stay in pme_fft.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace mhip {

struct FftPlan3d {
    void* r2c = nullptr; void* c2r = nullptr; bool dbl = false;
    void create(int nx, int ny, int nz, bool double_precision);
    void destroy();
    void forward(hipStream_t s, void* real_in, void* complex_out);      // real [nx][ny][nz] → complex [nx][ny][nz/2 + 1]
    void backward(hipStream_t s, void* complex_in, void* real_out);     // Hermitian half → real (the complex input may be overwritten)
};

}  // namespace mhip
