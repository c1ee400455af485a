// stochastic.h — the stochastic pieces next to the velocity-Verlet path: the fused Langevin "BAOA" update
// (simulate!(::Langevin) simulators.jl:1099-1220 with langevin_o_step_kernel! kernels.jl:726-741), the Andersen thermostat
// (apply_andersen_coupling_kernel! kernels.jl:706-723, coupling.jl:196-211) and Maxwell-Boltzmann velocities
// (random_velocities_kernel! kernels.jl:688-704).  All noise is counter based: Philox4x32-10 (Salmon et al., SC'11) keyed by
// (key, ctr1) with the 1-based ORIGINAL atom index as ctr0, so a run does not depend on the Hilbert order of the moment or on the
// launch shape.  The reference takes philox4x32_10 / randn_f32 / randn_f64 from PhiloxRNG.jl (compat "1", src/Molly.jl:26), which
// is not vendored under the reference tree: the word order of the counter and the uniform → normal transform below are this
// library's own (Box-Muller on the open interval) and are pinned by the Random123 known-answer vectors only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "common.h"

namespace mhip {

template <class T> struct StochP {
    T dt, dt_half;            // Langevin: full kick, two half drifts
    T vel_scale;              // exp(−γ dt)                                        simulators.jl:1091
    double noise_kt;          // Langevin: sqrt(1 − vel_scale²)·sqrt(kT); Andersen / random velocities: sqrt(kT)
    uint64_t key, ctr1, natoms;
    uint64_t prob_u64;        // Andersen: round(clamp(dt/τ, 0, prevfloat(1))·2⁶⁴)   coupling.jl:203-204
};

// one fused update of all owned atoms; cm_out (nullable) receives 4 doubles per block (Σ m v, Σ m) of the new velocities
template <class T>
void launch_langevin(hipStream_t s, int n_blocks, int64_t n, typename Vec<T>::T4* pos, typename Vec<T>::T4* vel, const typename Vec<T>::T4* frc,
                     const int32_t* orig, const StochP<T>& P, const T* vcm, const double* cm_in, int n_cm_in, double* cm_out, const GridP<T>& G,
                     const typename Vec<T>::T4* frc_add = nullptr);      // frc_add (nullable): a second force array added on the way (a small system's bonded sums)
// mode 0: Andersen re-draws (probability prob_u64 / 2⁶⁴ per atom); mode 1: every atom gets a Maxwell-Boltzmann velocity
template <class T>
void launch_redraw(hipStream_t s, int mode, int64_t n, typename Vec<T>::T4* vel, const int32_t* orig, const StochP<T>& P,
                   const T* vcm, const double* cm_in, int n_cm_in);
// test hook: the raw generator, out[4] = philox4x32_10(ctr, key) on the device
// the same generator on the host: per-step (key, ctr1) pairs of a thermostat are drawn from a stream keyed by the user's seed
void philox_host(uint64_t ctr0, uint64_t ctr1, uint64_t key, uint32_t* out4);
void launch_philox_probe(hipStream_t s, const uint32_t* ctr4_key2_dev, uint32_t* out4_dev);

}  // namespace mhip
