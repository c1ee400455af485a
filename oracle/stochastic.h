// TEST INFRASTRUCTURE — CPU restatement of the reference's counter-based noise and the integrator steps that use it.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Follows: randn_svec kernels.jl:664-686; random_velocities_kernel! :688-704; apply_andersen_coupling_kernel! :706-723 (host side
// coupling.jl:196-211); langevin_o_step! :726-757; simulate!(::Langevin) simulators.jl:1099-1220.
//
// PARITY UNPINNED for the uniform → normal transform and the counter word order: the reference imports philox4x32_10, randn_f32 and
// randn_f64 from PhiloxRNG.jl (compat "1", src/Molly.jl:26, Project.toml), which is not under /root/reference.  What IS pinned:
// Philox4x32-10 itself, against the Random123 known-answer vectors (tests/test_oracle_stochastic.py), and the deterministic parts
// of the steps (against velocity Verlet's pieces and the closed-form Ornstein-Uhlenbeck statistics).
#pragma once
#include <cmath>
#include <cstdint>

namespace orc_stoch {

// Philox4x32-10, Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3" (SC'11), Random123 philox.h
inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        if (round > 0) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }             // bumpkey between rounds
        const uint64_t m0 = uint64_t(0xD2511F53u) * c0, m1 = uint64_t(0xCD9E8D57u) * c2;
        const uint32_t hi0 = uint32_t(m0 >> 32), lo0 = uint32_t(m0), hi1 = uint32_t(m1 >> 32), lo1 = uint32_t(m1);
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// philox4x32_10(ctr0::UInt64, ctr1::UInt64, key::UInt64): low word first
inline void philox(uint64_t ctr0, uint64_t ctr1, uint64_t key, uint32_t out[4]) {
    const uint32_t c[4] = {uint32_t(ctr0), uint32_t(ctr0 >> 32), uint32_t(ctr1), uint32_t(ctr1 >> 32)}, k[2] = {uint32_t(key), uint32_t(key >> 32)};
    philox4x32_10(c, k, out);
}

template <class T> struct Normal;
template <> struct Normal<float> {     // randn_f32: four normals from one block (Box-Muller, open-interval uniforms (k + ½)/2²⁴)
    static void three(uint64_t ctr0, uint64_t ctr1, uint64_t key, uint64_t, float z[3]) {
        uint32_t w[4]; philox(ctr0, ctr1, key, w);
        float u[4]; for (int k = 0; k < 4; ++k) u[k] = (float(w[k] >> 8) + 0.5f) * 5.9604644775390625e-8f;
        const float r0 = std::sqrt(-2.0f * std::log(u[0])), r1 = std::sqrt(-2.0f * std::log(u[2]));
        const float a0 = 6.2831853071795864769f * u[1], a1 = 6.2831853071795864769f * u[3];
        z[0] = r0 * std::cos(a0); z[1] = r0 * std::sin(a0); z[2] = r1 * std::cos(a1);
    }
};
template <> struct Normal<double> {    // randn_f64: two normals per block; the third comes from the block at ctr0 + natoms (:683-685)
    static double unit(uint32_t lo, uint32_t hi) { return (double(((uint64_t(hi) << 32) | lo) >> 11) + 0.5) * 1.1102230246251565e-16; }
    static void three(uint64_t ctr0, uint64_t ctr1, uint64_t key, uint64_t natoms, double z[3]) {
        uint32_t w[4], w2[4]; philox(ctr0, ctr1, key, w); philox(ctr0 + natoms, ctr1, key, w2);
        const double r0 = std::sqrt(-2.0 * std::log(unit(w[0], w[1]))), a0 = 6.2831853071795864769 * unit(w[2], w[3]);
        const double r1 = std::sqrt(-2.0 * std::log(unit(w2[0], w2[1]))), a1 = 6.2831853071795864769 * unit(w2[2], w2[3]);
        z[0] = r0 * std::cos(a0); z[1] = r0 * std::sin(a0); z[2] = r1 * std::cos(a1);
    }
};

// sqrt(kT/m) factor of a draw, zero for massless atoms (simulators.jl:1145-1147); `pref` = sqrt(kT) (× noise_scale for Langevin)
template <class T> inline T thermal(double pref, T m) { return m == T(0) ? T(0) : T(pref * std::sqrt(1.0 / double(m))); }

// langevin_o_step! kernels.jl:743-757: v = muladd(vel_scale, v, noise·noise_scale_i), atom index 1-based in the counter
template <class T> void o_step(int64_t n, T* v, const T* m, T vel_scale, double pref, uint64_t ctr1, uint64_t key) {
    for (int64_t i = 0; i < n; ++i) {
        T z[3]; Normal<T>::three(uint64_t(i) + 1, ctr1, key, uint64_t(n), z);
        const T ns = thermal<T>(pref, m[i]);
        for (int d = 0; d < 3; ++d) v[3 * i + d] = std::fma(vel_scale, v[3 * i + d], z[d] * ns);
    }
}
// mode 0: apply_andersen_coupling_kernel! (:706-723) with prob_u64 = round(UInt64, clamp(prob, 0, prevfloat(1))·2⁶⁴) (coupling.jl:203-204)
// mode 1: random_velocities_kernel! (:688-704)
template <class T> void redraw(int mode, int64_t n, T* v, const T* m, double kT, double prob, uint64_t ctr1, uint64_t key) {
    const double pc = std::fmin(std::fmax(prob, 0.0), std::nextafter(1.0, 0.0));
    const uint64_t prob_u64 = uint64_t(std::nearbyint(std::ldexp(pc, 64)));
    for (int64_t i = 0; i < n; ++i) {
        uint64_t ctr0 = uint64_t(i) + 1;
        if (mode == 0) {
            uint32_t w[4]; philox(ctr0, ctr1, key, w);
            if (!(((uint64_t(w[1]) << 32) | w[0]) < prob_u64)) continue;
            ctr0 += uint64_t(n);
        }
        T z[3]; Normal<T>::three(ctr0, ctr1, key, uint64_t(n), z);
        const T sc = thermal<T>(std::sqrt(kT), m[i]);
        for (int d = 0; d < 3; ++d) v[3 * i + d] = z[d] * sc;
    }
}

}  // namespace orc_stoch
