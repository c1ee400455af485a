// physics.h — device-side pair physics: LennardJones (+6 cutoff strategies), Coulomb, CoulombReactionField,
// CoulombEwald, and the exact compare/select minimum image.  Behavioural spec (Molly.jl v0.23.3):
//   lennard_jones.jl:79-140, coulomb.jl:71-120, 748-814, 1384-1441, cutoffs.jl:15-253, mixing.jl:3-34,
//   spatial.jl:491-519, 573-586.
// Written for gfx950: fp32 uses the hardware v_rcp/v_sqrt/v_exp units (≤1 ulp) instead of the
// multi-instruction IEEE sequences; fp64 uses the precise library forms.
#pragma once
#include "common.h"

namespace mhip {

template <class T> struct M;
template <> struct M<float> {
    __device__ static inline float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    __device__ static inline float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
    __device__ static inline float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
    __device__ static inline float exp(float x) { return __expf(x); }
    __device__ static inline float erfc(float x) { return ::erfcf(x); }
    __device__ static inline float erf(float x) { return ::erff(x); }
    __device__ static inline float rint(float x) { return ::rintf(x); }
    __device__ static inline float floor(float x) { return ::floorf(x); }
    __device__ static inline float fabs(float x) { return ::fabsf(x); }
    __device__ static inline float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    // individually rounded operations: hipcc's default -ffp-contract=fast would otherwise fuse a*b+c into one
    // FMA (HIP's __fmul_rn/__fadd_rn are plain operators on AMD and do NOT prevent that)
    __device__ static inline float mul(float a, float b) {
#pragma clang fp contract(off)
        return a * b;
    }
    __device__ static inline float add(float a, float b) {
#pragma clang fp contract(off)
        return a + b;
    }
    __device__ static inline float sub(float a, float b) {
#pragma clang fp contract(off)
        return a - b;
    }
};
template <> struct M<double> {
    __device__ static inline double rcp(double x) { return 1.0 / x; }
    __device__ static inline double sqrt(double x) { return ::sqrt(x); }
    __device__ static inline double rsq(double x) { return 1.0 / ::sqrt(x); }
    __device__ static inline double exp(double x) { return ::exp(x); }
    __device__ static inline double erfc(double x) { return ::erfc(x); }
    __device__ static inline double erf(double x) { return ::erf(x); }
    __device__ static inline double rint(double x) { return ::rint(x); }
    __device__ static inline double floor(double x) { return ::floor(x); }
    __device__ static inline double fabs(double x) { return ::fabs(x); }
    __device__ static inline double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    __device__ static inline double mul(double a, double b) {
#pragma clang fp contract(off)
        return a * b;
    }
    __device__ static inline double add(double a, double b) {
#pragma clang fp contract(off)
        return a + b;
    }
    __device__ static inline double sub(double a, double b) {
#pragma clang fp contract(off)
        return a - b;
    }
};

// spatial.jl:491-500 vector_1D in its literal compare/select form.  With t = L - |v| both branches of the
// reference collapse to: |v| < t ? v : copysign(t, -v)   (v + L == L - (-v) and -(v - L) == L - v exactly).
// Non-contracted arithmetic: this is the form the bit-exact neighbour test is built on.
template <class T> __device__ inline T vector_1d_exact(T c1, T c2, T L) {
    T v = M<T>::sub(c2, c1);
    T a = M<T>::fabs(v);
    T t = M<T>::sub(L, a);
    return (a < t) ? v : ((v > T(0)) ? -t : t);
}
// r2 = sum(abs2, dr) = (dx² + dy²) + dz², every operation rounded separately (neighbors.jl:409)
template <class T> __device__ inline T norm2_exact(T dx, T dy, T dz) {
    return M<T>::add(M<T>::add(M<T>::mul(dx, dx), M<T>::mul(dy, dy)), M<T>::mul(dz, dz));
}
// spatial.jl:573-579 wrap_coord_1D
template <class T> __device__ inline T wrap_1d(T c, T L) {
    return M<T>::sub(c, M<T>::mul(M<T>::floor(c / L), L));
}

// ---- TriclinicBoundary ------------------------------------------------------------------------------
// vector(c1, c2, ::TriclinicBoundary) spatial.jl:528-551, every operation rounded on its own as Julia does
template <class T> __device__ inline void tri_min_image(T ax, T ay, T az, T bx, T by, T bz, const GridP<T>& G, T& dx, T& dy, T& dz) {
    using Mt = M<T>;
    if (G.triclinic == 1) {   // approx_images = true
        dx = Mt::sub(bx, ax); dy = Mt::sub(by, ay); dz = Mt::sub(bz, az);
        T k = Mt::floor(Mt::add(Mt::mul(dz, G.rs[2]), T(0.5)));
        dx = Mt::sub(dx, Mt::mul(G.bv[2][0], k)); dy = Mt::sub(dy, Mt::mul(G.bv[2][1], k)); dz = Mt::sub(dz, Mt::mul(G.bv[2][2], k));
        k = Mt::floor(Mt::add(Mt::mul(dy, G.rs[1]), T(0.5)));
        dx = Mt::sub(dx, Mt::mul(G.bv[1][0], k)); dy = Mt::sub(dy, Mt::mul(G.bv[1][1], k)); dz = Mt::sub(dz, Mt::mul(G.bv[1][2], k));
        k = Mt::floor(Mt::add(Mt::mul(dx, G.rs[0]), T(0.5)));
        dx = Mt::sub(dx, Mt::mul(G.bv[0][0], k)); dy = Mt::sub(dy, Mt::mul(G.bv[0][1], k)); dz = Mt::sub(dz, Mt::mul(G.bv[0][2], k));
        return;
    }
    T best = T(3.0e38) * T(3.0e38);   // typemax: +inf
    dx = T(0); dy = T(0); dz = T(0);
    for (int ox = -1; ox <= 1; ++ox) for (int oy = -1; oy <= 1; ++oy) for (int oz = -1; oz <= 1; ++oz) {
        T c[3] = {bx, by, bz}, e[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {   // ((c2 + ox·bv1) + oy·bv2) + oz·bv3, then − c1
            c[d] = Mt::add(Mt::add(Mt::add(c[d], Mt::mul(T(ox), G.bv[0][d])), Mt::mul(T(oy), G.bv[1][d])), Mt::mul(T(oz), G.bv[2][d]));
        }
        e[0] = Mt::sub(c[0], ax); e[1] = Mt::sub(c[1], ay); e[2] = Mt::sub(c[2], az);
        const T sq = Mt::add(Mt::add(Mt::mul(e[0], e[0]), Mt::mul(e[1], e[1])), Mt::mul(e[2], e[2]));
        if (sq < best) { best = sq; dx = e[0]; dy = e[1]; dz = e[2]; }
    }
}
// wrap_coords(v, ::TriclinicBoundary) spatial.jl:588-602
template <class T> __device__ inline void tri_wrap(T& x, T& y, T& z, const GridP<T>& G) {
    using Mt = M<T>;
    T k = Mt::floor(Mt::mul(z, G.rs[2]));
    x = Mt::sub(x, Mt::mul(G.bv[2][0], k)); y = Mt::sub(y, Mt::mul(G.bv[2][1], k)); z = Mt::sub(z, Mt::mul(G.bv[2][2], k));
    k = Mt::floor(Mt::mul(Mt::sub(y, Mt::mul(z, G.cot_bc)), G.rs[1]));
    x = Mt::sub(x, Mt::mul(G.bv[1][0], k)); y = Mt::sub(y, Mt::mul(G.bv[1][1], k)); z = Mt::sub(z, Mt::mul(G.bv[1][2], k));
    const T ddx = Mt::mul(z, G.cxz), ddy = Mt::mul(z, G.cyz);
    k = Mt::floor(Mt::mul(Mt::sub(Mt::sub(x, ddx), Mt::mul(Mt::sub(y, ddy), G.cot_ab)), G.rs[0]));
    x = Mt::sub(x, Mt::mul(G.bv[0][0], k)); y = Mt::sub(y, Mt::mul(G.bv[0][1], k)); z = Mt::sub(z, Mt::mul(G.bv[0][2], k));
}
// the boundary's vector(c1, c2) with the reference's arithmetic, and its wrap_coords — cubic or triclinic
template <class T> __device__ inline void min_image_exact(T ax, T ay, T az, T bx, T by, T bz, const GridP<T>& G, T& dx, T& dy, T& dz) {
    if (G.triclinic) { tri_min_image(ax, ay, az, bx, by, bz, G, dx, dy, dz); return; }
    dx = G.periodic[0] ? vector_1d_exact(ax, bx, G.L[0]) : M<T>::sub(bx, ax);
    dy = G.periodic[1] ? vector_1d_exact(ay, by, G.L[1]) : M<T>::sub(by, ay);
    dz = G.periodic[2] ? vector_1d_exact(az, bz, G.L[2]) : M<T>::sub(bz, az);
}
template <class T> __device__ inline void wrap_point(T& x, T& y, T& z, const GridP<T>& G) {
    if (G.triclinic) { tri_wrap(x, y, z, G); return; }
    if (G.periodic[0]) x = wrap_1d(x, G.L[0]);
    if (G.periodic[1]) y = wrap_1d(y, G.L[1]);
    if (G.periodic[2]) z = wrap_1d(z, G.L[2]);
}

// ---- cutoff strategies on a bare pair potential, cutoffs.jl ----------------------------------------
template <class T> struct LJBare {   // lennard_jones.jl:106-109, 137-140; params (σ², ϵ)
    T s2, e;
    __device__ inline T f(T r) const { T six = s2 * M<T>::rcp(r * r); six = six * six * six; return (T(24) * e * M<T>::rcp(r)) * (T(2) * six * six - six); }
    __device__ inline T v(T r) const { T six = s2 * M<T>::rcp(r * r); six = six * six * six; return T(4) * e * (six * six - six); }
};
template <class T> struct CoulBare {   // coulomb.jl:93-95, 118-120; params (ke, qi, qj)
    T kqq;
    __device__ inline T f(T r) const { return kqq * M<T>::rcp(r * r); }
    __device__ inline T v(T r) const { return kqq * M<T>::rcp(r); }
};

template <class T, class P> __device__ inline T cut_force(int kind, T rc, T ra, const P& p, T r) {
    switch (kind) {
    case MHIP_CUTOFF_NONE: return p.f(r);
    case MHIP_CUTOFF_DISTANCE:
    case MHIP_CUTOFF_SHIFTED_POTENTIAL: return (r <= rc) ? p.f(r) : T(0);
    case MHIP_CUTOFF_SHIFTED_FORCE: return (r <= rc) ? p.f(r) - p.f(rc) : T(0);
    case MHIP_CUTOFF_CUBIC_SPLINE: {
        if (r <= ra) return p.f(r);
        T w = rc - ra, t = (r - ra) / w;
        T f = -(T(6) * t * t - T(6) * t) * p.v(ra) / w - (T(3) * t * t - T(4) * t + T(1)) * (-p.f(ra));
        return (r <= rc) ? f : T(0);
    }
    default: {   // MHIP_CUTOFF_POLYNOMIAL
        if (r <= ra) return p.f(r);
        T w = rc - ra, t = (r - ra) / w;
        T t2 = t * t, t3 = t2 * t, t4 = t3 * t, t5 = t4 * t;
        T S = T(1) - T(6) * t5 + T(15) * t4 - T(10) * t3;
        T dS = (T(-30) * t4 + T(60) * t3 - T(30) * t2) / w;
        return (r <= rc) ? S * p.f(r) - dS * p.v(r) : T(0);
    }
    }
}
template <class T, class P> __device__ inline T cut_pe(int kind, T rc, T ra, const P& p, T r) {
    switch (kind) {
    case MHIP_CUTOFF_NONE: return p.v(r);
    case MHIP_CUTOFF_DISTANCE: return (r <= rc) ? p.v(r) : T(0);
    case MHIP_CUTOFF_SHIFTED_POTENTIAL: return (r <= rc) ? p.v(r) - p.v(rc) : T(0);
    case MHIP_CUTOFF_SHIFTED_FORCE: return (r <= rc) ? p.v(r) + (r - rc) * p.f(rc) - p.v(rc) : T(0);
    case MHIP_CUTOFF_CUBIC_SPLINE: {
        if (r <= ra) return p.v(r);
        T w = rc - ra, t = (r - ra) / w, t2 = t * t, t3 = t2 * t;
        T v = (T(2) * t3 - T(3) * t2 + T(1)) * p.v(ra) + (t3 - T(2) * t2 + t) * w * (-p.f(ra));
        return (r <= rc) ? v : T(0);
    }
    default: {
        if (r <= ra) return p.v(r);
        T w = rc - ra, t = (r - ra) / w;
        T t2 = t * t, t3 = t2 * t, t4 = t3 * t, t5 = t4 * t;
        T S = T(1) - T(6) * t5 + T(15) * t4 - T(10) * t3;
        return (r <= rc) ? S * p.v(r) : T(0);
    }
    }
}

// coulomb.jl:1384-1393 calc_erfc (Abramowitz & Stegun 7.1.26 when approximate_erfc)
template <class T> __device__ inline T ewald_erfc(T ar, T e, int approx) {
    if (approx) {
        T t = M<T>::rcp(T(1) + T(0.3275911) * ar);
        return (T(0.254829592) + (T(-0.284496736) + (T(1.421413741) + (T(-1.453152027) + T(1.061405429) * t) * t) * t) * t) * t * e;
    }
    return M<T>::erfc(ar);
}

enum { LJ_OFF = 0, LJ_DIST = 1, LJ_GENERIC = 2, LJ_DIST_UNIFORM = 3 };   // UNIFORM: every atom has the same σ, ϵ

// Sum over pairwise_inters for one pair (force.jl:79-92 / kernels.jl:3-17).  Returns `fr` such that the
// force on atom j is fr·dr (and −fr·dr on atom i, force.jl:873-874) and, if ENERGY, adds the pair energy.
// PRE_E (the fp32 loops): ei, ej arrive as √ϵ — 0 for an atom with σ = 0 —, taken once per staged atom instead of once per pair.
template <class T, int LJM, int COULM, bool ENERGY, bool PRE_E = false>
__device__ inline T pair_eval(const InterP<T>& I, T r2, T qi, T qj, T si, T sj, T ei, T ej, bool special, T& pe) {
    T fr = T(0);
    // fp32 with a reaction field or Ewald needs 1/r as well: one v_rsq_f32 and a product instead of v_rcp_f32 + v_sqrt_f32
    constexpr bool RSQ = sizeof(T) == 4 && (COULM == MHIP_COUL_REACTION_FIELD || COULM == MHIP_COUL_EWALD_DIRECT || COULM == COUL_EWALD_EXACT);
    [[maybe_unused]] T inv_r1 = T(0);
    T inv_r2;
    if constexpr (RSQ) { inv_r1 = M<T>::rsq(r2); inv_r2 = inv_r1 * inv_r1; } else inv_r2 = M<T>::rcp(r2);
    if constexpr (LJM == LJ_DIST_UNIFORM) {
        // one atom type: σ_mix = σ, ϵ_mix = ϵ (Lorentz / geometric mixing of equal values), hoisted to the host
        T six = I.lj_s2 * inv_r2; six = six * six * six;
        bool in = r2 <= I.lj_rc2;
        T w = special ? I.lj_w : T(1);
        T f = I.lj_24e * (T(2) * six * six - six) * inv_r2;
        fr += in ? f * w : T(0);
        if constexpr (ENERGY) pe += in ? I.lj_4e * (six * six - six) * w : T(0);
    } else if constexpr (LJM != LJ_OFF) {
        T s = PRE_E ? si + sj : (si + sj) * T(0.5);                 // LorentzMixing (PRE_E: both arrive halved — exact, so the sum is the same number)
        T e;
        if constexpr (PRE_E) e = ei * ej;                           // √ϵi·√ϵj, either factor 0 where σ = 0
        else {
            e = M<T>::sqrt(ei * ej);                                // GeometricMixing
            e = (si == T(0) || sj == T(0)) ? T(0) : e;              // LJZeroShortcut (ϵ == 0 already yields 0)
        }
        T w = special ? I.lj_w : T(1);
        if constexpr (LJM == LJ_DIST) {
            // DistanceCutoff: F/r = 24ϵ(2 s6² − s6)/r², zero past the cutoff (r ≤ rc ⇔ r² ≤ rc²)
            T six = (s * s) * inv_r2; six = six * six * six;
            bool in = r2 <= I.lj_rc2;
            T f = T(24) * e * (T(2) * six * six - six) * inv_r2;
            fr += in ? f * w : T(0);
            if constexpr (ENERGY) pe += in ? T(4) * e * (six * six - six) * w : T(0);
        } else {
            T r = M<T>::sqrt(r2);
            LJBare<T> p{s * s, e};
            fr += cut_force(I.lj_cut, I.lj_rc, I.lj_ra, p, r) * M<T>::rcp(r) * w;
            if constexpr (ENERGY) pe += cut_pe(I.lj_cut, I.lj_rc, I.lj_ra, p, r) * w;
        }
    }
    if constexpr (COULM == MHIP_COUL_PLAIN) {
        T r = M<T>::sqrt(r2);
        CoulBare<T> p{I.ke * qi * qj};
        T w = special ? I.c_w : T(1);
        fr += cut_force(I.coul_cut, I.c_rc, I.c_ra, p, r) * M<T>::rcp(r) * w;
        if constexpr (ENERGY) pe += cut_pe(I.coul_cut, I.c_rc, I.c_ra, p, r) * w;
    } else if constexpr (COULM == MHIP_COUL_REACTION_FIELD) {
        T inv_r = RSQ ? inv_r1 : M<T>::sqrt(inv_r2);
        T kqq = I.ke * qi * qj;
        T krf = special ? T(0) : I.krf;                              // 1-4 pairs: no reaction field
        T crf = special ? T(0) : I.crf;
        T w = special ? I.c_w : T(1);
        bool in = r2 <= I.c_rc2;
        fr += in ? kqq * (inv_r - T(2) * krf * r2) * inv_r2 * w : T(0);
        if constexpr (ENERGY) pe += in ? kqq * (inv_r + krf * r2 - crf) * w : T(0);
    } else if constexpr (COULM == MHIP_COUL_EWALD_DIRECT || COULM == COUL_EWALD_EXACT) {
        T inv_r = RSQ ? inv_r1 : M<T>::sqrt(inv_r2);
        T r = r2 * inv_r;
        T ar = I.alpha * r;
        T ex = M<T>::exp(-(ar * ar));
        T ec = ewald_erfc(ar, ex, COULM == MHIP_COUL_EWALD_DIRECT ? 1 : 0);   // (the engine picks the variant by approximate_erfc)
        T kqq = I.ke * qi * qj;
        bool in = r2 <= I.c_rc2;
        T f3 = kqq * inv_r * inv_r2;
        T g = special ? I.c_w : ec + I.two_over_sqrt_pi * ar * ex;   // special: plain Coulomb × weight
        fr += in ? f3 * g : T(0);
        if constexpr (ENERGY) pe += in ? kqq * inv_r * (special ? I.c_w : ec) : T(0);
    }
    return fr;
}

// ---- two partners side by side (fp32) -------------------------------------------------------------------------------------------
// The per-atom-parameter fp32 loops evaluate two list entries at once in the halves of 64-bit registers: everything between r² and the
// force factor is v_pk_*_f32 (two pairs per instruction; the packed pipe issues at 5.4 cycles per wave-instruction against 4.2 for a
// plain one: 1.55× per pair), the transcendental unit is asked three times per pair instead of five or six — v_rsq per partner, v_exp
// per partner, and ONE v_rcp per two partners for the Abramowitz–Stegun t = 1/(1 + p·αr): t_a = d_b/(d_a d_b), t_b = d_a/(d_a d_b),
// both d in [1, 2] inside the cutoff — and the cutoffs are a clamped packed FMA, clamp((rc²⁺ − r²)·2¹⁰⁰) ∈ {0, 1} exactly
// (rc²⁺ = the float after rc²: r² <= rc² as the reference tests r <= rc), instead of v_cmp + v_cndmask per partner.
// Same formulas as pair_eval<float, LJ_DIST, EWALD / RF, false, PRE_E = true> (lennard_jones.jl:106-109, coulomb.jl:764-803, 1384-1441,
// mixing.jl:3-34); neither partner may be a special (1-4) pair — rows that hold one take the one-partner path.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ inline v2f pk_clamp01(v2f r2, v2f neg_big, v2f rc2n_big) {      // {r2.x <= rc2, r2.y <= rc2} as 1.0f / 0.0f
    v2f in;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp\n\ts_nop 0" : "=v"(in) : "v"(r2), "v"(neg_big), "v"(rc2n_big));   // (the nop: a dependent read of a packed result needs a wait state the hazard recogniser cannot see)
    return in;
}
struct Pk2Consts {           // per launch, wave-uniform
    v2f cut_a, cut_b;
    __device__ inline explicit Pk2Consts(const InterP<float>& I) {
        const float a = __int_as_float(__float_as_int(I.lj_rc2) + 1);
        cut_a = (v2f){-0x1p100f, -0x1p100f}; cut_b = (v2f){a * 0x1p100f, a * 0x1p100f};
    }
    // (one cutoff for both interactions — what setup.jl's dist_cutoff gives; a launch with two different ones keeps the one-partner loop)
    __device__ static inline bool usable(const InterP<float>& I) { return I.lj_rc2 == I.c_rc2; }
};
// kqi = ke·q_i, si = σ_i/2, ei24 = 24·√ϵ_i (0 where σ_i = 0); qj, sj (= σ_j/2), ej (= √ϵ_j) of the two partners.
// F/r = [24ϵ(2 s6² − s6) + kqq·g(r)/r] / r² inside the cutoff, g = erfc(αr) + 2αr/√π·exp(−α²r²) (Ewald) or 1 − 2 krf r³ (reaction field)
template <int COULM>
__device__ inline v2f pair_eval2(const InterP<float>& I, const Pk2Consts& K, v2f r2, float kqi, v2f qj, float si, v2f sj, float ei24, v2f ej) {
    const v2f inv_r = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
    const v2f inv_r2 = inv_r * inv_r;
    const v2f in = pk_clamp01(r2, K.cut_a, K.cut_b);
    const v2f s = sj + (v2f){si, si};
    v2f six = (s * s) * inv_r2; six = six * six * six;
    const v2f t6 = __builtin_elementwise_fma(six, (v2f){2.f, 2.f}, (v2f){-1.f, -1.f}) * six;
    const v2f lj = (ej * (v2f){ei24, ei24}) * t6;
    const v2f kq_ir = (qj * (v2f){kqi, kqi}) * inv_r;
    v2f sum;
    if constexpr (COULM == MHIP_COUL_REACTION_FIELD) {
        const v2f r3 = r2 * (r2 * inv_r);
        const v2f g = __builtin_elementwise_fma(r3, (v2f){-2.f * I.krf, -2.f * I.krf}, (v2f){1.f, 1.f});
        sum = __builtin_elementwise_fma(kq_ir, g, lj);
    } else {
        const v2f r = r2 * inv_r;
        const float na2 = -1.4426950408889634f * I.alpha * I.alpha, pa = 0.3275911f * I.alpha, ca = I.two_over_sqrt_pi * I.alpha;
        const v2f x2 = r2 * (v2f){na2, na2};                                                     // exp(−(αr)²) = 2^(−α² r² log2 e)
        const v2f ex = {__builtin_amdgcn_exp2f(x2.x), __builtin_amdgcn_exp2f(x2.y)};
        const v2f d = __builtin_elementwise_fma(r, (v2f){pa, pa}, (v2f){1.f, 1.f});
        const float invd = __builtin_amdgcn_rcpf(d.x * d.y);
        const v2f t = (v2f){d.y, d.x} * (v2f){invd, invd};
        v2f pl = __builtin_elementwise_fma(t, (v2f){1.061405429f, 1.061405429f}, (v2f){-1.453152027f, -1.453152027f});
        pl = __builtin_elementwise_fma(pl, t, (v2f){1.421413741f, 1.421413741f});
        pl = __builtin_elementwise_fma(pl, t, (v2f){-0.284496736f, -0.284496736f});
        pl = __builtin_elementwise_fma(pl, t, (v2f){0.254829592f, 0.254829592f});
        const v2f ec = (pl * t) * ex;                                                             // calc_erfc, coulomb.jl:1384-1393
        const v2f g = __builtin_elementwise_fma(r * ex, (v2f){ca, ca}, ec);
        sum = __builtin_elementwise_fma(kq_ir, g, lj);
    }
    return sum * (inv_r2 * in);
}

}  // namespace mhip
