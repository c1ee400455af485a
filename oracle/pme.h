// oracle/pme.h — CPU restatement of Molly's smooth particle-mesh Ewald reciprocal-space term.  TEST INFRASTRUCTURE ONLY
// (same rules as oracle.cpp: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it).
//
// Follows /root/reference/src/interactions/ewald.jl:
//   pme_params            :479-482   mesh size per axis (done by the caller, pyoracle.pme_mesh)
//   pme_bspline_moduli    :311-358
//   grid_placement_inner! :484-493
//   update_bsplines_inner!:518-556
//   spread_charge_inner!  :598-621
//   recip_conv_inner!/recip_conv! :676-751
//   interpolate_force_inner! :805-840
//   ewald_pe_forces!      :873-929   (self and net-charge terms :917-927)
// recip_box = invert_box_vectors(boundary) (spatial.jl:327-347): diag(1/Lx, 1/Ly, 1/Lz) for a CubicBoundary, the lower-triangular inverse of the basis for a
// TriclinicBoundary; it enters the placement (:486), the wave vectors of the convolution (:688-694) and the force transform (:846-849).  The two FFTs (plan_fft!, plan_bfft!: unnormalised forward
// e^{-2πi jk/n} / backward e^{+2πi jk/n}) are evaluated as separable direct DFTs with double-precision twiddles —
// the meshes here are ~50³, and a direct sum needs no FFT library.
// Threads (nthreads > 1, ewald.jl's n_threads > 1 methods): the B-splines and the force interpolation run over blocks of atoms
// (Threads.@threads over atoms, :570, :854-859), the spreading into one private real mesh per thread over atoms chunk:n_threads:N with the
// serial sum of the buffers behind it (:632-646), the convolution over kx with per-thread energy / virial sums (:739-750).  The two
// transforms are threaded over lines (the reference calls FFTW there; a line's arithmetic does not depend on the thread count, so the
// results of this file are the same for every nthreads except for the order of the mesh and energy sums).
#pragma once
#include <cmath>
#include <complex>
#include <cstdint>
#include <thread>
#include <vector>

namespace orc_pme {

template <class T> struct Pme {
    int order, n[3];
    T alpha, f_div_er, L[3];
    T r[3][3];                       // recip_box, r[e][d] = recip_box[e+1][d+1]
    std::vector<T> bsm[3];

    // box: side lengths (triclinic: v1.x, v2.y, v3.z — volume(boundary) is their product either way, spatial.jl:311-312); bv9 (nullable): the three basis
    // vectors of a TriclinicBoundary, row-major
    Pme(int order_, const int32_t* mesh, T alpha_, T ke, T eps_r, const T* box, const double* bv9 = nullptr) : order(order_), alpha(alpha_), f_div_er(ke / eps_r) {
        for (int d = 0; d < 3; ++d) { n[d] = mesh[d]; L[d] = box[d]; }
        for (int e = 0; e < 3; ++e) for (int d = 0; d < 3; ++d) r[e][d] = T(0);
        if (!bv9) { for (int d = 0; d < 3; ++d) r[d][d] = T(1) / L[d]; }            // :327-336
        else {                                                                       // :338-347
            T bv[3][3]; for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) bv[i][k] = T(bv9[3 * i + k]);
            const T V = bv[0][0] * bv[1][1] * bv[2][2];
            r[0][0] = (bv[1][1] * bv[2][2]) / V;
            r[1][0] = (-bv[1][0] * bv[2][2]) / V; r[1][1] = (bv[0][0] * bv[2][2]) / V;
            r[2][0] = (bv[1][0] * bv[2][1] - bv[1][1] * bv[2][0]) / V; r[2][1] = (-bv[0][0] * bv[2][1]) / V; r[2][2] = (bv[0][0] * bv[1][1]) / V;
        }
        moduli();
    }

    // ewald.jl:311-358
    void moduli() {
        int nmax = std::max(n[0], std::max(n[1], n[2]));
        std::vector<T> data(order, T(0)), bdata(nmax + order + 1, T(0));
        data[0] = T(1);
        for (int k = 3; k <= order - 1; ++k) {
            T d = T(1) / (T(k) - T(1));
            data[k - 1] = T(0);
            for (int l = 1; l <= k - 2; ++l) data[k - l - 1] = d * (T(l) * data[k - l - 2] + T(k - l) * data[k - l - 1]);
            data[0] *= d;
        }
        T d = T(1) / (T(order) - T(1));
        data[order - 1] = T(0);
        for (int l = 1; l <= order - 2; ++l) data[order - l - 1] = d * (T(l) * data[order - l - 2] + T(order - l) * data[order - l - 1]);
        data[0] *= d;
        for (int i = 1; i <= order; ++i) bdata[i] = data[i - 1];             // bsplines_data[i+1] = data[i] (1-based)
        for (int dd = 0; dd < 3; ++dd) {
            int nd = n[dd];
            bsm[dd].assign(nd, T(0));
            for (int i = 1; i <= nd; ++i) {
                T sc = T(0), ss = T(0);
                for (int j = 1; j <= nd; ++j) {
                    T arg = T(2) * T(M_PI) * T(i - 1) * T(j - 1) / T(nd);
                    sc += bdata[j - 1] * std::cos(arg);
                    ss += bdata[j - 1] * std::sin(arg);
                }
                bsm[dd][i - 1] = sc * sc + ss * ss;
            }
            for (int i = 1; i <= nd; ++i)
                if (bsm[dd][i - 1] < T(1e-7)) bsm[dd][i - 1] = (bsm[dd][(i - 2 + nd) % nd] + bsm[dd][i % nd]) / T(2);
        }
    }

    // :484-493 and :518-556 for one atom; th/dth: [3][order]
    void place(const T* c, int* idx, T* th, T* dth) const {
        for (int d = 0; d < 3; ++d) {
            T t = c[0] * r[0][d] + c[1] * r[1][d] + c[2] * r[2][d];      // sum(coords[i] .* recip_box[:, d]) (:486)
            t = (t - std::floor(t)) * T(n[d]);
            int ti = (int)std::floor(t);
            T dr = t - T(ti);
            idx[d] = ti % n[d];
            T* b = th + d * order; T* db = dth + d * order;
            for (int k = 0; k < order; ++k) b[k] = T(0);
            b[order - 1] = T(0); b[1] = dr; b[0] = T(1) - dr;
            for (int k = 3; k <= order - 1; ++k) {
                T dv = T(1) / (T(k) - T(1));
                b[k - 1] = dv * dr * b[k - 2];
                for (int l = 1; l <= k - 2; ++l) b[k - l - 1] = dv * ((dr + T(l)) * b[k - l - 2] + (T(k - l) - dr) * b[k - l - 1]);
                b[0] *= dv * (T(1) - dr);
            }
            db[0] = -b[0];
            for (int k = 1; k <= order - 1; ++k) db[k] = b[k - 1] - b[k];
            T dv = T(1) / (T(order) - T(1));
            b[order - 1] = dv * dr * b[order - 2];
            for (int l = 1; l <= order - 2; ++l) b[order - l - 1] = dv * ((dr + T(l)) * b[order - l - 2] + (T(order - l) - dr) * b[order - l - 1]);
            b[0] *= dv * (T(1) - dr);
        }
    }

    template <class F> static void par_for(int64_t n, int nthreads, F body) {      // body(t, lo, hi) on contiguous blocks
        nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(nthreads, n));
        if (nthreads == 1) { body(0, (int64_t)0, n); return; }
        std::vector<std::thread> th;
        const int64_t per = (n + nthreads - 1) / nthreads;
        for (int t = 0; t < nthreads; ++t) th.emplace_back([=, &body] { body(t, std::min(n, t * per), std::min(n, (t + 1) * per)); });
        for (auto& x : th) x.join();
    }

    // unnormalised DFT along every axis, sign = -1 forward / +1 backward; grid index (x*ny + y)*nz + z
    void dft3(std::vector<std::complex<T>>& g, int sign, int nthreads = 1) const {
        const int64_t stride[3] = {(int64_t)n[1] * n[2], n[2], 1};
        for (int a = 0; a < 3; ++a) {
            const int na = n[a];
            std::vector<std::complex<double>> w(na);
            for (int m = 0; m < na; ++m) { double ang = sign * 2.0 * M_PI * m / na; w[m] = {std::cos(ang), std::sin(ang)}; }
            const int b = (a + 1) % 3, c = (a + 2) % 3;
            par_for((int64_t)n[b] * n[c], nthreads, [&](int, int64_t lo, int64_t hi) {
                std::vector<std::complex<double>> line(na), out(na);
                for (int64_t l = lo; l < hi; ++l) {
                    const int ib = (int)(l / n[c]), ic = (int)(l - (int64_t)ib * n[c]);
                    const int64_t base = ib * stride[b] + ic * stride[c];
                    for (int j = 0; j < na; ++j) line[j] = std::complex<double>(g[base + j * stride[a]]);
                    for (int k = 0; k < na; ++k) {
                        std::complex<double> s = 0; int m = 0;
                        for (int j = 0; j < na; ++j) { s += line[j] * w[m]; m += k; if (m >= na) m -= na; }
                        out[k] = s;
                    }
                    for (int k = 0; k < na; ++k) g[base + k * stride[a]] = std::complex<T>((T)out[k].real(), (T)out[k].imag());
                }
            });
        }
    }

    // ewald_pe_forces! :873-929.  x: 3n coords, q: n charges; fs (nullable): 3n forces, the PME force is ADDED (Fs[i] -= f, :838)
    // vir (nullable, 9 entries, ADDED): reciprocal-space virial of recip_conv_inner! (:701-723), halved like the energy (:747-750), plus the
    // net-charge term charge_E·I (:925-927)
    T run(int64_t natoms, const T* x, const T* q, T* fs, T* vir = nullptr, int nthreads = 1) const {
        const int nx = n[0], ny = n[1], nz = n[2];
        const size_t nmesh = (size_t)nx * ny * nz;
        nthreads = std::max(1, nthreads);
        std::vector<std::complex<T>> grid(nmesh, std::complex<T>(0, 0));
        std::vector<int> idx(3 * (size_t)natoms);
        std::vector<T> th(3 * (size_t)order * natoms), dth(3 * (size_t)order * natoms);
        par_for(natoms, nthreads, [&](int, int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; ++i) place(x + 3 * i, &idx[3 * i], &th[3 * order * i], &dth[3 * order * i]); });
        auto spread_one = [&](int64_t i, auto add) {   // spread_charge_inner! :598-621
            const T* t = &th[3 * order * i];
            for (int ix = 0; ix < order; ++ix) {
                int xi = (idx[3 * i] + ix) % nx; T qx = q[i] * t[ix];
                for (int iy = 0; iy < order; ++iy) {
                    int yi = (idx[3 * i + 1] + iy) % ny; T qxy = qx * t[order + iy];
                    for (int iz = 0; iz < order; ++iz) {
                        int zi = (idx[3 * i + 2] + iz) % nz;
                        add(((size_t)xi * ny + yi) * nz + zi, qxy * t[2 * order + iz]);
                    }
                }
            }
        };
        const int n_spread = std::min(nthreads, 4);       // n_spread_thr = min(n_threads, 4) (:888)
        if (n_spread == 1) {                              // :623-630
            for (int64_t i = 0; i < natoms; ++i) spread_one(i, [&](size_t at, T v) { grid[at] += std::complex<T>(v, T(0)); });
        } else {                                          // :632-646: a private real mesh per thread, atoms chunk : n_threads : N, then summed in order
            std::vector<std::vector<T>> buf(n_spread);
            std::vector<std::thread> tt;
            for (int c = 0; c < n_spread; ++c) tt.emplace_back([&, c] {
                buf[c].assign(nmesh, T(0));
                for (int64_t i = c; i < natoms; i += n_spread) spread_one(i, [&](size_t at, T v) { buf[c][at] += v; });
            });
            for (auto& t : tt) t.join();
            par_for((int64_t)nmesh, nthreads, [&](int, int64_t lo, int64_t hi) {
                for (int64_t at = lo; at < hi; ++at) { T s = buf[0][at]; for (int c = 1; c < n_spread; ++c) s += buf[c][at]; grid[at] = std::complex<T>(s, T(0)); }
            });
        }
        dft3(grid, -1, nthreads);
        // recip_conv! :727-751 (threaded over kx with per-thread sums, :739-750)
        const T V = L[0] * L[1] * L[2];
        const T factor = T(M_PI) * T(M_PI) / (alpha * alpha), boxfactor = T(M_PI) * V;
        const T maxk[3] = {T(0.5) * T(nx + 1), T(0.5) * T(ny + 1), T(0.5) * T(nz + 1)};
        std::vector<T> esum_t(nthreads, T(0)), vir_t((size_t)9 * nthreads, T(0));
        par_for(nx, nthreads, [&](int tq, int64_t lo, int64_t hi) {
          T esum = T(0); T* virq = &vir_t[(size_t)9 * tq];
          for (int kx = (int)lo; kx < (int)hi; ++kx) for (int ky = 0; ky < ny; ++ky) for (int kz = 0; kz < nz; ++kz) {
            if (kx == 0 && ky == 0 && kz == 0) continue;
            T mx = T(kx) < maxk[0] ? T(kx) : T(kx - nx), my = T(ky) < maxk[1] ? T(ky) : T(ky - ny), mz = T(kz) < maxk[2] ? T(kz) : T(kz - nz);
            T mhx = mx * r[0][0], mhy = mx * r[1][0] + my * r[1][1], mhz = mx * r[2][0] + my * r[2][1] + mz * r[2][2];      // :688-694
            T bx = boxfactor * bsm[0][kx], by = bsm[1][ky], bz = bsm[2][kz];
            std::complex<T>& gv = grid[((size_t)kx * ny + ky) * nz + kz];
            T d1 = gv.real(), d2 = gv.imag();
            T m2 = mhx * mhx + mhy * mhy + mhz * mhz;
            T denom = m2 * bx * by * bz;
            T eterm = f_div_er * std::exp(-factor * m2) / denom;
            gv = std::complex<T>(d1 * eterm, d2 * eterm);
            esum += eterm * (d1 * d1 + d2 * d2);
            if (vir) {   // V·P_k = E_k [I − 2(1 + factor·m²)(m ⊗ m)/m²]
                const T Ek = eterm * (d1 * d1 + d2 * d2), coeff = T(2) * (T(1) + factor * m2) / m2, mh[3] = {mhx, mhy, mhz};
                for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) virq[3 * a + b] += (Ek * ((a == b ? T(1) : T(0)) - coeff * mh[a] * mh[b])) / T(2);
            }
          }
          esum_t[tq] = esum;
        });
        T esum = T(0);
        for (int c = 0; c < nthreads; ++c) { esum += esum_t[c]; if (vir) for (int a = 0; a < 9; ++a) vir[a] += vir_t[(size_t)9 * c + a]; }
        const T recip_E = esum / T(2);
        dft3(grid, +1, nthreads);
        if (fs) {
          par_for(natoms, nthreads, [&](int, int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i) {   // interpolate_force_inner! :805-840
                const T *t = &th[3 * order * i], *dt = &dth[3 * order * i];
                T fx = 0, fy = 0, fz = 0;
                for (int ix = 0; ix < order; ++ix) {
                    int xi = (idx[3 * i] + ix) % nx; T tx = t[ix], dtx = dt[ix];
                    for (int iy = 0; iy < order; ++iy) {
                        int yi = (idx[3 * i + 1] + iy) % ny; T ty = t[order + iy], dty = dt[order + iy];
                        T dtx_ty = dtx * ty, tx_dty = tx * dty, txy = tx * ty;
                        for (int iz = 0; iz < order; ++iz) {
                            int zi = (idx[3 * i + 2] + iz) % nz; T tz = t[2 * order + iz], dtz = dt[2 * order + iz];
                            T gvv = grid[((size_t)xi * ny + yi) * nz + zi].real();
                            fx += dtx_ty * tz * gvv; fy += tx_dty * tz * gvv; fz += txy * dtz * gvv;
                        }
                    }
                }
                fs[3 * i + 0] -= q[i] * (fx * T(nx) * r[0][0]);                                                          // :846-849
                fs[3 * i + 1] -= q[i] * (fx * T(nx) * r[1][0] + fy * T(ny) * r[1][1]);
                fs[3 * i + 2] -= q[i] * (fx * T(nx) * r[2][0] + fy * T(ny) * r[2][1] + fz * T(nz) * r[2][2]);
            }
          });
        }
        T pc_sum = 0, pc_abs2 = 0;
        for (int64_t i = 0; i < natoms; ++i) { pc_sum += q[i]; pc_abs2 += q[i] * q[i]; }
        const T charge_E = -f_div_er * T(M_PI) * pc_sum * pc_sum / (T(2) * V * alpha * alpha);
        const T self_E = f_div_er * -pc_abs2 * alpha / std::sqrt(T(M_PI)) + charge_E;
        if (vir) for (int a = 0; a < 3; ++a) vir[4 * a] += charge_E;
        return recip_E + self_E;
    }
};

}  // namespace orc_pme
