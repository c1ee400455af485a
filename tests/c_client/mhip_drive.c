/* mhip_drive.c — a plain C99 client of libmollyhip.so (no Python, no ctypes): the call sequence a host program — or the Julia
 * shim of INTEGRATION.md — makes through the C ABI of include/mollyhip.h:
 *
 *   mhip_create → mhip_set_atoms → mhip_set_state → mhip_forces / mhip_potential_energy / mhip_kinetic_energy
 *   → mhip_vv_run → mhip_get_state → [mhip_set_state → mhip_forces(step_n)] × k → mhip_get_stats → mhip_destroy
 *
 * The system is a small argon-like Lennard-Jones lattice generated here from an LCG (fp64).  Everything the library returned is written
 * to a binary file; tests/test_gpu_c_client.py compiles this file with gcc, runs it on the GPU box and checks the numbers against the
 * CPU oracle on the same inputs.  Usage: mhip_drive <n_side> <n_steps> <out.bin>
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mollyhip.h"

#define CHECK(call) do { int32_t rc_ = (call); if (rc_ != MHIP_OK) { fprintf(stderr, "%s failed: %d (%s)\n", #call, (int)rc_, mhip_last_error(ctx)); return 2; } } while (0)

static uint64_t lcg_state = 0x9E3779B97F4A7C15ull;
static double lcg_uniform(void) {   /* [0, 1) */
    lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(lcg_state >> 11) * (1.0 / 9007199254740992.0);
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s n_side n_steps out.bin\n", argv[0]); return 1; }
    const int n_side = atoi(argv[1]), n_steps = atoi(argv[2]);
    const int64_t n = (int64_t)n_side * n_side * n_side;
    const double spacing = 0.36183, box = n_side * spacing, dt = 0.002;
    double *x = malloc(3 * n * sizeof(double)), *v = malloc(3 * n * sizeof(double)), *f0 = malloc(3 * n * sizeof(double));
    double *x1 = malloc(3 * n * sizeof(double)), *v1 = malloc(3 * n * sizeof(double)), *f1 = malloc(3 * n * sizeof(double));
    double *sigma = malloc(n * sizeof(double)), *eps = malloc(n * sizeof(double)), *mass = malloc(n * sizeof(double));
    if (!x || !v || !f0 || !x1 || !v1 || !f1 || !sigma || !eps || !mass) return 1;
    int64_t a = 0;
    for (int i = 0; i < n_side; ++i) for (int j = 0; j < n_side; ++j) for (int k = 0; k < n_side; ++k, ++a) {
        const int g[3] = {i, j, k};
        for (int d = 0; d < 3; ++d) {
            x[3 * a + d] = (g[d] + 0.5) * spacing + (lcg_uniform() - 0.5) * 0.04;
            v[3 * a + d] = (lcg_uniform() + lcg_uniform() + lcg_uniform() - 1.5) * 0.27;   /* ~N(0, 0.135²) nm/ps: argon near 85 K */
        }
        sigma[a] = 0.34; eps[a] = 0.997; mass[a] = 39.948;
    }
    double vcm[3] = {0, 0, 0};
    for (a = 0; a < n; ++a) for (int d = 0; d < 3; ++d) vcm[d] += v[3 * a + d] / (double)n;
    for (a = 0; a < n; ++a) for (int d = 0; d < 3; ++d) v[3 * a + d] -= vcm[d];

    mhip_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.precision = 64; cfg.device_id = 0; cfg.n_atoms = n;
    for (int d = 0; d < 3; ++d) { cfg.box[d] = box; cfg.origin[d] = 0.0; cfg.periodic[d] = 1; }
    cfg.rebuild_every = 10; cfg.r_list = 1.2;
    cfg.inter.lj_enabled = 1; cfg.inter.lj_cutoff_kind = MHIP_CUTOFF_DISTANCE; cfg.inter.lj_rc = 1.0; cfg.inter.lj_weight_special = 1.0;
    cfg.inter.coul_kind = MHIP_COUL_NONE; cfg.inter.coul_ke = 138.93545764; cfg.inter.coul_weight_special = 1.0; cfg.inter.rf_dielectric = 1.0;

    mhip_ctx* ctx = NULL;
    int32_t rc = mhip_create(&ctx, &cfg);
    if (rc != MHIP_OK) { fprintf(stderr, "mhip_create failed: %d (%s)\n", (int)rc, mhip_last_error(NULL)); return 2; }
    CHECK(mhip_set_atoms(ctx, NULL, sigma, eps, mass, NULL, MHIP_MEM_HOST));
    CHECK(mhip_set_state(ctx, x, v, MHIP_MEM_HOST));
    CHECK(mhip_forces(ctx, 0, 0, f0, NULL, MHIP_MEM_HOST));
    double pe0 = 0, ke0 = 0, pe1 = 0, ke1 = 0;
    CHECK(mhip_potential_energy(ctx, 0, &pe0));
    CHECK(mhip_kinetic_energy(ctx, &ke0));
    CHECK(mhip_vv_run(ctx, 0, n_steps, dt, 1));
    CHECK(mhip_get_state(ctx, x1, v1, MHIP_MEM_HOST));
    CHECK(mhip_potential_energy(ctx, n_steps, &pe1));
    CHECK(mhip_kinetic_energy(ctx, &ke1));
    /* the drop-in cadence: hand the same coordinates back five times and ask for forces at consecutive step numbers */
    mhip_stats st0, st1;
    CHECK(mhip_get_stats(ctx, &st0));
    for (int k = 0; k < 5; ++k) {
        CHECK(mhip_set_state(ctx, x1, NULL, MHIP_MEM_HOST));
        CHECK(mhip_forces(ctx, n_steps + 1 + k, 0, f1, NULL, MHIP_MEM_HOST));
    }
    CHECK(mhip_get_stats(ctx, &st1));
    CHECK(mhip_check_finite(ctx));
    /* a barostat's trial move (coupling.jl:886-932): box and coordinates scaled by 1 %, the energy there, the move taken back, the old energy again */
    double pe_scaled = 0, pe_back = 0;
    {
        const double box2[3] = {1.01 * box, 1.01 * box, 1.01 * box}, box1[3] = {box, box, box};
        double* xs = (double*)malloc(sizeof(double) * 3 * n);
        if (!xs) return 1;
        for (int64_t k = 0; k < 3 * n; ++k) xs[k] = 1.01 * x1[k];
        CHECK(mhip_set_box(ctx, box2, NULL));
        CHECK(mhip_set_state(ctx, xs, NULL, MHIP_MEM_HOST));
        CHECK(mhip_potential_energy(ctx, n_steps + 6, &pe_scaled));
        CHECK(mhip_set_box(ctx, box1, NULL));
        CHECK(mhip_set_state(ctx, x1, NULL, MHIP_MEM_HOST));
        CHECK(mhip_potential_energy(ctx, n_steps + 6, &pe_back));
        free(xs);
    }
    CHECK(mhip_destroy(ctx));

    FILE* out = fopen(argv[3], "wb");
    if (!out) return 1;
    const double head[8] = {(double)n, box, dt, (double)n_steps, pe0, ke0, pe1, ke1};
    const double tail[6] = {(double)(st1.n_outer_builds - st0.n_outer_builds), (double)(st1.n_filter_passes - st0.n_filter_passes),
                            (double)(st1.n_force_calls - st0.n_force_calls), (double)st1.n_pairs_full, pe_scaled, pe_back};
    fwrite(head, sizeof(double), 8, out);
    fwrite(x, sizeof(double), 3 * n, out); fwrite(v, sizeof(double), 3 * n, out); fwrite(f0, sizeof(double), 3 * n, out);
    fwrite(x1, sizeof(double), 3 * n, out); fwrite(v1, sizeof(double), 3 * n, out); fwrite(f1, sizeof(double), 3 * n, out);
    fwrite(tail, sizeof(double), 6, out);
    fclose(out);
    printf("mhip_drive ok: %lld atoms, %d steps, PE %.6f -> %.6f kJ/mol, KE %.6f -> %.6f kJ/mol, searches during the set_state loop: %d\n",
           (long long)n, n_steps, pe0, pe1, ke0, ke1, (int)tail[0]);
    free(x); free(v); free(f0); free(x1); free(v1); free(f1); free(sigma); free(eps); free(mass);
    return 0;
}
