// hilbert.h — generalised Hilbert ("gilbert") space-filling curve over an arbitrary nx×ny×nz cell grid.
// Host-side, run once per grid.  Consecutive cells of the curve are spatially adjacent (or nearly so for
// odd sizes), so any run of consecutive cell-sorted atoms is spatially compact — the property the LDS
// tiles of k_build / k_forces rely on.  (The reference orders atoms by a 30-bit Morton code,
// src/kernels.jl:575-645; Morton runs have long jumps, which would blow up tile sizes here.)
#pragma once
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace mhip {

struct Gilbert3D {
    std::vector<uint32_t>& rank;   // rank[(z*ny + y)*nx + x] = position along the curve
    int nx, ny, nz;
    uint32_t next = 0;
    bool bad = false;
    static int sgn(int v) { return (v > 0) - (v < 0); }
    static int half(int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); }   // floor division by 2
    void visit(int x, int y, int z) {
        if (x < 0 || y < 0 || z < 0 || x >= nx || y >= ny || z >= nz) { bad = true; ++next; return; }
        rank[((size_t)z * ny + y) * nx + x] = next++;
    }

    void gen(int x, int y, int z, int ax, int ay, int az, int bx, int by, int bz, int cx, int cy, int cz) {
        int w = std::abs(ax + ay + az), h = std::abs(bx + by + bz), d = std::abs(cx + cy + cz);
        int dax = sgn(ax), day = sgn(ay), daz = sgn(az), dbx = sgn(bx), dby = sgn(by), dbz = sgn(bz), dcx = sgn(cx), dcy = sgn(cy), dcz = sgn(cz);
        if (h == 1 && d == 1) { for (int i = 0; i < w; ++i) { visit(x, y, z); x += dax; y += day; z += daz; } return; }
        if (w == 1 && d == 1) { for (int i = 0; i < h; ++i) { visit(x, y, z); x += dbx; y += dby; z += dbz; } return; }
        if (w == 1 && h == 1) { for (int i = 0; i < d; ++i) { visit(x, y, z); x += dcx; y += dcy; z += dcz; } return; }
        int ax2 = half(ax), ay2 = half(ay), az2 = half(az), bx2 = half(bx), by2 = half(by), bz2 = half(bz), cx2 = half(cx), cy2 = half(cy), cz2 = half(cz);
        int w2 = std::abs(ax2 + ay2 + az2), h2 = std::abs(bx2 + by2 + bz2), d2 = std::abs(cx2 + cy2 + cz2);
        if ((w2 % 2) && w > 2) { ax2 += dax; ay2 += day; az2 += daz; }   // prefer even steps
        if ((h2 % 2) && h > 2) { bx2 += dbx; by2 += dby; bz2 += dbz; }
        if ((d2 % 2) && d > 2) { cx2 += dcx; cy2 += dcy; cz2 += dcz; }
        if (2 * w > 3 * h && 2 * w > 3 * d) {          // wide: split along a only
            gen(x, y, z, ax2, ay2, az2, bx, by, bz, cx, cy, cz);
            gen(x + ax2, y + ay2, z + az2, ax - ax2, ay - ay2, az - az2, bx, by, bz, cx, cy, cz);
        } else if (3 * h > 4 * d) {                    // do not split along c
            gen(x, y, z, bx2, by2, bz2, cx, cy, cz, ax2, ay2, az2);
            gen(x + bx2, y + by2, z + bz2, ax, ay, az, bx - bx2, by - by2, bz - bz2, cx, cy, cz);
            gen(x + (ax - dax) + (bx2 - dbx), y + (ay - day) + (by2 - dby), z + (az - daz) + (bz2 - dbz),
                -bx2, -by2, -bz2, cx, cy, cz, -(ax - ax2), -(ay - ay2), -(az - az2));
        } else if (3 * d > 4 * h) {                    // do not split along b
            gen(x, y, z, cx2, cy2, cz2, ax2, ay2, az2, bx, by, bz);
            gen(x + cx2, y + cy2, z + cz2, ax, ay, az, bx, by, bz, cx - cx2, cy - cy2, cz - cz2);
            gen(x + (ax - dax) + (cx2 - dcx), y + (ay - day) + (cy2 - dcy), z + (az - daz) + (cz2 - dcz),
                -cx2, -cy2, -cz2, -(ax - ax2), -(ay - ay2), -(az - az2), bx, by, bz);
        } else {                                       // regular: split along all three
            gen(x, y, z, bx2, by2, bz2, cx2, cy2, cz2, ax2, ay2, az2);
            gen(x + bx2, y + by2, z + bz2, cx, cy, cz, ax2, ay2, az2, bx - bx2, by - by2, bz - bz2);
            gen(x + (bx2 - dbx) + (cx - dcx), y + (by2 - dby) + (cy - dcy), z + (bz2 - dbz) + (cz - dcz),
                ax, ay, az, -bx2, -by2, -bz2, -(cx - cx2), -(cy - cy2), -(cz - cz2));
            gen(x + (ax - dax) + bx2 + (cx - dcx), y + (ay - day) + by2 + (cy - dcy), z + (az - daz) + bz2 + (cz - dcz),
                -cx, -cy, -cz, -(ax - ax2), -(ay - ay2), -(az - az2), bx - bx2, by - by2, bz - bz2);
            gen(x + (ax - dax) + (bx2 - dbx), y + (ay - day) + (by2 - dby), z + (az - daz) + (bz2 - dbz),
                -bx2, -by2, -bz2, cx2, cy2, cz2, -(ax - ax2), -(ay - ay2), -(az - az2));
        }
    }
};

// Fills rank[] (size nx*ny*nz).  Returns false (and a plain row-major order) if the curve failed to be a
// bijection — correctness never depends on the ordering, only tile compactness does.
inline bool hilbert_cell_ranks(int nx, int ny, int nz, std::vector<uint32_t>& rank) {
    const size_t n = (size_t)nx * ny * nz;
    rank.assign(n, UINT32_MAX);
    Gilbert3D g{rank, nx, ny, nz};
    if (nx >= ny && nx >= nz) g.gen(0, 0, 0, nx, 0, 0, 0, ny, 0, 0, 0, nz);
    else if (ny >= nx && ny >= nz) g.gen(0, 0, 0, 0, ny, 0, nx, 0, 0, 0, 0, nz);
    else g.gen(0, 0, 0, 0, 0, nz, nx, 0, 0, 0, ny, 0);
    bool ok = g.next == n && !g.bad;
    if (ok) { std::vector<uint8_t> seen(n, 0); for (size_t i = 0; i < n && ok; ++i) { if (rank[i] >= n || seen[rank[i]]) ok = false; else seen[rank[i]] = 1; } }
    if (!ok) for (size_t i = 0; i < n; ++i) rank[i] = (uint32_t)i;
    return ok;
}

}  // namespace mhip
